mkdir -p gpurun_out/j18
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "swiglu" > gpurun_out/j18/k.log 2>&1; tail -3 gpurun_out/j18/k.log
for i in 1 2; do for v in 1 0; do GROVE_FUSE_SWIGLU_BWD=$v timeout 600 python bench.py --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('fuse=$v', b['ms_per_step'], b['value'])"; done; done
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_branches_gpu.py tests/test_parity_r2_gpu.py -q -x > gpurun_out/j18/model.log 2>&1; tail -3 gpurun_out/j18/model.log
timeout 900 python -m pytest tests/test_full_depth_gpu.py -q -k "training_vs_oracle_autograd and deep_narrow" > gpurun_out/j18/fd.log 2>&1; tail -3 gpurun_out/j18/fd.log
