cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_abi.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r03/job3_kernels.log 2>&1
echo "abi+kernels rc=$?"; tail -3 gpurun_out/r03/job3_kernels.log
timeout 1800 python -m pytest tests/test_model_gpu.py tests/test_parity_r2_gpu.py tests/test_train_gpu.py tests/test_preprocess_gpu.py -x -q -m gpu > gpurun_out/r03/job3_model.log 2>&1
echo "model rc=$?"; tail -5 gpurun_out/r03/job3_model.log
timeout 600 python tools/bench_decode.py > gpurun_out/r03/decode_bench.json 2> gpurun_out/r03/decode_bench.err
echo "decode rc=$?"; cat gpurun_out/r03/decode_bench.json
timeout 900 python bench.py --no_cpu_baseline > gpurun_out/r03/bench_job3.json 2> gpurun_out/r03/bench_job3.err
echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_job3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('greedy_decode'))"
