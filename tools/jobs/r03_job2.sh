cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests/test_full_depth_gpu.py -x -q -m gpu -s > gpurun_out/r03/full_depth_tests.log 2>&1
echo "full_depth rc=$?"
tail -3 gpurun_out/r03/full_depth_tests.log
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_abi.py -x -q -m gpu > gpurun_out/r03/train_tests.log 2>&1
echo "train rc=$?"
tail -5 gpurun_out/r03/train_tests.log
timeout 900 python bench.py > gpurun_out/r03/bench_job2.json 2> gpurun_out/r03/bench_job2.err
echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_job2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['cpu_baseline'])"
