cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -4 gpurun_out/r03/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03/bench_job5.json 2> gpurun_out/r03/bench_job5.err
echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_job5.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('greedy_decode'), d['cpu_baseline']['value'])"
