cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests/test_full_depth_gpu.py -x -q -m gpu -k "fp8" -s > gpurun_out/r03/job4_fp8.log 2>&1
echo "fp8 full-depth rc=$?"; tail -3 gpurun_out/r03/job4_fp8.log
timeout 600 python -m pytest tests/test_parity_r2_gpu.py -x -q -m gpu -k "fp8" -s 2>&1 | grep -E "fp8:|passed|failed"
timeout 600 python tools/bench_decode.py > gpurun_out/r03/decode_bench2.json 2> gpurun_out/r03/decode_bench2.err
cat gpurun_out/r03/decode_bench2.json
for dt in fp8 bf16; do
timeout 900 python bench.py --mode infer --frames 32 --dtype $dt --steps 5 --warmup 2 > gpurun_out/r03/bench_infer_$dt.json 2> gpurun_out/r03/bench_infer_$dt.err
echo "infer $dt rc=$?"; python -c "
import json
d=json.loads(open('gpurun_out/r03/bench_infer_$dt.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('box_l1_vs_oracle'))"
done
