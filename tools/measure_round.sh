#!/bin/bash
# The measurement pass of a round on the GPU box: bench line, GEMM shape report, rocprofv3 kernel stats + one-step breakdown, PMC traffic of the
# dominant GEMM shapes, attention and decode microbenchmarks -> gpurun_out/$1/ (copy what is to be judged into profiles/).
#   gpurun --timeout 3000 -- "bash tools/measure_round.sh r04"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-round}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 1200 python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo bench_rc=$?
GROVE_GEMM_REPORT=$O/gemm_shapes.txt timeout 600 python3 $R/bench.py --no_cpu_baseline --serial_towers > $O/bench_serial.json 2>> $O/bench.err; echo serial_rc=$?
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --no_cpu_baseline --serial_towers --steps 5 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; echo prof_rc=$?
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 $R/tools/step_breakdown.py $T > $O/step_breakdown.txt 2>&1; cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -f $T
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/pmc_gemm.py > /dev/null 2> $O/pmc.err; echo pmcf_rc=$?
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/pmc_gemm.py > /dev/null 2>> $O/pmc.err; echo pmcw_rc=$?
python3 $R/tools/pmc_gemm_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_gemm_traffic.json 2>> $O/pmc.err
timeout 600 python3 $R/tools/bench_flash.py > $O/flash_bench.txt 2>&1; echo flash_rc=$?
timeout 600 python3 $R/tools/bench_decode.py > $O/decode_bench.json 2> $O/decode.err; echo decode_rc=$?
ls -la $O | head -30; du -sh $O
# round 6 additions: config 5 (bf16 / fp8 default policy, alternated), config 2, the Winograd stage bench, the fp8 MLP microbench
for i in 1 2; do for dt in bf16 fp8; do timeout 600 python3 $R/bench.py --mode infer --frames 32 --dtype $dt --steps 5 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 > $O/bench_infer_${dt}_$i.json; done; done; echo infer_rc=$?
timeout 900 python3 $R/bench.py --mode infer_iground --no_cpu_baseline 2>/dev/null | tail -1 > $O/bench_infer_iground.json; echo iground_rc=$?
timeout 300 python3 $R/tools/bench_winograd.py > $O/winograd_bench.txt 2>&1; echo wino_rc=$?
timeout 300 python3 $R/tools/dev/bench_sam_mlp_fp8.py > $O/sam_mlp_fp8_bench.txt 2>&1; echo mlp8_rc=$?
timeout 600 python3 $R/tools/bench_decode.py 8 > $O/decode_bench_b8.json 2>> $O/decode.err; echo decode8_rc=$?
ls -la $O | head -40
