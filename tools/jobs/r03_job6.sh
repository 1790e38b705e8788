cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03/prof
# 1. per-kernel time of the committed bench command with the towers serialised (the roofline's own configuration)
GROVE_GEMM_REPORT=gpurun_out/r03/gemm_shapes.txt rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof -o r03 --output-format csv -- python3 bench.py --serial_towers --steps 5 --warmup 2 --no_cpu_baseline > gpurun_out/r03/bench_serial.json 2> gpurun_out/r03/bench_serial.err
echo "prof rc=$?"
cp $(find gpurun_out/r03/prof -name "*kernel_stats.csv" | head -1) gpurun_out/r03/kernel_stats.csv
python3 tools/step_breakdown.py $(find gpurun_out/r03/prof -name "*kernel_trace.csv" | head -1) > gpurun_out/r03/step_breakdown.txt 2>&1
find gpurun_out/r03/prof -name "*kernel_trace.csv" -delete; find gpurun_out/r03/prof -name "*.db" -delete
tail -1 gpurun_out/r03/bench_serial.json | cut -c1-300
# 2. memory-side bytes of the dominant GEMM shapes: separate --pmc passes (the guide's HBM section)
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r03/pmc_fetch -o f --output-format csv -- python3 tools/pmc_gemm.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r03/pmc_write -o w --output-format csv -- python3 tools/pmc_gemm.py > /dev/null 2>&1
python3 tools/pmc_gemm_traffic.py gpurun_out/r03/pmc_fetch gpurun_out/r03/pmc_write > gpurun_out/r03/pmc_gemm_traffic.json 2> gpurun_out/r03/pmc_traffic.err
echo "traffic rc=$?"; head -c 600 gpurun_out/r03/pmc_gemm_traffic.json
# 3. decode: kernel-trace of the replayed step
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_decode -o dec --output-format csv -- python3 tools/bench_decode.py > gpurun_out/r03/decode_prof.json 2>/dev/null
cp $(find gpurun_out/r03/prof_decode -name "*kernel_stats.csv" | head -1) gpurun_out/r03/decode_kernel_stats.csv
find gpurun_out/r03/prof_decode -name "*kernel_trace.csv" -delete; find gpurun_out/r03/prof_decode -name "*.db" -delete
head -12 gpurun_out/r03/decode_kernel_stats.csv | cut -c1-150
