cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python3 tools/bench_flash.py > gpurun_out/r03/flash_bench_base.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/r03/pmc_attn -o pmc --output-format csv -- python3 tools/pmc_flash_general.py > gpurun_out/r03/pmc_attn.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r03/pmc_attn flash > gpurun_out/r03/pmc_attention_general_base.json 2>> gpurun_out/r03/pmc_attn.log
rm -rf gpurun_out/r03/pmc_attn/*/*.db 2>/dev/null
python3 bench.py > gpurun_out/r03/bench_base.json 2> gpurun_out/r03/bench_base.err
tail -1 gpurun_out/r03/bench_base.json | cut -c1-600
cat gpurun_out/r03/flash_bench_base.txt
