cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc_win -o pmc -- python3 tools/pmc_flash_win.py > gpurun_out/pmc_win.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_win win > gpurun_out/pmc_win_summary.json
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/pmc_win2 -o pmc -- python3 tools/pmc_flash_win.py > gpurun_out/pmc_win2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_win2 win > gpurun_out/pmc_win2_summary.json
python3 - <<'PY'
import json
for f in ('gpurun_out/pmc_win_summary.json','gpurun_out/pmc_win2_summary.json'):
    d=json.load(open(f))
    for k,v in d.items():
        print(k[:40], {c:(round(x,4) if isinstance(x,float) and abs(x)<100 else int(x)) for c,x in v.items()})
PY
