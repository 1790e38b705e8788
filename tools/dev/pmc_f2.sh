cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc_f2 -o pmc -- python3 tools/pmc_flash_general.py > gpurun_out/pmc_f2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_f2 flash > gpurun_out/pmc_f2_summary.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_f2_summary.json'))
for k,v in d.items():
    if 'flash2' in k or 'fwd' in k:
        print(k, {c: (round(x,4) if isinstance(x,float) and x<10 else int(x)) for c,x in v.items()})
PY
