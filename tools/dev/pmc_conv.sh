cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $c | cut -c1-8 | tr ' ' _)
  rocprofv3 --pmc $c -d gpurun_out/pmc_conv/$n -o p -- python3 tools/dev/pmc_conv.py > gpurun_out/pmc_conv_$n.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_conv/$n gemm > gpurun_out/pmc_conv_$n.json
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/pmc_conv_*.json')):
    d=json.load(open(f))
    for k,v in d.items():
        print(f.split('pmc_conv_')[1][:8], k[:55], {c:(round(x,4) if isinstance(x,float) and abs(x)<100 else int(x)) for c,x in v.items()})
PY
