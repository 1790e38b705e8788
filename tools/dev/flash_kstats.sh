#!/bin/bash
# per-kernel average durations of tools/bench_flash.py (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-flash_ks}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/tools/bench_flash.py > $O/bench.txt 2> $O/prof.err
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv
python3 - $S <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::|void ","",r["Name"]).split("(")[0]
    if any(k in n for k in ("flash","win_attn","rel_bias")):
        print(f'{n:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1000:9.1f} us')
PY
