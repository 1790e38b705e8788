#!/bin/bash
# one profiled bench (serial towers) -> step breakdown + the neighbours of every copyBuffer / Fill launch in the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05a}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --no_cpu_baseline --serial_towers --steps 4 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; echo prof_rc=$?
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 $R/tools/step_breakdown.py $T 2 70 > $O/step_breakdown.txt 2>&1; cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - $T > $O/copy_neighbours.txt <<'PY'
import csv,sys,re
rows=[(r['Kernel_Name'],int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:r[1])
short=lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::","",n).split("(")[0][:70]
idx=[i for i,r in enumerate(rows) if 'copyBuffer' in r[0] or 'FillFunctor' in r[0] or 'fillBuffer' in r[0]]
seen=set()
for i in idx[-60:]:
    ctx=tuple(short(rows[j][0]) for j in range(max(0,i-2),min(len(rows),i+3)))
    if ctx in seen: continue
    seen.add(ctx)
    print((rows[i][2]-rows[i][1])/1000,'us |',' -> '.join(ctx))
PY
rm -f $T
head -50 $O/step_breakdown.txt; cat $O/copy_neighbours.txt
