#!/bin/bash
# kernel timeline of one training step inside [lo, hi] ms: prof_timeline.sh <out> <lo> <hi>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tl}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o ks -- python3 $R/bench.py --no_cpu_baseline --steps 4 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; echo prof_rc=$?
D=$(find $O/prof -name "*.db" | head -1); python3 $R/tools/step_timeline.py $D 4 ${2:-90} ${3:-112} > $O/timeline.txt 2>&1; python3 $R/tools/step_gaps.py $D 4 5 > $O/gaps.txt 2>&1; rm -f $D; head -3 $O/timeline.txt; wc -l $O/timeline.txt
