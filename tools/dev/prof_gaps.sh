#!/bin/bash
# GPU idle gaps inside one training step (default overlap of the towers): rocprofv3 kernel trace as a rocpd database -> tools/step_gaps.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gaps}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o ks -- python3 $R/bench.py --no_cpu_baseline --steps 4 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; echo prof_rc=$?
D=$(find $O/prof -name "*.db" | head -1); python3 $R/tools/step_gaps.py $D 4 25 > $O/step_gaps.txt 2>&1; rm -f $D; cat $O/step_gaps.txt
