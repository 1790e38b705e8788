#!/bin/bash
# one profiled bench (serial towers) -> step breakdown + the torch-side launches of the step with their neighbours
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06b}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --no_cpu_baseline --serial_towers --steps 4 --warmup 2 > $O/prof_bench.json 2> $O/prof.err; echo prof_rc=$?
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); python3 $R/tools/step_breakdown.py $T 2 80 > $O/step_breakdown.txt 2>&1; cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 $R/tools/dev/step_torch_ops.py $T 2 > $O/step_torch_ops.txt 2>&1
rm -f $T
head -12 $O/step_breakdown.txt; cat $O/step_torch_ops.txt
