#!/bin/bash
# kernel stats of the infer_iground bench (batched decode included): top kernels by total time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_inf}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --mode infer_iground --no_cpu_baseline --steps 1 --warmup 1 > $O/bench.json 2> $O/err.txt
cd $R
python3 tools/dev/top_kernels.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 45 > $O/top_kernels.txt
rm -rf $O/prof
cat $O/top_kernels.txt
