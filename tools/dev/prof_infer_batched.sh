#!/bin/bash
# kernel stats of config 2's batched form alone (two passes: divide by 2)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_infb}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/tools/dev/infer_batched_only.py 2 > $O/out.txt 2> $O/err.txt
cd $R
python3 tools/dev/top_kernels.py $(find $O/prof -name "*kernel_stats.csv" | head -1) 60 > $O/top_kernels.txt
rm -rf $O/prof
cat $O/out.txt $O/top_kernels.txt
