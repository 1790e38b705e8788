#!/bin/bash
# kernel-level view of the cached decode step at B sequences (default 8): rocprofv3 kernel stats of tools/bench_decode.py
R=$GRAFT_REPO_ROOT; B=${1:-8}; O=$R/gpurun_out/decode_b$B; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/tools/bench_decode.py $B > $O/bench.json 2> $O/prof.err; echo rc=$?
S=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv; rm -f $(find $O/prof -name "*kernel_trace.csv")
cat $O/bench.json; head -14 $O/kernel_stats.csv | cut -c1-200
