#!/bin/bash
# same-box A/B of the cached decode step and the GEMV shapes: the product library against libgrove_hip_prev.so (a build with the previous gemv.hip)
cd $GRAFT_REPO_ROOT
for lib in libgrove_hip_prev.so libgrove_hip.so; do echo "== $lib"; GROVE_HIP_LIB=$GRAFT_REPO_ROOT/grove_amd/csrc/$lib python3 tools/bench_gemv.py 8 2>&1 | grep "M=8"; done
for i in 1 2 3; do
  for lib in libgrove_hip_prev.so libgrove_hip.so; do
    for B in 8 1; do
    GROVE_HIP_LIB=$GRAFT_REPO_ROOT/grove_amd/csrc/$lib python3 tools/bench_decode.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib B=$B', {k: d[k] for k in d if 'ms_per_token' in k})"
    done
  done
done
