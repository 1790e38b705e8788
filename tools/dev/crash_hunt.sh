#!/bin/bash
# which combination aborts? (each in its own process)
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/hunt_$name.log 2>&1; echo "$name rc=$? $(grep -c PASSED gpurun_out/hunt_$name.log) $(tail -1 gpurun_out/hunt_$name.log | cut -c1-120)"; }
K="generated_rows or test_gemv_decode" run C X=1
K="full_size_greedy or test_gemv_decode" run D X=1
K="generated_rows or test_gemv_decode" run B GROVE_GEMV_SPLIT_NORM=0
K="generated_rows or test_gemv_decode" run E AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=2
grep -n "rror\|fault\|abort" gpurun_out/hunt_E.log | head -20 | cut -c1-300
