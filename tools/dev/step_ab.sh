#!/bin/bash
# same-box A/B of the whole step: the product library against a build without the round-5 epilogue drains (libgrove_hip_nodrain.so,
# made by `tools/dev/step_ab.sh build` here), alternated
if [ "$1" = build ]; then
  set -e
  cd /root/repo/grove_amd/csrc
  OBJS=$(ls build/*.o | grep -v "build/gemm.o\|build/gemm_tn.o\|_dbg\|_abl\|_drain\|_nd")
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DPP_NO_EPI_DRAIN -c gemm.hip -o build/gemm_nd.o &
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DTN_NO_EPI_DRAIN -c gemm_tn.hip -o build/gemm_tn_nd.o &
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libgrove_hip_nodrain.so $OBJS build/gemm_nd.o build/gemm_tn_nd.o
  ls -la libgrove_hip_nodrain.so
  exit 0
fi
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libgrove_hip.so libgrove_hip_nodrain.so; do
    GROVE_HIP_LIB=$GRAFT_REPO_ROOT/grove_amd/csrc/$lib python3 bench.py --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done
