#!/bin/bash
# ablation builds of the TN weight-gradient kernel (wrong results by construction): libgrove_hip_tnabl<n>.so, n = 1 plain ds_read_b64
# instead of the transposed read, 2 no LDS-DMA after the first K tile, 3 no fragment reads after it, 4 neither (MFMAs + barriers)
set -e
cd /root/repo/grove_amd/csrc
OBJS=$(ls build/*.o | grep -v "gemm_tn\|_dbg\|_abl")
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DTN_ABL=$n -c gemm_tn.hip -o build/gemm_tn_abl$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libgrove_hip_tnabl$n.so $OBJS build/gemm_tn_abl$n.o
done
ls -la libgrove_hip_tnabl*.so
