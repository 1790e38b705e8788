#!/bin/bash
# debug build of the library: flash_attn2.hip with -DFLASH2_DEBUG, every other object from the product build
set -e
cd /root/repo/grove_amd/csrc
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -DFLASH2_DEBUG -c flash_attn2.hip -o build/flash_attn2_dbg.o
OBJS=$(ls build/*.o | grep -v flash_attn2)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libgrove_hip_dbg.so $OBJS build/flash_attn2_dbg.o
ls -la libgrove_hip_dbg.so
