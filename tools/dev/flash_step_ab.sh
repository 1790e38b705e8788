#!/bin/bash
# same-box A/B of the whole step over the eight-wave attention mask (grove_flash_attn_set_v2): 15 = shipped, 31 = + the dQ kernel at head
# dim 128 (LLaMA), 0 = the four-wave kernels everywhere
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for m in ${@:-15 31 0}; do
    GROVE_FLASH_V2=$m python3 bench.py --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $m', d['ms_per_step'], d['value'])"
  done
done
