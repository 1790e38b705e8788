#!/bin/bash
# same-box A/B: priority of the SAM tower's stream (and of the optimizer stream)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for pr in 0 -1; do
    GROVE_SAM_STREAM_PRIORITY=$pr python3 bench.py --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sam priority $pr', d['ms_per_step'], d['value'])"
  done
done
