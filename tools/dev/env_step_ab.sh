#!/bin/bash
# same-box A/B of the whole training step over one environment knob: env_step_ab.sh VAR A B [rounds]
V=$1; A=$2; B=$3; N=${4:-3}
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for x in $A $B; do
    env $V=$x python3 bench.py --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', d['ms_per_step'], d['value'])"
  done
done
