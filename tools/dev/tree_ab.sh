#!/bin/bash
# same-box A/B of the whole training step: this tree against the tree in ab_old/ (git archive of an earlier commit + its own build), alternated
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for t in ab_old .; do
    (cd $t && python3 bench.py --no_cpu_baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree $t', d['ms_per_step'], d['value'], d.get('step_roofline', {}).get('frac'))")
  done
done
