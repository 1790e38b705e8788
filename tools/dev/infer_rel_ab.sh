#!/bin/bash
# config 2's batched form: rel-pos in the window kernels (1) against the two streams (0), alternated on one box
for r in 0 1 0 1; do echo "GROVE_SAM_REL_IN_KERNEL=$r"; GROVE_SAM_REL_IN_KERNEL=$r python3 tools/dev/infer_batched_only.py 4 2>&1 | grep pass | tail -3; done
