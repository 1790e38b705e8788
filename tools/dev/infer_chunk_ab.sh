#!/bin/bash
# config 2's batched form at several windows-per-forward settings (same box)
for w in 40 20 10 8 5 4; do echo "windows per forward $w"; GROVE_INFER_WINDOWS_PER_FORWARD=$w python3 tools/dev/infer_batched_only.py 3 2>&1 | grep pass; done
