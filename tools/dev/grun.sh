#!/bin/bash
# gpurun with retries while the pod's slots are busy (exit 3 = nothing charged). usage: grun.sh <timeout> <logfile> '<command>'
T=$1; LOG=$2; shift 2
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
