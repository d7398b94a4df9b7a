#!/bin/bash
# local helper: gpurun with retry while the pod's GPU slots are busy (exit 3).  usage: tools/dev/grun.sh <timeout> <log> -- cmd...
T=$1; LOG=$2; shift 3
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > "$LOG" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
