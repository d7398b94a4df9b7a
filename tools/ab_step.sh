#!/bin/bash
# like tools/ab.sh but compares ms_per_step (whole step, all kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    BONSAI_AMD_LIB=$PWD/$lib python bench.py --no-cpu --steps 20 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib step_ms %.3f kernel_ms %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
