#!/bin/bash
# A/B/n of device-library builds inside ONE gpurun call:  tools/abn.sh "libA.so libB.so ..." [bench args]
# First a parity run per library (the in-bench oracle sample), then 3 interleaved timing rounds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1; shift
for lib in $LIBS; do
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib parity', d.get('parity_sample'), d.get('error'))"
done
for rep in 1 2 3; do
  for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --steps 20 "$@" 2>&1 | tail -1 | python tools/_ab_line.py "$lib"
  done
done
