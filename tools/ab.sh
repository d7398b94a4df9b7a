#!/bin/bash
# A/B of device-library builds inside ONE gpurun call (boxes differ by ~10%): tools/ab.sh libA.so libB.so [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in "$A" "$B"; do
    BONSAI_AMD_LIB=$PWD/$lib python bench.py --no-cpu --steps 20 "$@" 2>&1 | tail -1 | python tools/_ab_line.py "$lib"
  done
done
