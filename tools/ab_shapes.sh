#!/bin/bash
# several builds x several workload shapes inside one box: tools/ab_shapes.sh "libA.so libB.so" 
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1
for shape in "" "--len-dist hiseq"; do
  for rep in 1 2; do
    for lib in $LIBS; do
      BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --steps 10 $shape 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s [%s] kernel_ms %.3f' % ('$lib', '$shape', d['roofline']['kernel_ms']))"
    done
  done
done
