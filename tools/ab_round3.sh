#!/bin/bash
# tools/ab_round3.sh "libA.so libB.so ...": parity sample on four shapes per build, then interleaved timing on four shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1
for lib in $LIBS; do
  for shape in "" "--len-dist miseq" "--paired" "--genome-len 262144 --db-window 0"; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 2 --warmup 1 --no-probe --cpu-sample 200000 $shape 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib [$shape] parity', d.get('parity_sample'), d.get('error'))"
  done
done
for shape in "" "--genome-len 262144 --db-window 0" "--paired" "--len-dist hiseq"; do
  for rep in 1 2 3; do
    for lib in $LIBS; do
      BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --steps 20 $shape 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s [%s] kernel_ms %.3f  M reads/s %.1f' % ('$lib', '$shape', d['roofline']['kernel_ms'], d['value']/1e6))"
    done
  done
done
