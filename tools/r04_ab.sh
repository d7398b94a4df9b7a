#!/bin/bash
# round-4 kernel A/B inside ONE gpurun call: parity tests on the new library, then interleaved timings of "libA libB ..." on
# the shapes the diet is aimed at.   usage: tools/r04_ab.sh <tag> "<libs>" [quick]
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; LIBS=$2; MODE=${3:-full}
O=gpurun_out/$TAG; mkdir -p "$O"
if [ "$MODE" != "notest" ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_properties.py -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
fi
shape() { name=$1; shift
  for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$name $lib parity', d.get('parity_sample'), d.get('error'))"
  done
  for rep in 1 2 3; do for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --steps 20 "$@" 2>&1 | tail -1 | python tools/_ab_line.py "$name:$lib"
  done; done; }
shape default
shape hiseq --len-dist hiseq
shape k21 --k 21
shape allk --genome-len 262144 --db-window 0
if [ "$MODE" == "full" ]; then
  shape paired --paired
  shape len100 --read-len 100
fi
