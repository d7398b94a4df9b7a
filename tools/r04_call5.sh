#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c5; mkdir -p "$O"
timeout 2400 python -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.log"
LIBS="libbonsai_amd_r03.so libbonsai_amd.so"
shape() { name=$1; shift
  for rep in 1 2; do for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib timeout 900 python bench.py --no-probe --steps 5 --warmup 1 --cpu-sample 200000 "$@" 2>/dev/null | tail -1 > "$O/$name.$lib.$rep.json"; echo -n "$name:$lib "; python tools/_line.py "$O/$name.$lib.$rep.json"
  done; done; }
shape allk34 --genome-len 262144 --db-window 0 --table-buckets 67000000
shape allk3p6e9 --genome-len 262144 --db-window 0 --genomes 16384 --log2-buckets 33
shape big8e9 --genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load
