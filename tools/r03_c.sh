#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03c}; mkdir -p "$O"
timeout 2400 python -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.log"
run() { name=$1; shift; timeout 1500 python bench.py --no-probe --steps 5 --warmup 1 --no-cpu "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; grep "bench.py: table" "$O/$name.err"; }
run default
run w50_9e8 --genomes 4096 --log2-buckets 31
AK="--genome-len 262144 --db-window 0"
run allk $AK
run allk_1e9 $AK --genomes 4096 --log2-buckets 31
run allk_load34 $AK --table-buckets 67000000
