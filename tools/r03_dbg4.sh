#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03e}
rm -rf "$O"; mkdir -p "$O"
run() { name=$1; shift; timeout 900 python bench.py --no-cpu --no-probe --steps 5 "$@" > "$O/$name.json" 2> "$O/$name.err"; python tools/_line.py "$O/$name.json"; }
A="--genome-len 262144 --db-window 0"
run a67_1M $A --table-buckets 67000000 --reads 1000000
run a67_span8 $A --table-buckets 67000000 --min-span 8
run a80 $A --table-buckets 80000000
run a100 $A --table-buckets 100000000
run a67_p2 $A --bucket-slots-log2 29
run a67_wide $A --table-buckets 67000000 --identity 52
