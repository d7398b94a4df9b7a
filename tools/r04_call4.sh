#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c4; mkdir -p "$O"
BIG="--genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load --no-probe --no-cpu --steps 3 --warmup 1"
BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 900 python bench.py $BIG > "$O/count_8e9.json" 2> "$O/count_8e9.err"; echo "rc=$?"; python tools/_line.py "$O/count_8e9.json"; grep -o '"debug_fetch_count.*' "$O/count_8e9.json" | cut -c1-700
MID="--genomes 10240 --log2-buckets 32 --no-probe --no-cpu --steps 3 --warmup 1"
BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 900 python bench.py $MID > "$O/count_2e9.json" 2> "$O/count_2e9.err"; echo "rc=$?"; python tools/_line.py "$O/count_2e9.json"; grep -o '"debug_fetch_count.*' "$O/count_2e9.json" | cut -c1-700
