#!/bin/bash
# RefSeq-scale tables (configs[3]'s db size on ONE GPU):  gpurun --timeout 3000 -- bash tools/r03_big.sh [tag]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03big}; mkdir -p "$O"
run() { name=$1; shift; timeout 1500 python bench.py --no-probe --steps 5 --warmup 1 "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; grep "bench.py: table" "$O/$name.err"; }
run w50_9e8 --genomes 4096 --log2-buckets 31 --cpu-sample 200000
AK="--genome-len 262144 --db-window 0"
run allk_1e9 $AK --genomes 4096 --log2-buckets 31 --cpu-sample 200000
run allk_4e9 $AK --genomes 16384 --log2-buckets 33 --cpu-sample 200000
run allk_4e9_id32 $AK --genomes 16384 --log2-buckets 33 --no-cpu --identity 32
timeout 1500 python tools/big_stream.py 36000 34 > "$O/big8e9.log" 2>&1; echo "big rc=$?"; grep -v amdgpu.ids "$O/big8e9.log" | tail -8
