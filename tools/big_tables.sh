#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03big2}; mkdir -p "$O"
timeout 2400 python -m pytest tests -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.log"
run() { name=$1; shift; timeout 1500 python bench.py --no-probe --steps 5 --warmup 1 --no-cpu "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; grep "bench.py: table" "$O/$name.err"; }
run w50_9e8_id32 --genomes 4096 --log2-buckets 31 --identity 32
run w50_9e8_id52 --genomes 4096 --log2-buckets 31 --identity 52
AK="--genome-len 262144 --db-window 0"
run allk_1e9_id32 $AK --genomes 4096 --log2-buckets 31 --identity 32
run allk_1e9_id52 $AK --genomes 4096 --log2-buckets 31 --identity 52
run allk_2e9_id32 $AK --genomes 8192 --log2-buckets 32 --identity 32
run allk_2e9_id52 $AK --genomes 8192 --log2-buckets 32 --identity 52
timeout 1500 python tools/big_stream.py 36000 34 > "$O/big8e9.log" 2>&1; echo "big rc=$?"; grep -v amdgpu.ids "$O/big8e9.log" | tail -6
