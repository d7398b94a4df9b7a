#!/bin/bash
# configs[3]'s per-rank slice next to the 8e9-key every-k-mer db, ranks 0 and 7 of 8, on the final round-4 library
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04slice; rm -rf "$O"; mkdir -p "$O"
for r in 0 7; do
  A="--genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load --scaling strong --total-reads 1000000000 --world 8 --steps 3 --warmup 1 --cpu-sample 200000 --no-probe --emulate-rank $r"
  echo "$A" > "$O/c3_allk_8e9_r$r.args"
  timeout 1500 python bench.py $A > "$O/c3_allk_8e9_r$r.json" 2> "$O/c3_allk_8e9_r$r.err"; echo "rank $r rc=$?"; python tools/_line.py "$O/c3_allk_8e9_r$r.json"
done
