#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03c}
rm -rf "$O"; mkdir -p "$O"
small="--genomes 32 --genome-len 65536 --log2-buckets 22 --reads 40000 --steps 2 --warmup 1 --no-probe --no-cpu"
BNS_BENCH_ONE_DEVICE=1 BNS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 $small > "$O/g2.json" 2> "$O/g2.err"; echo "g2 rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/g2.json") if l.startswith("{")][-1])
print(d.get("per_rank"), d.get("error"))
PY
grep -v "amdgpu.ids\|socket.cpp" "$O/g2.err" | tail -5
