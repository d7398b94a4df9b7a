#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03b}
rm -rf "$O"; mkdir -p "$O"
small="--genomes 32 --genome-len 65536 --log2-buckets 22 --reads 40000 --steps 2 --warmup 1 --no-probe --cpu-sample 40000"
BNS_BENCH_FORCE_DIST=1 timeout 600 python bench.py $small > "$O/fd.json" 2> "$O/fd.err"; echo "fd rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/fd.json") if l.startswith("{")][-1])
print(d.get("per_rank"), d.get("parity_sample"), d.get("error"))
PY
tail -3 "$O/fd.err"
run() { name=$1; shift; timeout 900 python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; tail -2 "$O/$name.err" | grep -v amdgpu.ids; }
run allk34 --genome-len 262144 --db-window 0 --table-buckets 67000000 --no-probe --cpu-sample 400000 --steps 5
run allk34_ovcoff --genome-len 262144 --db-window 0 --table-buckets 67000000 --no-probe --no-cpu --steps 5 --ablate 0x2000
run allk17 --genome-len 262144 --db-window 0 --table-buckets 134000000 --no-probe --no-cpu --steps 5
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_dist.py::test_bench_self_launch_two_ranks > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -25 "$O/pytest.log"
