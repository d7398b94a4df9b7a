#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03d}
rm -rf "$O"; mkdir -p "$O"
cnt() { name=$1; shift; BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 900 python bench.py --no-cpu --no-probe --steps 3 --warmup 1 "$@" > "$O/$name.json" 2> "$O/$name.err"; python - <<PY
import json
d=json.loads([l for l in open("$O/$name.json") if l.startswith("{")][-1])
f=d["debug_fetch_count"]; n=d["config"]["reads_per_gpu"]
print("$name kernel %.2f ms  buckets/read %.2f  passes/read %.2f ovf lookups/read %.3f rounds %.3f  ovf %s spilled %s" % (d["roofline"]["kernel_ms"], f["buckets_fetched_per_launch"]/n, f["probe_passes_per_launch"]/n, f["overflow_lookups_per_launch"]/n, f["rounds_with_overflow_lookups_per_launch"]/n, d["config"]["table_overflow_keys"], d["config"]["table_spilled_keys"]))
PY
}
cnt allk34 --genome-len 262144 --db-window 0 --table-buckets 67000000
cnt allk17 --genome-len 262144 --db-window 0 --table-buckets 134000000
cnt allk8 --genome-len 262144 --db-window 0
cnt default
BNS_BENCH_ONE_DEVICE=1 BNS_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --genomes 32 --genome-len 65536 --log2-buckets 22 --reads 40000 --steps 2 --warmup 1 --no-probe --no-cpu > "$O/g2.json" 2> "$O/g2.err"; echo "g2 rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/g2.json") if l.startswith("{")][-1])
print([r.get("parity_sample") for r in d.get("per_rank")], d.get("error"))
PY
