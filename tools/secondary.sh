#!/bin/bash
# Secondary measurements (one gpurun call; rounds 3-4): the bench line on the other workload shapes, the big tables, the 8e9-key
# streamed db, the PCIe-inclusive host entry points and the CLI.  Reduced by tools/summarize_secondary.py into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r04c}
rm -rf "$O"; mkdir -p "$O"
run() { name=$1; shift; echo "$*" > "$O/$name.args"; timeout 1500 python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; }
S="--no-probe --no-cpu"
run load34 --table-buckets 66000000 $S
run load17 --table-buckets 132000000 $S
run wide52 --identity 52 $S
run packed --packed $S
run paired --paired $S
run hiseq --len-dist hiseq $S
run len100 --read-len 100 $S
run len250 --read-len 250 $S
run k21 --k 21 $S
run k27 --k 27 $S
run repeats --genome-model repeats $S
run allk --genome-len 262144 --db-window 0 $S
run allk_load34 --genome-len 262144 --db-window 0 --table-buckets 67000000 $S
B="--steps 5 --warmup 1 $S"
run w50_9e8 --genomes 4096 --log2-buckets 31 $B
AK="--genome-len 262144 --db-window 0"
run allk_9e8 $AK --genomes 4096 --log2-buckets 31 $B
run allk_1p8e9 $AK --genomes 8192 --log2-buckets 32 $B
run allk_3p6e9 $AK --genomes 16384 --log2-buckets 33 $B
run w50_2e9 --genomes 10240 --log2-buckets 32 $B
run allk_8e9_streamed --genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load --steps 5 --warmup 1 --no-probe --cpu-sample 200000
run len10k --read-len 10000 --reads 100000 --steps 20 $S
run spaced_paired --spacing 1x15,0x15 --paired --log2-buckets 31 --steps 20 $S
timeout 600 python tools/host_path_bench.py 10000000 2> "$O/host_path.err" | grep '^{' > "$O/host_path.jsonl"; echo "host_path rc=$?"; cut -c1-160 "$O/host_path.jsonl"
{
  for a in "" "--paired" "-P 1" "-p 8"; do timeout 300 python tools/cli_bench.py 64000000 $a 2>&1 | grep -E "^\[|^args"; done
  echo "== ingest forms (tools/ingest_bench.py 32000000)"
  timeout 1500 python tools/ingest_bench.py 32000000 2>&1 | grep -v amdgpu
} > "$O/cli.txt" 2>&1; echo "cli rc=$?"; grep "^args\|M reads/s" "$O/cli.txt" | cut -c1-200
