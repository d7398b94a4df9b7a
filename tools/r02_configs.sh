#!/bin/bash
# Secondary bench lines of round 2, ONE gpurun call:  gpurun --timeout 3000 -- bash tools/r02_configs.sh [tag]
#   * on the default db (configs[1] as named): configs[2] (spaced seed, paired), paired, k = 21, 100/250 bp, HiSeq / MiSeq lengths
#   * the every-k-mer db rounds 1-2 were tuned on, for continuity
#   * load-factor sweep: the same reads against the same db in a clustered table of 16x / 4x / 2x / 1x, and against dbs of
#     1e9 and 4e9 keys in the 137 GB table
# Each line of gpurun_out/<tag>/configs.jsonl = {"name": ..., "args": ..., "bench": <the bench JSON line>}.
set -u
TAG=${1:-r02b}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG
mkdir -p "$O"; : > "$O/configs.jsonl"
run() {
  name=$1; shift
  echo "== $name: $*"
  line=$(timeout 900 python bench.py --steps 10 --warmup 2 --no-probe "$@" 2> "$O/$name.err" | grep '^{' | tail -1)
  if [ -z "$line" ]; then line='null'; tail -3 "$O/$name.err"; fi
  printf '{"name": "%s", "args": "%s", "bench": %s}\n' "$name" "$*" "$line" >> "$O/configs.jsonl"
  echo "$line" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('   %.1f M reads/s  kernel %.3f ms  frac %.3f  load %.3f  overflow_keys %s  parity %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['load_factor'], d['config']['table_overflow_keys'], d.get('parity_sample')))
except Exception as e: print('   no line', e)"
}
# the default db (configs[1] as named: w = 50 entropy minimizers of 1024 genomes x 2.6 Mb)
run c2_spaced_paired --spacing 1x15,0x15 --paired --log2-buckets 31
run paired --paired
run k21 --k 21
run k27 --k 27
run len100 --read-len 100
run len250 --read-len 250
run hiseq_lengths --len-dist hiseq
run miseq_lengths --len-dist miseq
run load_4x --bucket-slots-log2 31
run load_2x --bucket-slots-log2 30
run load_1x --bucket-slots-log2 29
run keys_1e9 --genomes 4096 --log2-buckets 31 --no-cpu
# the every-k-mer db of rounds 1-2 (1024 genomes x 256 kb, every k-mer a key: the heavier vote)
AK="--genome-len 262144 --db-window 0"
run allkmers $AK
run allkmers_c2_spaced_paired $AK --spacing 1x15,0x15 --paired
run allkmers_paired $AK --paired
run allkmers_hiseq_lengths $AK --len-dist hiseq
run allkmers_load_1x $AK --bucket-slots-log2 29
run allkmers_keys_4e9 $AK --genomes 16384 --log2-buckets 33 --no-cpu
run allkmers_keys_8e9_khash $AK --genomes 36000 --log2-buckets 34 --layout khash --no-cpu --steps 3 --warmup 1
cat "$O/configs.jsonl" | wc -l
