#!/bin/bash
# in-bench parity sample (GPU vs oracle at full table size) over read lengths and modes.  usage: tools/parity_scan.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python bench.py --steps 3 --warmup 1 --no-probe --reads 1000000 --cpu-sample 100000 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', d['parity_sample'], 'err' if 'error' in d else 'ok')"; }
for L in 31 32 62 94 95 101 151 158 159 250 1000 2100 4200; do run --read-len $L; done
run --read-len 151 --paired
run --read-len 100 --paired
run --spacing 1x15,0x15 --log2-buckets 31
run --spacing 1x15,0x15 --genome-len 262144 --db-window 0
run --genome-len 262144 --db-window 0
run --layout bucket
