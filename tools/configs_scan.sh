#!/bin/bash
# secondary configurations (not the headline): paired, spaced, other layouts, long reads.  usage: tools/configs_scan.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { tag=$1; shift; python bench.py --no-cpu --steps 10 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-12s %8.1f Mreads/s  step %.2f ms  kernel %.2f ms' % ('$tag', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))"; }
run single
run paired --paired
run spaced --spacing 1x15,0x15
run bucket --layout bucket
run khash --layout khash
run len1000 --read-len 1000 --reads 2000000
run len100 --read-len 100
run len250 --read-len 250 --reads 5000000
