#!/bin/bash
# spaced-seed classification (BASELINE configs[2]) under the three table layouts.  usage: tools/spaced_scan.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { tag=$1; shift; python bench.py --no-cpu --steps 10 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-22s %8.1f Mreads/s  step %.2f ms  kernel %.2f ms' % ('$tag', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for lay in minbucket bucket khash; do
  run "spaced $lay" --spacing 1x15,0x15 --layout $lay
  run "spaced paired $lay" --spacing 1x15,0x15 --layout $lay --paired
done
