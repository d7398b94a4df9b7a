#!/bin/bash
# classify_kernel time vs resident blocks per CU (profiling aid; BNS_BLOCKS_PER_CU caps the grid)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for b in 1 2 4 6 8; do
  BNS_BLOCKS_PER_CU=$b python bench.py --no-cpu --steps 10 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('blocks/CU $b kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done
