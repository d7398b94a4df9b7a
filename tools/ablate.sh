#!/bin/bash
# profiling aid: classify_kernel time under ablation bits (1 no probe, 2 no vote, 4 no minimizer window)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
BNS_ABLATION=1 python -c "from bonsai_amd.build import build_device_library as b; b(force=True)" > /dev/null   # ablation build (restore with python -m bonsai_amd.build)
LAYOUT=${1:-minbucket}
for extra in ""; do
  for ab in 0 1 2 3 8 16 24 32 35 43; do
    python bench.py --no-cpu --layout $LAYOUT --ablate $ab $extra 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ablate=$ab', '$extra', 'kernel_ms=%.2f' % d['roofline']['kernel_ms'], 'step_ms=%.2f' % d['ms_per_step'])"
  done
done
