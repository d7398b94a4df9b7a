#!/bin/bash
# usage: tools/try.sh  -> runs parity tests (quick) + bench, prints one line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
python bench.py --no-cpu --steps 20 "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('RESULT Mreads/s=%.1f step_ms=%.3f kernel_ms=%.3f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']))"
