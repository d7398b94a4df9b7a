#!/bin/bash
# quick check of a kernel change: parity-heavy GPU tests, then a few bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03q}; mkdir -p "$O"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_properties.py -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
run() { name=$1; shift; timeout 900 python bench.py --no-probe --steps 20 --cpu-sample 400000 "$@" > "$O/$name.json" 2> "$O/$name.err"; python tools/_line.py "$O/$name.json"; }
run default
run default2
run paired --paired
run load34 --table-buckets 66000000
AK="--genome-len 262144 --db-window 0"
run allk $AK
run allk34 $AK --table-buckets 67000000
run hiseq --len-dist hiseq
run k21 --k 21
