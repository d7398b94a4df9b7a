#!/bin/bash
# Round-3 first GPU pass: GPU test tier, then the default bench line and a handful of variants (one JSON line each)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r03a}
rm -rf "$O"; mkdir -p "$O"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"; tail -15 "$O/pytest.log"
fi
run() { name=$1; shift; timeout 900 python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; tail -2 "$O/$name.err"; }
run default
run wide52 --identity 52 --no-cpu --no-probe
run load34 --table-buckets 66000000 --no-cpu --no-probe
run load34_wide --table-buckets 66000000 --identity 52 --no-cpu --no-probe
run load17 --table-buckets 132000000 --no-cpu --no-probe
run allk --genome-len 262144 --db-window 0 --no-cpu --no-probe
run allk_load34 --genome-len 262144 --db-window 0 --table-buckets 67000000 --no-cpu --no-probe
run paired --paired --no-probe --cpu-sample 400000
run hiseq --len-dist hiseq --no-probe --cpu-sample 400000
run k21 --k 21 --no-probe --cpu-sample 400000
run k27 --k 27 --no-probe --cpu-sample 400000
run repeats --genome-model repeats --no-probe --cpu-sample 1000000
