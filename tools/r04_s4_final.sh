#!/bin/bash
# round 4, last session: full GPU suite on the library with the inflate kernel, bench line, kernel stats of the inflate bench, BGZF CLI forms
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s4
rm -rf "$O"; mkdir -p "$O"
timeout 1700 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" "$O/pytest.log" | tail -3
timeout 600 python bench.py > "$O/bench.out" 2> "$O/bench.err"; grep '^{' "$O/bench.out" | tail -1 | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_inflate" -o inflate -- python tools/inflate_bench.py 512 4096,16384 > "$O/prof_inflate.log" 2>&1; echo "rocprof rc=$?"
find "$O/prof_inflate" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats_inflate.csv"; head -5 "$O/kernel_stats_inflate.csv" | cut -c1-200
timeout 900 python tools/bgzf_gpu_bench.py 32000000 > "$O/bgzf_gpu.txt" 2>&1; echo "bgzf rc=$?"; grep "M reads/s\|identical" "$O/bgzf_gpu.txt" | cut -c1-150
