#!/bin/bash
# round 4, last pass on the final sources: full GPU suite, bench line, CLI (plain / paired / ingest forms), gzip readers
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04final
rm -rf "$O"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" "$O/pytest.log" | tail -3
timeout 600 python bench.py > "$O/bench.out" 2> "$O/bench.err"; grep '^{' "$O/bench.out" | tail -1 | cut -c1-200
{
  for a in "" "--paired" "-P 1"; do timeout 300 python tools/cli_bench.py 64000000 $a 2>&1 | grep -E "^\[|^args"; done
  echo "== ingest forms (tools/ingest_bench.py 32000000)"
  timeout 1500 python tools/ingest_bench.py 32000000 2>&1 | grep -v amdgpu
} > "$O/cli.txt" 2>&1; echo "cli rc=$?"; grep "^args\|M reads/s" "$O/cli.txt" | cut -c1-180
rm -rf /tmp/clibench /tmp/ingestbench; timeout 900 python tools/gz_bench.py 8000000 > "$O/gz_bench.txt" 2>&1; grep "M reads/s\|identical" "$O/gz_bench.txt" | cut -c1-140
rm -rf /tmp/gzbench; bash tools/gz_scaling.sh > "$O/gz_scaling.txt" 2>&1; grep "reads/s" "$O/gz_scaling.txt" | cut -c1-140
