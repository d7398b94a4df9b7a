#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c7; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_nthash.py -m gpu -q -x > "$O/pytest.log" 2>&1; echo "tests rc=$?"; tail -3 "$O/pytest.log"
timeout 1800 python tools/ingest_bench.py 32000000 > "$O/ingest.txt" 2>&1; echo "ingest rc=$?"; cat "$O/ingest.txt" | grep -v amdgpu | cut -c1-330
