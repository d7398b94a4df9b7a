#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c6; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -q -x > "$O/pytest_cli.log" 2>&1; echo "cli tests rc=$?"; tail -3 "$O/pytest_cli.log"
timeout 1500 python tools/ingest_bench.py 32000000 > "$O/ingest.txt" 2>&1; echo "ingest rc=$?"; cat "$O/ingest.txt" | grep -v amdgpu | cut -c1-330
