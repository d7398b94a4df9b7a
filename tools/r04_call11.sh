#!/bin/bash
# round 4, second session: full GPU suite, fetch counters of the 8e9-key table under both fills, the parallel gzip reader
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s2
rm -rf "$O"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" "$O/pytest.log" | tail -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBNS_COUNT_FETCHES -Iinclude bonsai_amd/csrc/bns_api.hip -o bonsai_amd/lib/libbonsai_amd_count.so 2>/dev/null
BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 1500 python tools/big_stream.py 36000 34 0 0 10,0 > "$O/big8e9_counts.txt" 2>&1; echo "big rc=$?"; grep "^dbg" "$O/big8e9_counts.txt"
timeout 1200 python tools/gz_bench.py 8000000 > "$O/gz_bench.txt" 2>&1; echo "gz rc=$?"; grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" "$O/gz_bench.txt"
