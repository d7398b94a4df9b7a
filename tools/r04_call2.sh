#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c2; mkdir -p "$O"
mkdir -p tools/micro/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/valu_rate tools/micro/valu_rate.hip 2>/dev/null && timeout 300 tools/micro/bin/valu_rate > "$O/valu_rate.txt" 2>&1; tail -80 "$O/valu_rate.txt"
bash tools/r04_ab.sh r04c2 "libbonsai_amd_r03.so libbonsai_amd.so" quick
for a in "-P 4 -p 8 -g 0" "-P 4 -p 8 -g 0,0" "-P 4 -p 8 -g 0,0,0,0"; do echo "== $a" >> "$O/cli2.txt"; timeout 600 python tools/cli_bench.py 64000000 $a >> "$O/cli2.txt" 2>&1; done
grep "M reads/s" "$O/cli2.txt" | cut -c1-170
