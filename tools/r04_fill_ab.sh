#!/bin/bash
# Group-aware fill vs arrival-order fill (round 4): same shapes, debug bit 0x10 = arrival order, 0 = the loader's choice,
# 0x8010 / 0x8000 = the crowded-table lookup (cooperative overflow lookup + tag bits) forced.   gpurun -- bash tools/r04_fill_ab.sh [tag] [big]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-fillab}
rm -rf "$O"; mkdir -p "$O"
run() { name=$1; shift; echo "$*" > "$O/$name.args"; timeout 900 python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; }
S="--no-probe --no-cpu --steps 20 --warmup 2"
AK="--genome-len 262144 --db-window 0"
for d in 0x10 0 0x8010 0x8000; do
  run allk34_$d $AK --table-buckets 67000000 --ablate $d $S
done
for d in 0x10 0 0x8000; do
  run w50load34_$d --table-buckets 66000000 --ablate $d $S
  run allk8_$d $AK --ablate $d $S
done
run default_0 $S
if [ "${2:-}" = "big" ]; then
  timeout 1500 python tools/big_stream.py 36000 34 0 0 10,0 > "$O/big8e9.txt" 2>&1; echo "big rc=$?"; grep -v "^ROCm\|^Hostname\|^Librccl" "$O/big8e9.txt"
fi
