#!/bin/bash
# round 4, second session: tail-pair rounds -- parity, then A/B on the shapes they are for (HiSeq lengths, 100-base reads, k = 21)
# and the default shape (must not move)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04pair
rm -rf "$O"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" "$O/pytest.log" | tail -3
run() { name=$1; shift; echo "$*" > "$O/$name.args"; timeout 900 python bench.py "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; }
S="--no-probe --steps 50 --warmup 3 --cpu-sample 300000"
for d in 0x40 0 0x40 0; do
  run hiseq_$d --len-dist hiseq --ablate $d $S
  run len100_$d --read-len 100 --ablate $d $S
  run k21_$d --k 21 --ablate $d $S
  run len101_$d --read-len 101 --ablate $d $S
done
run default --no-probe --steps 100 --warmup 3 --cpu-sample 300000
run default2 --no-probe --steps 100 --warmup 3 --cpu-sample 300000
run miseq --len-dist miseq $S
run paired $S --paired
