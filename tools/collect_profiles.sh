#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command, then PMC counters in
# their own passes (never combined with other trace domains).  Output under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=gpurun_out/prof_$TAG
rm -rf "$P"; mkdir -p "$P"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o bench -- python bench.py --no-cpu $EXTRA > "$P/bench_kt.log" 2>&1
for pass in "FETCH_SIZE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
            "TCC_BUBBLE_sum TCC_REQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d" " -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$P/pmc_$n" -o bench -- python bench.py --no-cpu --steps 2 --warmup 1 $EXTRA > "$P/pmc_$n.log" 2>&1
done
grep '^{' "$P/bench_kt.log" | tail -1 | cut -c1-400
