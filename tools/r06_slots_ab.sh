#!/bin/bash
# round 6: the page-locked slots of the BGZF path from hipHostMalloc (BNS_BGZF_SLOT_MALLOC=1, as before) against registered huge-page memory of
# our own (tools/micro/pin_bench.hip: 2 ms per 96 MiB instead of 15-45):  tools/r06_slots_ab.sh [reads=64000000]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
python tools/r05_bgzf_make.py $N | tail -1
D=/tmp/bgzfbench
cp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz; cat $D/r2.bgzf.fq.gz > /dev/null
run() {  # label files...
  local label=$1; shift
  s=$(date +%s.%N)
  BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp "$@" 2>&1 | grep -E "process_dataset|page-lock" | sed -E 's/.*(page-lock [0-9.]+ \(summed\), first batch inflated after [0-9.]+ s).*/\1/' | tr '\n' ' '
  e=$(date +%s.%N)
  python3 -c "print('<- $label: wall %.2f s = %.1f M reads(mates)/s' % ($e - $s, $N * $# / ($e - $s) / 1e6))"
}
for rep in 1 2 3; do
  BNS_BGZF_SLOT_MALLOC=1 run "hipHostMalloc slots, BGZF" $D/r.bgzf.fq.gz
  run "registered slots, BGZF" $D/r.bgzf.fq.gz
  BNS_BGZF_SLOT_MALLOC=1 run "hipHostMalloc slots, BGZF pair" $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
  run "registered slots, BGZF pair" $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
done
