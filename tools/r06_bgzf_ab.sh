#!/bin/bash
# round 6: the BGZF device paths (one file, a pair) with the round-5 tree (gpurun_scratch/r05: one classify launch per 64 MiB slice) against the
# current one (records of a call's slices in one launch), interleaved on one box:  tools/r06_bgzf_ab.sh [reads=64000000]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
python tools/r05_bgzf_make.py $N | tail -1
D=/tmp/bgzfbench
cp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz; cat $D/r2.bgzf.fq.gz > /dev/null
OLD=gpurun_scratch/r05/bonsai_amd/bin/bonsai; NEW=bonsai_amd/bin/bonsai
run() {  # label binary files...
  local label=$1 bin=$2; shift 2
  s=$(date +%s.%N)
  BNS_CLI_TIMING=1 $bin classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp "$@" 2>&1 | grep -E "BGZF|process_dataset" | sed -E 's/; page-lock.*//' | cut -c1-420
  e=$(date +%s.%N)
  python3 -c "print('$label: wall %.2f s = %.1f M reads(mates)/s' % ($e - $s, $N * $# / ($e - $s) / 1e6))"
}
for rep in 1 2 3; do
  [ -x $OLD ] && run "r05 single" $OLD $D/r.bgzf.fq.gz
  run "r06 single" $NEW $D/r.bgzf.fq.gz
done
for rep in 1 2; do
  [ -x $OLD ] && run "r05 pair" $OLD $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
  run "r06 pair" $NEW $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
done
