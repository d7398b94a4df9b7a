#!/bin/bash
# round 5: a PAIR of BGZF files through the CLI (the same file as both mates' file: 2 x N mates), device path against the host reader:  tools/r05_bgzf_pair.sh [reads=64000000]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
python tools/r05_bgzf_make.py $N | tail -1
D=/tmp/bgzfbench
cp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz; cat $D/r2.bgzf.fq.gz > /dev/null
for mode in dev dev host; do
  s=$(date +%s.%N)
  if [ $mode = dev ]; then BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz 2>&1 | grep -E "pair of BGZF|process_dataset|lassified" | fold -w 230
  else BNS_TEXT_GPU=0 BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz 2>&1 | grep -E "process_dataset|lassified" | fold -w 230; fi
  e=$(date +%s.%N)
  python3 -c "print('$mode: wall %.2f s = %.1f M pairs/s = %.1f M mates/s' % ($e - $s, $N / ($e - $s) / 1e6, 2 * $N / ($e - $s) / 1e6))"
done
