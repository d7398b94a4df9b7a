#!/bin/bash
# round 5: plain-text device path, blocks of 128 against 64 MiB, interleaved on one box:  tools/r05_block_ab.sh [reads=256000000] [reps=3]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-256000000}; REPS=${2:-3}
D=/tmp/clibig; mkdir -p $D
python tools/make_fastq.py $N $D/r.fq; cat $D/r.fq > /dev/null
for rep in $(seq $REPS); do
  for mb in 128 64; do
    for args in "-K" ""; do
      t0=$(date +%s.%N)
      BNS_TEXT_BLOCK_MB=$mb BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -p 4 $args -o /dev/null $D/bns.db $D/nodes.dmp $D/r.fq 2> $D/err.txt
      t1=$(date +%s.%N)
      python3 -c "import re;e=open('$D/err.txt').read();m=re.search(r'process_dataset ([0-9.]+)',e);print('blocks %3d MiB args [%-2s] process_dataset %s s, wall %.3f s = %.1f M reads/s' % ($mb, '$args', m.group(1) if m else '?', $t1-$t0, $N/($t1-$t0)/1e6))"
    done
  done
done
