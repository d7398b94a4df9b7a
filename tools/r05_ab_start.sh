#!/bin/bash
# round 5: same-box A/B of two CLI binaries (bonsai_prev = the commit before the start-up changes) on a plain and a BGZF file
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
D=/tmp/clibig; mkdir -p $D
python tools/make_fastq.py $N $D/r.fq; cat $D/r.fq > /dev/null
true
G=/tmp/bgzfbench
for rep in 1 2 3 4 5 6; do
  for b in bonsai_prev bonsai; do
    for f in "$D/bns.db $D/nodes.dmp $D/r.fq"; do
      t0=$(date +%s.%N)
      BNS_CLI_TIMING=1 bonsai_amd/bin/$b classify -a -p 4 -K -o /dev/null $f 2> /tmp/err.txt
      t1=$(date +%s.%N)
      python3 -c "import re;e=open('/tmp/err.txt').read();m=re.search(r'process_dataset ([0-9.]+)',e);print('%-12s %-16s process_dataset %s s, wall %.3f s' % ('$b', '${f##*/}', m.group(1) if m else '?', $t1-$t0))"
    done
  done
done
