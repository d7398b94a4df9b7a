#!/bin/bash
# round 5: the plain PAIR path, block size (BNS_TEXT_BLOCK_MB), interleaved on one box: 2 x N/2 mates made by tools/cli_bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-128000000}
python tools/cli_bench.py $N --paired > /dev/null 2>&1
D=/tmp/clibench
for rep in 1 2 3; do
  for mb in 96 48 32; do
    t0=$(date +%s.%N)
    BNS_TEXT_BLOCK_MB=$mb BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -p 4 -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r_1.fq $D/r_2.fq 2> $D/err.txt
    t1=$(date +%s.%N)
    python3 -c "import re;e=open('$D/err.txt').read();m=re.search(r'process_dataset ([0-9.]+)',e);print('pair, blocks %3d MiB: process_dataset %s s, wall %.3f s = %.1f M mates/s%s' % ($mb, m.group(1) if m else '?', $t1-$t0, $N/($t1-$t0)/1e6, '  HANDED BACK' if 'host parser takes the rest' in e else ''))"
  done
done
