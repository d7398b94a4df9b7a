#!/bin/bash
# round 5: the BGZF device path of the CLI, -K only, whole timing line (file made by tools/r05_bgzf_make.py):  tools/r05_bgzf_k.sh [reads=64000000] [runs=2]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}; RUNS=${2:-2}
python tools/r05_bgzf_make.py $N | tail -1
D=/tmp/bgzfbench
for r in $(seq $RUNS); do
  s=$(date +%s.%N)
  BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.bgzf.fq.gz 2>&1 | grep -E "BGZF text|process_dataset" | fold -w 220
  e=$(date +%s.%N)
  python3 -c "print('wall %.2f s = %.1f M reads/s' % ($e - $s, $N / ($e - $s) / 1e6))"
done
