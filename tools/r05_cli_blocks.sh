#!/bin/bash
# round 5: block size of the plain-text device path (BNS_TEXT_BLOCK_MB) against start-up: process_dataset and wall for N reads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
D=/tmp/clibig; mkdir -p $D
python tools/make_fastq.py $N $D/r.fq
cat $D/r.fq > /dev/null
for mb in ${SIZES:-128 64 32 16}; do
  for rep in 1 2; do
    t0=$(date +%s.%N)
    BNS_TEXT_BLOCK_MB=$mb BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -p 4 -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.fq 2> $D/err.txt
    t1=$(date +%s.%N)
    grep -E "process_dataset" $D/err.txt | tr '\n' ' '
    python3 -c "print('  blocks of $mb MiB: wall %.3f s = %.1f M reads/s' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))"
  done
  grep -E "text on the device" $D/err.txt | cut -c1-330
done
