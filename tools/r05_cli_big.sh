#!/bin/bash
# round 5: `bonsai classify` on a plain FASTQ long enough for the steady state (default 256 M reads = 80 GB of text)
cd /root/repo
N=${1:-256000000}
D=/tmp/clibig; mkdir -p $D
free -g | head -2; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
python tools/make_fastq.py $N $D/r.fq
ls -l $D/r.fq
cat $D/r.fq > /dev/null
for args in "-K" "-K" ""; do
  for rd in 12 14; do
    t0=$(date +%s.%N)
    BNS_CLI_TIMING=1 BNS_TEXT_READERS=$rd bonsai_amd/bin/bonsai classify -a -p 4 $args -o $D/out.txt $D/bns.db $D/nodes.dmp $D/r.fq 2> $D/err.txt
    t1=$(date +%s.%N)
    grep -E "text on the device|process_dataset|lassified" $D/err.txt | cut -c1-420
    python3 -c "print('   ^ args [$args] readers $rd: wall %.3f s = %.1f M reads/s' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))"
  done
done
