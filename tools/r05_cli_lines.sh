#!/bin/bash
# round 5: Kraken lines at scale (256 M reads, output to /dev/null and to a file), appended to gpurun_out/r05_cli_lines.txt
cd /root/repo
OUT=gpurun_out/r05_cli_lines.txt; mkdir -p gpurun_out; : > $OUT
N=${1:-256000000}
D=/tmp/clibig; mkdir -p $D
df -h /tmp | tail -1 >> $OUT
python tools/make_fastq.py $N $D/r.fq
cat $D/r.fq > /dev/null
run() {
  tag=$1; shift
  t0=$(date +%s.%N)
  env BNS_CLI_TIMING=1 "$@" 2> $D/err.txt
  rc=$?
  t1=$(date +%s.%N)
  echo "== $tag (rc $rc)" >> $OUT
  grep -E "text on the device|process_dataset|lassified|start-up" $D/err.txt | cut -c1-700 >> $OUT
  grep -q lassified $D/err.txt || tail -5 $D/err.txt >> $OUT
  python3 -c "print('   wall %.3f s = %.1f M reads/s' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))" >> $OUT
}
B="bonsai_amd/bin/bonsai classify -a"
F="$D/bns.db $D/nodes.dmp $D/r.fq"
for p in 4 6; do for rd in 8 10; do
  run "plain FASTQ, $N reads, Kraken lines to /dev/null, -p $p, $rd readers" BNS_TEXT_READERS=$rd $B -p $p -o /dev/null $F
done; done
run "plain FASTQ, Kraken lines to a file, -p 4, 8 readers" BNS_TEXT_READERS=8 $B -p 4 -o $D/out.txt $F
ls -l $D/out.txt >> $OUT
cat $OUT
