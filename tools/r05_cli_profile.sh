#!/bin/bash
# round 5: the CLI's host path on one MI355X, 16-CPU quota -> gpurun_out/r05_cli.txt (copied to profiles/r05_cli.txt)
#   plain FASTQ (256 M reads, 80 GB): -K, Kraken lines; one context and four contexts on the device
#   BGZF (96 M reads): text left on the device, against the host reader
cd /root/repo
OUT=gpurun_out/r05_cli.txt; mkdir -p gpurun_out; : > $OUT
N=${1:-256000000}
D=/tmp/clibig; mkdir -p $D
{ echo "# $(date -u) ; nproc $(nproc) ; cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; free -g | head -2; } >> $OUT
python tools/make_fastq.py $N $D/r.fq
cat $D/r.fq > /dev/null
run() {  # tag, env..., -- args
  tag=$1; shift
  t0=$(date +%s.%N)
  env BNS_CLI_TIMING=1 "$@" 2> $D/err.txt
  t1=$(date +%s.%N)
  echo "== $tag" >> $OUT
  grep -E "text on the device|process_dataset|lassified|start-up|reader:|wait-for-reader" $D/err.txt | cut -c1-700 >> $OUT
  python3 -c "print('   wall %.3f s = %.1f M reads/s' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))" >> $OUT
}
B="bonsai_amd/bin/bonsai classify -a -p 4"
F="$D/bns.db $D/nodes.dmp $D/r.fq"
for rep in 1 2; do
  run "plain FASTQ, $N reads, -K (text on the device, 14 readers)" BNS_TEXT_READERS=14 $B -K -o $D/out.txt $F
done
run "plain FASTQ, -K, default readers" $B -K -o $D/out.txt $F
run "plain FASTQ, Kraken lines (text on the device, 14 readers)" BNS_TEXT_READERS=14 $B -o $D/out.txt $F
ls -l $D/out.txt >> $OUT
run "plain FASTQ, Kraken lines, default readers" $B -o $D/out.txt $F
run "plain FASTQ, -K, -g 0,0 (two contexts on the one device)" BNS_TEXT_READERS=14 $B -K -g 0,0 -o $D/out.txt $F
run "plain FASTQ, -K, -g 0,0,0,0 (four contexts on the one device)" BNS_TEXT_READERS=14 $B -K -g 0,0,0,0 -o $D/out.txt $F
run "plain FASTQ, -K, -b taxa, text on the device" BNS_TEXT_READERS=14 $B -K -b $D/t1.bin -o $D/out.txt $F
head -c $((64000000*314)) $D/r.fq > $D/r64.fq
N=64000000
F64="$D/bns.db $D/nodes.dmp $D/r64.fq"
run "64 M reads: plain FASTQ, -K, -b taxa, text on the device" $B -K -b $D/t1.bin -o $D/out.txt $F64
run "64 M reads: plain FASTQ, -K, -b taxa, HOST parser (BNS_TEXT_GPU=0, round-4 path)" BNS_TEXT_GPU=0 $B -K -b $D/t2.bin -o $D/out.txt $F64
cmp $D/t1.bin $D/t2.bin && echo "taxa of the two paths: identical ($(stat -c %s $D/t1.bin) bytes)" >> $OUT
run "64 M reads: Kraken lines, text on the device" $B -o $D/o1.txt $F64
run "64 M reads: Kraken lines, HOST parser" BNS_TEXT_GPU=0 $B -o $D/o2.txt $F64
cmp $D/o1.txt $D/o2.txt && echo "Kraken lines of the two paths: identical ($(stat -c %s $D/o1.txt) bytes)" >> $OUT
rm -f $D/r.fq $D/r64.fq $D/o1.txt $D/o2.txt $D/out.txt
echo "# ---- BGZF" >> $OUT
timeout 1500 python tools/r05_bgzf.py 96000000 2>&1 | grep -E "==|BGZF|reads,|process_dataset|taxa equal" | cut -c1-900 >> $OUT
cat $OUT
