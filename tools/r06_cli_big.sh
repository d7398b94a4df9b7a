#!/bin/bash
# round 6: `bonsai classify` on a plain FASTQ long enough for the steady state (default 256 M reads = 80 GB of text), page-locked buffers from
# hipHostMalloc (BNS_PIN_MALLOC=1, rounds 3-5) against registered memory of our own (round 6): -K, and Kraken lines to /dev/null
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-256000000}
D=/tmp/clibig; mkdir -p $D
python tools/make_fastq.py $N $D/r.fq | tail -1
ls -l $D/r.fq | cut -c1-80
cat $D/r.fq > /dev/null
run() {  # label [env...] -- args
  local label=$1; shift
  t0=$(date +%s.%N)
  env BNS_CLI_TIMING=1 "$@" 2> $D/err.txt
  t1=$(date +%s.%N)
  grep -E "text on the device|process_dataset" $D/err.txt | sed -E 's/; callers waited.*//' | cut -c1-330
  python3 -c "print('   ^ $label: wall %.3f s = %.1f M reads/s' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))"
}
B="bonsai_amd/bin/bonsai classify -a"
F="$D/bns.db $D/nodes.dmp $D/r.fq"
for rep in 1 2; do
  run "-K, hipHostMalloc buffers" BNS_PIN_MALLOC=1 $B -K -o /dev/null $F
  run "-K, registered buffers" $B -K -o /dev/null $F
  run "Kraken lines to /dev/null -p 6, hipHostMalloc buffers" BNS_PIN_MALLOC=1 $B -p 6 -o /dev/null $F
  run "Kraken lines to /dev/null -p 6, registered buffers" $B -p 6 -o /dev/null $F
done
