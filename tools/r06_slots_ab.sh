#!/bin/bash
# round 6: the page-locked slots of the BGZF path from hipHostMalloc (BNS_PIN_MALLOC=1, as before) against registered huge-page memory of
# our own (tools/micro/pin_bench.hip: 2 ms per 96 MiB instead of 15-45):  tools/r06_slots_ab.sh [reads=64000000]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
python tools/r05_bgzf_make.py $N | tail -1
D=/tmp/bgzfbench
cp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz; cat $D/r2.bgzf.fq.gz > /dev/null
run() {  # label files...
  local label=$1; shift
  local mode="-K"; [ -n "${KRAKEN:-}" ] && mode="-p6"
  s=$(date +%s.%N)
  BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a $mode -o /dev/null $D/bns.db $D/nodes.dmp "$@" 2>&1 | grep -E "process_dataset|page-lock" | sed -E 's/.*(page-lock [0-9.]+ (\(summed\)|s)).*/\1/' | tr '\n' ' '
  e=$(date +%s.%N)
  python3 -c "print('<- $label: wall %.2f s = %.1f M reads(mates)/s' % ($e - $s, $N * $# / ($e - $s) / 1e6))"
}
if [ -z "${NO_PLAIN:-}" ]; then
python3 - <<PY
import zlib
with open("$D/p_1.fq", "wb") as o, open("$D/r.bgzf.fq.gz", "rb") as f:
    d = zlib.decompressobj(31)
    while True:
        b = f.read(1 << 24)
        if not b: break
        while b:
            o.write(d.decompress(b))
            if d.eof: b = d.unused_data; d = zlib.decompressobj(31)
            else: b = b""
PY
cp $D/p_1.fq $D/p_2.fq; cat $D/p_1.fq $D/p_2.fq $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz > /dev/null
fi
for rep in 1 2 3; do
  BNS_PIN_MALLOC=1 run "hipHostMalloc, BGZF" $D/r.bgzf.fq.gz
  run "registered, BGZF" $D/r.bgzf.fq.gz
  BNS_PIN_MALLOC=1 run "hipHostMalloc, BGZF pair" $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
  run "registered, BGZF pair" $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
  if [ -z "${NO_PLAIN:-}" ]; then
    BNS_PIN_MALLOC=1 run "hipHostMalloc, plain" $D/p_1.fq
    run "registered, plain" $D/p_1.fq
    BNS_PIN_MALLOC=1 run "hipHostMalloc, plain pair" $D/p_1.fq $D/p_2.fq
    run "registered, plain pair" $D/p_1.fq $D/p_2.fq
    KRAKEN=1 BNS_PIN_MALLOC=1 run "hipHostMalloc, plain, Kraken lines" $D/p_1.fq
    KRAKEN=1 run "registered, plain, Kraken lines" $D/p_1.fq
  fi
done
