#!/bin/bash
# round 6: the BGZF and pair paths with one context and with several contexts ON THE ONE DEVICE there is (-g 0,0: what the chain of turns
# costs or gains when the second context has only idle CUs to use):  tools/r06_cli_multi.sh [reads=64000000]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}
python tools/r05_bgzf_make.py $N | tail -1
D=/tmp/bgzfbench
cp $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz; cat $D/r2.bgzf.fq.gz > /dev/null
if [ -z "${NO_PLAIN:-}" ]; then
python3 - <<PY
import gzip, zlib, os, struct
# the plain pair: the BGZF file's text, twice
src = "$D/r.bgzf.fq.gz"
with open("$D/p_1.fq", "wb") as o:
    d = zlib.decompressobj(31)
    with open(src, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b: break
            while b:
                o.write(d.decompress(b))
                if d.eof:
                    b = d.unused_data; d = zlib.decompressobj(31)
                else: b = b""
PY
cp $D/p_1.fq $D/p_2.fq; cat $D/p_1.fq $D/p_2.fq > /dev/null
fi
BIN=bonsai_amd/bin/bonsai
run() {  # label devices files...
  local label=$1 dev=$2; shift 2
  s=$(date +%s.%N)
  BNS_CLI_TIMING=1 $BIN classify -a -K -g $dev -o /dev/null $D/bns.db $D/nodes.dmp "$@" 2>&1 | grep -E "process_dataset" | tr '\n' ' '
  e=$(date +%s.%N)
  python3 -c "print('<- $label -g $dev: wall %.2f s = %.1f M reads(mates)/s' % ($e - $s, $N * $# / ($e - $s) / 1e6))"
}
for rep in 1 2; do
  for dev in ${DEVS:-0 0,0 0,0,0,0}; do
    run "BGZF" $dev $D/r.bgzf.fq.gz
    run "BGZF pair" $dev $D/r.bgzf.fq.gz $D/r2.bgzf.fq.gz
    [ -z "${NO_PLAIN:-}" ] && run "plain pair" $dev $D/p_1.fq $D/p_2.fq
    [ -z "${NO_PLAIN:-}" ] && run "plain" $dev $D/p_1.fq
  done
done
