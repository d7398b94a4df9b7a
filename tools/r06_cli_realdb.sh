#!/bin/bash
# round 6: `bonsai classify` end to end against a db of the BENCHMARK's size (configs[1]: 2.25e8 keys, 6.6 GB of bns.db, a 34.5 GB table -- the CLI
# runs of rounds 3-5 used a db of six small genomes, which the L2 holds): bench.py writes its db, 10 M of its reads as FASTQ and what its kernel
# says about them; the CLI's -b file must be identical; then the same reads x REP as one long file for the rate.  tools/r06_cli_realdb.sh [rep=12]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REP=${1:-12}
D=/tmp/realdb; mkdir -p $D
python bench.py --save-db $D --save-reads 10000000 --steps 2 --warmup 1 --no-cpu --no-probe --no-text --no-inflate 2>/dev/null | cut -c1-160
ls -l $D | cut -c1-100
cat $D/bns.db $D/reads.fq > /dev/null
BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -b $D/cli.u32 -o /dev/null $D/bns.db $D/nodes.dmp $D/reads.fq 2>&1 | grep -E "start-up|process_dataset|lassified" | cut -c1-260
cmp $D/cli.u32 $D/taxa.u32 && echo "CLI taxa == bench kernel taxa on 10 M reads"
rm -f $D/long.fq; for i in $(seq $REP); do cat $D/reads.fq >> $D/long.fq; done
N=$((10000000 * REP))
python tools/r05_bgzf_make.py 1000 > /dev/null 2>&1
python3 - <<PY
# the long file as BGZF, too (members of 65280 text bytes, zlib level 6, 16 processes)
import os, struct, zlib
from multiprocessing import Pool
def member(chunk):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); body = co.compress(chunk) + co.flush(); bsize = 12 + 6 + len(body) + 8 - 1
    return b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
with open("$D/reads.fq", "rb") as f, open("$D/reads.bgzf.fq.gz", "wb") as o, Pool(16) as p:
    data = f.read()
    o.write(b"".join(p.map(member, [data[i:i + 65280] for i in range(0, len(data), 65280)], chunksize=64)))
    o.write(member(b""))
PY
rm -f $D/long.bgzf.fq.gz
python3 - <<PY
# (a BGZF file is a concatenation of members: REP copies of the one file, its empty end member dropped from all but the last)
d = open("$D/reads.bgzf.fq.gz", "rb").read()
empty = 28
with open("$D/long.bgzf.fq.gz", "wb") as o:
    for i in range($REP): o.write(d[:-empty])
    o.write(d[-empty:])
PY
cat $D/long.fq $D/long.bgzf.fq.gz > /dev/null
run() {
  local label=$1; shift
  t0=$(date +%s.%N)
  BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a "$@" 2>&1 | grep -E "start-up|process_dataset|lassified" | cut -c1-230
  t1=$(date +%s.%N)
  python3 -c "print('   ^ $label: wall %.3f s = %.1f M reads/s' % ($t1 - $t0, $N / ($t1 - $t0) / 1e6))"
}
for rep in 1 2; do
  run "plain FASTQ $N reads, -K" -K -o /dev/null $D/bns.db $D/nodes.dmp $D/long.fq
  run "plain FASTQ $N reads, Kraken lines to /dev/null" -p 6 -o /dev/null $D/bns.db $D/nodes.dmp $D/long.fq
  run "BGZF $N reads, -K" -K -o /dev/null $D/bns.db $D/nodes.dmp $D/long.bgzf.fq.gz
done
