#!/bin/bash
# round 6: `bonsai classify` on ONE plain gzip stream: inflated on the device (process_gz_gpu) against the host readers (BNS_GZ_GPU=0: the
# parallel host inflater pgzip; BNS_NO_PGZ=1 as well: zlib, what the reference does).  tools/r06_gz.sh [reads=64000000] [random|binned]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-64000000}; Q=${2:-random}
python tools/r06_gz_make.py $N $Q | tail -1
D=/tmp/gzbench; F=$D/r.$Q.fq.gz
BIN=bonsai_amd/bin/bonsai
run() {  # label, env...
  local label=$1; shift
  s=$(date +%s.%N)
  env BNS_CLI_TIMING=1 "$@" $BIN classify -a -K -o /dev/null -b $D/taxa.$label.bin $D/bns.db $D/nodes.dmp $F 2>&1 | grep -E "gzip text|process_dataset" | cut -c1-700
  e=$(date +%s.%N)
  python3 -c "print('$label: wall %.2f s = %.1f M reads/s' % ($e - $s, $N / ($e - $s) / 1e6))"
}
for rep in 1 2 3; do run device A=1; done
if [ "$N" -le 16000000 ]; then
  run pgzip BNS_GZ_GPU=0
  run zlib BNS_GZ_GPU=0 BNS_NO_PGZ=1
  cmp $D/taxa.device.bin $D/taxa.pgzip.bin && cmp $D/taxa.device.bin $D/taxa.zlib.bin && echo "taxa identical on all three paths"
fi
for kb in 32 128; do run device BNS_GZ_CHUNK_KB=$kb; done
for mb in 64 256; do run device BNS_GZ_PIECE_MB=$mb; done
