#!/bin/bash
# CLI text path sweep (round 5): readers / block / piece on the 64 M-read FASTQ of tools/cli_bench.py
cd /root/repo
python tools/cli_bench.py 64000000 > /dev/null 2>&1   # (makes the file; first run warms the page cache)
for cfg in "8 128 64" "12 128 64" "14 128 64" "12 256 64" "12 128 32" "12 64 32" "14 256 32"; do
  set -- $cfg
  echo "== readers $1 block $2 MiB piece $3 MiB"
  BNS_TEXT_READERS=$1 BNS_TEXT_BLOCK_MB=$2 BNS_TEXT_PIECE_MB=$3 python tools/cli_bench.py 64000000 2>&1 | grep -E "text on the device|args" | sed -e 's/.*text on the device: //' | cut -c1-260
done
