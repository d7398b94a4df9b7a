#!/bin/bash
# round 5: inflate_members_kernel with 8 / 16 / 32 / 64 busy lanes per wavefront (BNS_INFLATE_MPW), kernel GB/s of text by batch size
cd /root/repo
for mpw in 8 16 32 64; do
  for lut in 0; do
    echo "== MPW $mpw (tables without the direct part)"
    BNS_INFLATE_MPW=$mpw BNS_INFLATE_LUT=$lut timeout 600 python tools/inflate_bench.py 512 4096,8192,16384,32768 2>&1 | grep -E "members \(" | cut -c1-200
  done
done
echo "== default (8 lanes, direct tables up to 10240 members)"
timeout 600 python tools/inflate_bench.py 512 4096,8192,16384,32768 2>&1 | grep -E "members|handles" | cut -c1-200
