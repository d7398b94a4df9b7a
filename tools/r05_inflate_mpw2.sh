#!/bin/bash
# round 5: is inflate_members_kernel bound by DIVERGENCE between the members of a wavefront?  1 / 2 / 4 / 8 busy lanes, both table forms
cd /root/repo
for mpw in 1 2 4 8; do for lut in 1 0; do
  echo "== MPW $mpw LUT $lut"
  BNS_INFLATE_MPW=$mpw BNS_INFLATE_LUT=$lut timeout 600 python tools/inflate_bench.py 512 1024,4096,8192 2>&1 | grep -E "members \(" | cut -c1-140
done; done
