#!/bin/bash
# Marginal cost of one more instruction per round in classify_kernel: builds padded variants HERE (no GPU needed),
#   tools/pad.sh build      -> gpurun_scratch/pad_*.so
# and times them inside one gpurun call:
#   gpurun -- 'bash tools/pad.sh run'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ "$1" = build ]; then
  mkdir -p gpurun_scratch
  for v in base BNS_PAD_VALU=64 BNS_PAD_VFAST=64 BNS_PAD_SALU=64 BNS_PAD_LDS=16; do
    if [ $v = base ]; then BNS_EXTRA_DEFINES="" python -m bonsai_amd.build > /dev/null; else BNS_EXTRA_DEFINES="$v" python -m bonsai_amd.build > /dev/null; fi
    cp bonsai_amd/lib/libbonsai_amd.so gpurun_scratch/pad_${v%%=*}.so
  done
  python -m bonsai_amd.build > /dev/null
else
  for rep in 1 2; do
    for v in base BNS_PAD_VALU BNS_PAD_VFAST BNS_PAD_SALU BNS_PAD_LDS; do
      BONSAI_AMD_LIB=$PWD/gpurun_scratch/pad_$v.so python bench.py --no-cpu --steps 20 2>&1 | tail -1 | python tools/_ab_line.py pad_$v
    done
  done
fi
