#!/bin/bash
# tools/ab_all.sh "libA.so libB.so": parity on five shapes (N-rich reads included), timing on six
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-12s %-26s kernel %.3f ms parity %s' % ('$1', '$2', d['roofline']['kernel_ms'], d.get('parity_sample')))"; }
AK="--genome-len 262144 --db-window 0"
for lib in $LIBS; do
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 2>/dev/null | show $lib "default(parity)"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 --paired --len-dist miseq 2>/dev/null | show $lib "miseq-paired(parity)"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 --read-len 700 --reads 2000000 2>/dev/null | show $lib "700bp(parity)"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 $AK --spacing 1x15,0x15 --paired 2>/dev/null | show $lib "spaced-paired(parity)"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 $AK --k 21 2>/dev/null | show $lib "k21(parity)"
done
for rep in 1 2 3; do
  for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe 2>/dev/null | show $lib default
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --paired 2>/dev/null | show $lib paired
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --len-dist hiseq 2>/dev/null | show $lib hiseq
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --len-dist miseq 2>/dev/null | show $lib miseq
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe $AK 2>/dev/null | show $lib allkmers
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe $AK --spacing 1x15,0x15 --paired 2>/dev/null | show $lib spaced-paired
  done
done
