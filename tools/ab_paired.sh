#!/bin/bash
# tools/ab_paired.sh "libA.so libB.so": paired shapes (default db, every-k-mer db, spaced) + a parity sample on ragged pairs
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-12s %-34s kernel %.3f ms parity %s' % ('$1', '$2', d['roofline']['kernel_ms'], d.get('parity_sample')))"; }
for lib in $LIBS; do
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 --paired --len-dist miseq 2>/dev/null | show $lib "miseq-paired(parity)"
done
for rep in 1 2 3; do
  for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --paired 2>/dev/null | show $lib paired
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --paired --genome-len 262144 --db-window 0 2>/dev/null | show $lib paired-allkmers
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --paired --spacing 1x15,0x15 --genome-len 262144 --db-window 0 2>/dev/null | show $lib paired-spaced-allkmers
  done
done
