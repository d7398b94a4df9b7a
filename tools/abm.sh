#!/bin/bash
# tools/abm.sh "libA.so libB.so ..." : parity + timing on the default db and on the every-k-mer db, with the table's spill counts
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1
AK="--genome-len 262144 --db-window 0"
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('%-12s %-22s kernel %.3f ms  m=%s spilled %d ovf %s parity %s' % ('$1', '$2', d['roofline']['kernel_ms'], c['table_minimizer_m'], c['table_spilled_keys'], c['table_overflow_keys'], d.get('parity_sample')))"; }
for lib in $LIBS; do
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 2>/dev/null | show $lib "default(parity)"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 $AK 2>/dev/null | show $lib "allkmers(parity)"
done
for rep in 1 2 3; do
  for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe 2>/dev/null | show $lib default
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe $AK 2>/dev/null | show $lib allkmers
  done
done
