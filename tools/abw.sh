#!/bin/bash
# tools/abw.sh "libA.so libB.so ...": several builds on the default db in several tables / shapes (the loader chooses the window)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS=$1
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('%-12s %-18s kernel %.3f ms  m=%s spilled %.3f %% ovf %s parity %s' % ('$1', '$2', d['roofline']['kernel_ms'], c['table_minimizer_m'], 100.0*c['table_spilled_keys']/max(1,c['db_keys']), c['table_overflow_keys'], d.get('parity_sample')))"; }
for lib in $LIBS; do
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 2>/dev/null | show $lib "default(parity)"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --steps 3 --warmup 1 --no-probe --cpu-sample 200000 --paired --len-dist miseq 2>/dev/null | show $lib "miseq-paired(parity)"
done
for rep in 1 2; do
  for lib in $LIBS; do
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe 2>/dev/null | show $lib default
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --paired 2>/dev/null | show $lib paired
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --len-dist hiseq 2>/dev/null | show $lib hiseq
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --bucket-slots-log2 30 2>/dev/null | show $lib load2x
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --bucket-slots-log2 29 2>/dev/null | show $lib load1x
    BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/$lib python bench.py --no-cpu --no-probe --genomes 4096 --log2-buckets 31 2>/dev/null | show $lib keys1e9
  done
done
