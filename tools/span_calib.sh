#!/bin/bash
# Where should the spill threshold of the minimizer-window choice sit?  Forced windows on dbs of different density / load.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('%-28s kernel %.3f ms  m=%s spilled %.3f %% of %d keys  ovf=%s' % ('$1', d['roofline']['kernel_ms'], c['table_minimizer_m'], 100.0*c['table_spilled_keys']/max(1,c['db_keys']), c['db_keys'], c['table_overflow_keys']))"; }
for sp in 8 11 15; do
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --bucket-slots-log2 29 2>/dev/null | show "load1x span $sp"
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --bucket-slots-log2 30 2>/dev/null | show "load2x span $sp"
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --bucket-slots-log2 31 2>/dev/null | show "load4x span $sp"
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --genomes 4096 --log2-buckets 31 2>/dev/null | show "keys1e9 span $sp"
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --db-window 40 2>/dev/null | show "w40 span $sp"
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --db-window 35 --genome-len 1048576 2>/dev/null | show "w35 1Mb span $sp"
  python bench.py --no-probe --no-cpu --steps 10 --min-span $sp --db-window 50 --db-score lex 2>/dev/null | show "w50 lex span $sp"
done
