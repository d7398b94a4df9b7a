cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1: %.1f M reads/s kernel %.3f ms frac %.3f m=%s spilled=%s ovf=%s parity %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['table_minimizer_m'], d['config']['table_spilled_keys'], d['config']['table_overflow_keys'], d.get('parity_sample')))"; }
python bench.py --no-probe --cpu-sample 200000 2>/dev/null | show default
python bench.py --no-probe --no-cpu --min-span 8 2>/dev/null | show default_span8
python bench.py --no-probe --no-cpu --min-span 11 2>/dev/null | show default_span11
python bench.py --no-probe --no-cpu --min-span 15 2>/dev/null | show default_span15
python bench.py --no-probe --cpu-sample 200000 --genome-len 262144 --db-window 0 2>/dev/null | show allkmers
python bench.py --no-probe --no-cpu --genome-len 262144 --db-window 0 --min-span 11 2>/dev/null | show allkmers_span11
python bench.py --no-probe --cpu-sample 200000 --paired 2>/dev/null | show paired
python bench.py --no-probe --cpu-sample 200000 --len-dist hiseq 2>/dev/null | show hiseq
python bench.py --no-probe --cpu-sample 200000 --len-dist miseq 2>/dev/null | show miseq
python bench.py --no-probe --cpu-sample 200000 --k 21 2>/dev/null | show k21
python bench.py --no-probe --cpu-sample 200000 --k 27 2>/dev/null | show k27
python bench.py --no-probe --no-cpu --bucket-slots-log2 29 2>/dev/null | show load1x
python bench.py --no-probe --no-cpu --genomes 4096 --log2-buckets 31 2>/dev/null | show keys1e9
