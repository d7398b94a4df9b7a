cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1: %.1f M reads/s kernel %.3f ms frac %.3f parity %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('parity_sample')))"; }
AK="--genome-len 262144 --db-window 0"
python bench.py --no-probe --cpu-sample 400000 --paired 2>/dev/null | show paired
python bench.py --no-probe --cpu-sample 400000 --paired $AK 2>/dev/null | show paired_allkmers
python bench.py --no-probe --cpu-sample 400000 --paired --spacing 1x15,0x15 --log2-buckets 31 2>/dev/null | show c2_spaced_paired
python bench.py --no-probe --cpu-sample 400000 --paired --spacing 1x15,0x15 $AK 2>/dev/null | show c2_spaced_paired_allkmers
python bench.py --no-probe --cpu-sample 400000 --paired --len-dist miseq 2>/dev/null | show paired_miseq
