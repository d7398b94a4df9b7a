cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
show() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('%-6s %-18s kernel %.3f ms parity %s' % ('$1', '$2', d['roofline']['kernel_ms'], d.get('parity_sample')))"; }
AK="--genome-len 262144 --db-window 0"
for rep in 1 2; do
for lib in final; do
  unset BONSAI_AMD_LIB
  python bench.py --no-probe --cpu-sample 200000 --steps 10 2>/dev/null | show $lib default
  python bench.py --no-probe --no-cpu --steps 10 --paired 2>/dev/null | show $lib paired
  python bench.py --no-probe --cpu-sample 200000 --steps 10 --bucket-slots-log2 28 2>/dev/null | show $lib load67
  python bench.py --no-probe --cpu-sample 200000 --steps 10 $AK 2>/dev/null | show $lib allkmers
  python bench.py --no-probe --cpu-sample 200000 --steps 10 $AK --bucket-slots-log2 29 2>/dev/null | show $lib allkmers-load34
  python bench.py --no-probe --no-cpu --steps 10 $AK --genomes 16384 --log2-buckets 33 2>/dev/null | show $lib allkmers-4e9
done
done
