cd $GRAFT_REPO_ROOT
for lib in S32w6 S24w7 S40w5 S48w4; do for m in 14 15; do
BNS_SPACED_M=$m BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/lib$lib.so python bench.py --spacing 1x15,0x15 --paired --steps 10 --no-probe --cpu-sample 100000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib m=$m: %.1f M reads/s kernel %.3f ms ovf %d parity %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['table_overflow_keys'], d.get('parity_sample',{}).get('mismatches')))"
done; done
