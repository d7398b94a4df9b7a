#!/bin/bash
# How much do the second probe passes cost?  Same reads model against dbs of 64 / 256 / 1024 genomes in the SAME 137 GB table:
# fewer minimizer groups -> fewer shared buckets -> fewer continuation passes.  Prints kernel ms and (count build) passes per read.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in "64 25" "256 27" "1024 29"; do
  set -- $spec; g=$1; lb=$2
  echo "genomes $g log2-buckets $lb"
  python bench.py --no-cpu --no-probe --genomes $g --log2-buckets $lb --bucket-slots-log2 33 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('  kernel_ms %.3f keys %d load %.4f ovf %d' % (d['roofline']['kernel_ms'], d['config']['db_keys'], d['config']['load_factor'], d['config']['table_overflow_keys']))"
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so python bench.py --no-cpu --no-probe --steps 3 --warmup 1 --genomes $g --log2-buckets $lb --bucket-slots-log2 33 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('  fetch/read %.2f passes/read %.3f' % (d['debug_fetch_count']['buckets_fetched_per_launch']/1e7, d['debug_fetch_count']['probe_passes_per_launch']/1e7))"
done
