#!/bin/bash
# round 6: per-kernel times of `bonsai classify` on one plain gzip stream (rocprofv3 --kernel-trace --stats) -> gpurun_out/r06_gz_kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
python tools/r06_gz_make.py ${1:-16000000} binned | tail -1
D=/tmp/gzbench
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gzprof
BNS_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats -d /tmp/gzprof -o gz --output-format csv -- $R/bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.binned.fq.gz > /tmp/gzprof.log 2>&1
F=$(find /tmp/gzprof -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/r06_gz_kernel_stats.csv
python3 - "$F" <<'PY'
import csv, sys
print("kernels of `bonsai classify -K` on a 16 M-read gzip FASTQ (qualities in eight bins):")
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 2e5: print("  %-44s calls %5s total %8.2f ms avg %9.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
