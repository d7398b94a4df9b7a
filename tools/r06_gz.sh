#!/bin/bash
# round 6: `bonsai classify` on ONE plain gzip stream (and a pair of them) inflated on the device (process_gz_gpu: bns_inflate_stream_device)
# against the host readers (BNS_GZ_GPU=0: the parallel host inflater pgzip; BNS_NO_PGZ=1 as well: zlib, what the reference does), for two
# quality models (tools/r06_gz_make.py); per-kernel times of one device run.   tools/r06_gz.sh [reads=32000000]  -> gpurun_out/r06_gz.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-32000000}
D=/tmp/gzbench; BIN=bonsai_amd/bin/bonsai
mkdir -p gpurun_out
run() {  # label, input files..., then env after --
  local label=$1; shift
  local files=(); while [ "$1" != "--" ]; do files+=("$1"); shift; done; shift
  s=$(date +%s.%N)
  env BNS_CLI_TIMING=1 "$@" timeout 300 $BIN classify -a $KFLAG -o /dev/null -b $D/taxa.$label.bin $D/bns.db $D/nodes.dmp "${files[@]}" 2>&1 | grep -E "gzip text|gzip files|process_dataset" | sed -E 's/; waits:.*(; classify calls)/\1/' | cut -c1-560
  e=$(date +%s.%N)
  python3 -c "print('$label: wall %.2f s = %.1f M reads(mates)/s' % ($e - $s, $N * ${#files[@]} / ($e - $s) / 1e6))"
}
{
for Q in random binned; do
  python tools/r06_gz_make.py $N $Q | tail -1
  F=$D/r.$Q.fq.gz
  KFLAG=-K
  for rep in 1 2 3; do run device $F -- A=1; done
  run pgzip $F -- BNS_GZ_GPU=0
  if [ $Q = random ]; then run zlib $F -- BNS_GZ_GPU=0 BNS_NO_PGZ=1; cmp $D/taxa.device.bin $D/taxa.zlib.bin && echo "taxa: device = zlib"; fi
  cmp $D/taxa.device.bin $D/taxa.pgzip.bin && echo "taxa: device = pgzip"
  KFLAG=
  for rep in 1 2; do run device_lines $F -- A=1; done
  run pgzip_lines $F -- BNS_GZ_GPU=0
done
# a pair: the binned file as both mates
cp $D/r.binned.fq.gz $D/r2.binned.fq.gz; cat $D/r2.binned.fq.gz > /dev/null
KFLAG=-K
for rep in 1 2; do run pair_device $D/r.binned.fq.gz $D/r2.binned.fq.gz -- A=1; done
run pair_pgzip $D/r.binned.fq.gz $D/r2.binned.fq.gz -- BNS_GZ_GPU=0
cmp $D/taxa.pair_device.bin $D/taxa.pair_pgzip.bin && echo "taxa: pair device = pair pgzip"
} 2>&1 | tee gpurun_out/r06_gz.txt
# the kernels of one device run
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gzprof && rocprofv3 --kernel-trace --stats -d /tmp/gzprof -o gz --output-format csv -- $OLDPWD/$BIN classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.binned.fq.gz > /dev/null 2>&1)
cp /tmp/gzprof/gz_kernel_stats.csv gpurun_out/r06_gz_kernel_stats.csv 2>/dev/null
python3 - <<'PY' | tee -a gpurun_out/r06_gz.txt
import csv
print("kernels of one device run on the binned file (rocprofv3 --kernel-trace --stats):")
for r in csv.DictReader(open("gpurun_out/r06_gz_kernel_stats.csv")):
    if float(r["TotalDurationNs"]) > 2e5: print("  %-44s calls %5s total %8.2f ms avg %9.1f us" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
