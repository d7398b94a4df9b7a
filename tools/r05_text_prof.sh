#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats of the text path (tools/text_bench.py: 6 M reads of FASTQ text through bns_classify_text, 5 calls)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${BNS_PROF_NAME:-r05_text_prof}; rm -rf $O; mkdir -p $O
for mode in taxon runs; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o text -- python tools/text_bench.py 6000000 $mode > $O/$mode.log 2>&1
  tail -1 $O/$mode.log | cut -c1-400
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode: $f"; head -25 "$f" | cut -c1-200
  cp "$f" $O/${mode}_kernel_stats.csv
  find $O/$mode -name "*_kernel_trace.csv" -delete
done
