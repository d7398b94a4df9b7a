#!/bin/bash
# where does the CLI's FASTQ pipeline stop scaling?  stage timings per setting (64 M reads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pk in 1 2 3; do for a in "-P 2" "-P 4 -p 8" "-P 3 -p 6"; do
  echo "== packers $pk  $a"; BNS_CLI_PACKERS=$pk timeout 300 python tools/cli_bench.py 64000000 $a 2>&1 | grep -E "^\[|^args" | sed 's/\x27, \x27\[timing\]/\n   /g' | grep -v "threads bound" | grep "reader:\|wait-for\|process_dataset\|^args" | cut -c1-230
done; done
