#!/bin/bash
# CLI pipeline after the second packer and the second formatter: tests, a short CLI fuzz, then stage timings (64 M reads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ref_golden.py -m gpu -x -q -k "cli or CLI" 2>&1 | tail -3
timeout 200 python tools/fuzz_cli.py 150 2>&1 | tail -2
for a in "-P 2" "-P 3 -p 6" "--paired"; do
  echo "== $a"; timeout 300 python tools/cli_bench.py 64000000 $a 2>&1 | grep -E "^\[|^args" | sed 's/\x27, \x27\[timing\]/\n   /g' | grep -v "threads bound" | grep "reader:\|wait-for\|process_dataset\|^args" | cut -c1-230
done
