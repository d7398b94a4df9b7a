#!/bin/bash
# kernel time vs read length (rounds of 64 k-mers): separates per-unit from per-round cost.  usage: tools/lenscan.sh [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in 62 94 126 150 158 222 286 350; do
  python bench.py --no-cpu --steps 10 --read-len $L "$@" 2>&1 | tail -1 | python tools/_ab_line.py "len_$L"
done
