#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03g; mkdir -p $O
BNS_DUMP_OVF=$PWD/$O/ovf.npy BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 900 python bench.py --no-cpu --no-probe --steps 3 --warmup 1 --genome-len 262144 --db-window 0 --table-buckets 67000000 > $O/a.json 2>$O/a.err
python -c "
import json
d=json.loads([l for l in open('$O/a.json') if l.startswith('{')][-1]); print(d['debug_fetch_count'])"
