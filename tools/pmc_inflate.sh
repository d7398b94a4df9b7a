#!/bin/bash
# instruction mix and wait cycles of inflate_members_kernel on a 4096-member batch (tools/inflate_bench.py):  tools/pmc_inflate.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc_inflate}
O=gpurun_out/$TAG; mkdir -p "$O"
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/pmc$i" -o b -- python tools/inflate_bench.py 512 4096 > "$O/pmc$i.log" 2>&1
  i=$((i+1))
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for p in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "inflate_members_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
sym = 4096 * 65280.0                                   # output bytes ~ symbols (FASTQ text with few matches)
for c in sorted(acc):
    v = acc[c] / max(1, len(n[c]))
    print("%-24s %.4g per launch of 4096 members = %.3g per output byte" % (c, v, v / sym))
PY
