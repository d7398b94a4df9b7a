#!/bin/bash
# per-read instruction mix of classify_kernel for a bench configuration:  tools/pmc_insts2.sh <tag> [bench args]
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/pmc$i" -o b -- python bench.py --no-cpu --no-probe --steps 2 --warmup 1 "$@" > "$O/pmc$i.log" 2>&1
  i=$((i+1))
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for p in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "classify_" in r["Kernel_Name"] and "overflow" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(acc):
    print("%-24s %.4g per launch" % (c, acc[c] / max(1, len(n[c]))))
PY
