#!/bin/bash
# instruction mix per read of classify_kernel for the given bench arguments
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmc_insts; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/a -o b -- python bench.py --no-cpu --no-probe --steps 2 --warmup 1 "$@" > $O/a.log 2>&1
python - <<'PY'
import csv, collections
a=collections.defaultdict(float); disp=set()
for r in csv.DictReader(open("gpurun_out/pmc_insts/a/b_counter_collection.csv")):
    if "classify_kernel" in r["Kernel_Name"]:
        a[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
n=len(disp)
for c,v in sorted(a.items()): print("%-18s %.1f per read" % (c, v/n/1e7))
PY
