#!/bin/bash
# quick PMC passes for classify_kernel (profiling aid).  usage: tools/pmc_quick.sh <tag> [bench args]
TAG=${1:-q}; shift || true
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pass in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
            "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_BUSY_CU_CYCLES"; do
  n=$(echo $pass | cut -d" " -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/pmcq_${TAG}_$n -o b -- python bench.py --no-cpu --steps 2 --warmup 1 "$@" > /dev/null 2>&1
done
python - <<PY
import csv, collections, glob
for d in sorted(glob.glob("gpurun_out/pmcq_${TAG}_*")):
    agg=collections.defaultdict(float); disp=set()
    for r in csv.DictReader(open(d+"/b_counter_collection.csv")):
        if "classify_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    print({k: "%.3g"%(v/max(1,len(disp))) for k,v in agg.items()})
PY
