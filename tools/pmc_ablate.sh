#!/bin/bash
# instruction counts + kernel time of classify_kernel under ablation bits (profiling aid; builds an ablation library on the box)
# bits: 1 no probe, 2 no vote, 4 no minimizer window, 8 no stores, 16 no resolve
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
BNS_ABLATION=1 python -c "from bonsai_amd.build import build_device_library as b; b(force=True)" > /dev/null   # restore with python -m bonsai_amd.build
for ab in 0 1 2 3 5 7; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmca_$ab -o b -- python bench.py --no-cpu --steps 2 --warmup 1 --ablate $ab > /dev/null 2>&1
  python - <<PY
import csv, collections
agg=collections.defaultdict(float); disp=set()
for r in csv.DictReader(open("gpurun_out/pmca_$ab/b_counter_collection.csv")):
    if "classify_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
print("ablate=$ab per read:", {k.replace("SQ_",""): round(v/len(disp)/1e7,1) for k,v in sorted(agg.items())})
PY
  python bench.py --no-cpu --steps 10 --ablate $ab 2>&1 | tail -1 | python tools/_ab_line.py "ablate=$ab"
done
