#!/bin/bash
# instruction mix per read under the ablation switches of a -DBNS_ABLATION build (results are wrong by design)
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for ab in 0 1 2 3 4 7; do
  O=gpurun_out/pmc_abl_$ab; rm -rf $O; mkdir -p $O
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libAbl.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O -o b -- python bench.py --no-cpu --no-probe --steps 2 --warmup 1 --ablate $ab > $O/log 2>&1
  python - $ab <<'PY'
import csv, collections, sys
ab=sys.argv[1]
a=collections.defaultdict(float); disp=set(); dur=[]
for r in csv.DictReader(open("gpurun_out/pmc_abl_%s/b_counter_collection.csv" % ab)):
    if "classify_kernel" in r["Kernel_Name"]:
        a[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
for r in csv.DictReader(open("gpurun_out/pmc_abl_%s/b_kernel_trace.csv" % ab)):
    if "classify_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
n=len(disp)
print("ablate %s: " % ab + "  ".join("%s %.1f" % (c[9:], v/n/1e7) for c,v in sorted(a.items())) + "   ms %.2f" % (sum(dur)/len(dur)))
PY
done
