#!/bin/bash
# instruction mix and wave-time split of the spaced classify instantiation (configs[2])
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmc_spaced; rm -rf $O; mkdir -p $O
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum"; do
  n=$(echo $pass | cut -d" " -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/$n -o b -- python bench.py --spacing 1x15,0x15 --paired --no-cpu --no-probe --steps 2 --warmup 1 > $O/$n.log 2>&1
done
python - <<'PY'
import csv, collections, glob
a=collections.defaultdict(float); d=set()
for p in glob.glob("gpurun_out/pmc_spaced/*/b_counter_collection.csv"):
    disp=set()
    for r in csv.DictReader(open(p)):
        if "classify_kernel" in r["Kernel_Name"]:
            a[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    for c in list(a):
        pass
    n=len(disp)
    print(p.split("/")[-2], "launches", n)
    globals().setdefault("N",{})[p]=n
rounds=5e6*2*2   # 5M pairs x 2 mates x 2 rounds (105 k-mers)
n=3
for c,v in sorted(a.items()):
    print("%-24s %.4g per launch  %.1f per round-wave" % (c, v/n, v/n/rounds))
PY
