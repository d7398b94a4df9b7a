#!/bin/bash
# instruction mix and wait cycles of inflate_wave_kernel (one member per wavefront) on a batch of N members:  tools/pmc_inflate_wave.sh <tag> [N=1024]
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-pmc_inflate_wave}; N=${2:-1024}
O=gpurun_out/$TAG; mkdir -p "$O"
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  BNS_INFLATE_FORM=wave timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/pmc$i" -o b -- python tools/inflate_bench.py 512 $N > "$O/pmc$i.log" 2>&1
  i=$((i+1))
done
python - "$O" "$N" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]; N = int(sys.argv[2])
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for p in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "inflate_wave_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(acc):
    v = acc[c] / max(1, len(n[c]))
    print("%-24s %.4g per launch of %d members = %.4g per member = %.3g per output byte" % (c, v, N, v / N, v / N / 65280.0))
PY
