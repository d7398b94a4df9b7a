#!/bin/bash
# round 6: instruction mix, wait cycles and HBM requests of the gzip-stream kernels (gz_decode_kernel, gz_search_kernel) on one call of
# tools/gz_stream_bench.py (483 MB of FASTQ text, 260 MB of DEFLATE, ~3000 chunks).  Every --pmc pass in its own run, --kernel-trace only beside it.
#   tools/pmc_gz.sh [tag=r06_gz_pmc]  -> gpurun_out/<tag>/summary.txt
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r06_gz_pmc}
O=gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/pmc$i" -o b -- python tools/gz_stream_bench.py 1500000 > "$O/pmc$i.log" 2>&1
  i=$((i+1))
done
python - "$O" <<'PY' | tee "$O/summary.txt"
import csv, glob, sys, collections
O = sys.argv[1]
TEXT = 483.0e6
for kern in ("gz_decode_kernel", "gz_search_kernel"):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for p in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    print(kern + " (per launch: one call on 260 MB of DEFLATE = 483 MB of text):")
    for c in sorted(acc):
        v = acc[c] / max(1, len(n[c]))
        print("  %-24s %.4g = %.3g per byte of text" % (c, v, v / TEXT))
PY
