#!/bin/bash
# Measurement pass behind profiles/<tag>_*, ONE gpurun call:   gpurun --timeout 2400 -- bash tools/measure.sh [tag]
#   1. pytest -m gpu
#   2. bench.py (default: configs[1] as named) -> bench.json
#   3. counter calibration: tools/micro/bin/gather_calib (known byte counts) under the TCC read-request counters
#   4. the same counters, the write counters and the SQ counters on the bench command (each --pmc pass in its own run,
#      only --kernel-trace beside it)
#   5. rocprofv3 --kernel-trace --stats of the bench command
#   6. the fetch-count build (-DBNS_COUNT_FETCHES): distinct buckets fetched and probe passes per launch
# Everything lands in gpurun_out/<tag>/; tools/summarize.py turns it into profiles/<tag>_* (and profiles/traffic.json,
# profiles/probe_traffic.json, each stamped with the sha256 of the library sources it was measured on).
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
if [ ! -x tools/micro/bin/gather_calib ]; then mkdir -p tools/micro/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/gather_calib tools/micro/gather_calib.hip 2>/dev/null; fi
if [ ! -f bonsai_amd/lib/libbonsai_amd_count.so ] || [ bonsai_amd/csrc/bns_kernels.hip -nt bonsai_amd/lib/libbonsai_amd_count.so ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBNS_COUNT_FETCHES -Iinclude bonsai_amd/csrc/bns_api.hip -o bonsai_amd/lib/libbonsai_amd_count.so 2>/dev/null; fi
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"; tail -3 "$O/pytest.log"
fi
timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; cut -c1-600 "$O/bench.json"
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "FETCH_SIZE TCC_BUBBLE_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_LEVEL_sum"
)
SQPASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM"
)
i=0
for pass in "${PASSES[@]}"; do
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/calib_pmc$i" -o calib -- tools/micro/bin/gather_calib 29 200 > "$O/calib_pmc$i.log" 2>&1
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/bench_pmc$i" -o bench -- python bench.py --no-cpu --no-probe --no-text --no-inflate --steps 2 --warmup 1 > "$O/bench_pmc$i.log" 2>&1
  i=$((i+1))
done
tools/micro/bin/gather_calib 29 200 > "$O/calib_plain.log" 2>&1
# the standalone probe kernel under the read-request counters (bench with its probe leg, 2 steps)
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$O/probe_pmc" -o bench -- python bench.py --no-cpu --no-text --no-inflate --steps 2 --warmup 1 > "$O/probe_pmc.log" 2>&1
# configs[2] (spaced, paired) bench line + its kernel-trace stats
timeout 600 python bench.py --spacing 1x15,0x15 --paired --log2-buckets 31 --no-probe --steps 10 > "$O/bench_c2.json" 2> "$O/bench_c2.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_c2" -o bench -- python bench.py --spacing 1x15,0x15 --paired --log2-buckets 31 --no-cpu --no-probe --steps 10 > "$O/bench_kt_c2.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d "$O/c2_pmc" -o bench -- python bench.py --spacing 1x15,0x15 --paired --log2-buckets 31 --no-cpu --no-probe --steps 2 --warmup 1 > "$O/c2_pmc.log" 2>&1
j=0
for pass in "${SQPASSES[@]}"; do
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$O/bench_sq$j" -o bench -- python bench.py --no-cpu --no-probe --no-text --no-inflate --steps 2 --warmup 1 > "$O/bench_sq$j.log" 2>&1
  j=$((j+1))
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o bench -- python bench.py --no-cpu --no-text > "$O/bench_kt.log" 2>&1
if [ -f bonsai_amd/lib/libbonsai_amd_count.so ]; then
  BONSAI_AMD_LIB=$PWD/bonsai_amd/lib/libbonsai_amd_count.so timeout 600 python bench.py --no-cpu --no-probe --no-text --steps 4 --warmup 1 > "$O/bench_count.json" 2> "$O/bench_count.err"
  cut -c1-200 "$O/bench_count.json"; grep -o '"debug_fetch_count.*' "$O/bench_count.json"
fi
# keep what goes back small: drop the per-dispatch traces of torch's own kernels
find "$O" -name "*_kernel_trace.csv" -size +20M -delete
du -sh "$O"
