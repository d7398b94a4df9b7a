#!/bin/bash
# Round-4 pass "every BASELINE config's per-GPU workload on the one GPU there is" (VERDICT r03 next #1), ONE gpurun call:
#   gpurun --timeout 2700 -- bash tools/r04_item1.sh [tag]
#   1. the db-shape scale tests (configs[1] / configs[2] shapes at >= 1e8 keys)
#   2. configs[3] per-rank slice: rank 0 and rank 7 of an 8-rank, 1e9-read strong-scaling job (1.25e8 reads per launch) against
#      a RefSeq-scale minimizer db (2.2e9 keys) and against the 8e9-key every-k-mer db (streamed load)
#   3. configs[4] long-read leg: classify of 1e5 x 10 kb reads, RollingHasher over 10 kb reads (kernel times from rocprofv3)
#   4. `bonsai classify -g 0,0,0,0` vs `-g 0` on a 64 M-read FASTQ: does ONE reader / formatter feed more than one context?
set -u
TAG=${1:-r04item1}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
STEP=${STEPS:-1234}
if [[ $STEP == *1* ]]; then
  timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -x -k "config1 or config2" > "$O/pytest_shapes.log" 2>&1; echo "shape tests rc=$?"; tail -3 "$O/pytest_shapes.log"
fi
run() { name=$1; shift; timeout 1500 python bench.py --no-probe "$@" > "$O/$name.json" 2> "$O/$name.err"; echo "$name rc=$?"; python tools/_line.py "$O/$name.json"; grep "bench.py:" "$O/$name.err" | head -3; }
if [[ $STEP == *2* ]]; then
  S3="--scaling strong --total-reads 1000000000 --world 8 --steps 3 --warmup 1 --cpu-sample 200000"
  run c3_w50_2e9_r0 --genomes 10240 --log2-buckets 32 $S3 --emulate-rank 0
  run c3_w50_2e9_r7 --genomes 10240 --log2-buckets 32 $S3 --emulate-rank 7
  run c3_allk_8e9_r0 --genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load $S3 --emulate-rank 0
  run c3_allk_8e9_r7 --genomes 36000 --genome-len 262144 --db-window 0 --log2-buckets 34 --stream-load $S3 --emulate-rank 7
fi
if [[ $STEP == *3* ]]; then
  run c4_len10k --read-len 10000 --reads 100000 --steps 10 --cpu-sample 20000
  run c4_len10k_allk --read-len 10000 --reads 100000 --steps 10 --cpu-sample 20000 --genome-len 262144 --db-window 0
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_len10k" -o b -- python bench.py --no-probe --no-cpu --read-len 10000 --reads 100000 --steps 10 > "$O/kt_len10k.log" 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_rolling" -o b -- python tools/rolling_bench.py 20000 > "$O/rolling.log" 2>&1; grep -v amdgpu "$O/rolling.log" | tail -4
  find "$O" -name "*_kernel_trace.csv" -size +5M -delete
fi
if [[ $STEP == *4* ]]; then
  for g in 0 0,0 0,0,0,0; do
    echo "== -g $g" >> "$O/cli.txt"
    timeout 900 python tools/cli_bench.py 64000000 -g $g >> "$O/cli.txt" 2>&1
  done
  grep -c "M reads/s" "$O/cli.txt"; grep "M reads/s" "$O/cli.txt" | cut -c1-160
fi
du -sh "$O"
