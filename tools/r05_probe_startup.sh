#!/bin/bash
cd /root/repo
D=/tmp/clibig; mkdir -p $D
python tools/make_fastq.py 32000000 $D/r.fq
cat $D/r.fq > /dev/null
for i in 1 2 3; do
  t0=$(date +%s.%N)
  BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.fq 2> $D/err.txt
  t1=$(date +%s.%N)
  grep -E "start-up|process_dataset|since start|text on the device" $D/err.txt | cut -c1-330
  python3 -c "print('   wall %.3f s' % ($t1 - $t0))"
done
echo "--- trivial HIP process: time to first hipMalloc"
cat > /tmp/hipinit.cpp <<'EOC'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main(){ auto t0=std::chrono::steady_clock::now(); void*p; hipMalloc(&p,1<<20); auto t1=std::chrono::steady_clock::now(); hipStream_t s; hipStreamCreate(&s); void*h; hipHostMalloc(&h,64<<20,0); auto t2=std::chrono::steady_clock::now();
 printf("first hipMalloc %.3f s, stream + 64 MiB pinned %.3f s\n", std::chrono::duration<double>(t1-t0).count(), std::chrono::duration<double>(t2-t1).count()); return 0; }
EOC
/opt/rocm/bin/hipcc -O2 -o /tmp/hipinit /tmp/hipinit.cpp 2>/dev/null && for i in 1 2; do t0=$(date +%s.%N); /tmp/hipinit; t1=$(date +%s.%N); python3 -c "print('   wall %.3f s' % ($t1 - $t0))"; done
