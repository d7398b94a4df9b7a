#!/bin/bash
# round 6: what `bonsai classify` spends before its first read when the db has the benchmark's size (configs[1]: 2.25e8 keys in 2^29 khash
# buckets = 6.6 GB of bns.db; the CLI benchmarks of rounds 3-5 ran against a db of six small genomes):  tools/r06_db_load.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=/tmp/bigdb; mkdir -p $D
python bench.py --save-db $D --steps 1 --warmup 1 --reads 1000000 --no-cpu --no-probe --no-text --no-inflate 2>/dev/null | cut -c1-120
ls -l $D | cut -c1-90
mkdir -p /tmp/rdsmall; python tools/make_fastq.py 4000000 /tmp/rdsmall/r.fq > /dev/null 2>&1      # (reads of another world: all that matters here is what happens before them)
cat $D/bns.db > /dev/null
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp /tmp/rdsmall/r.fq 2>&1 | grep -E "start-up|process_dataset|lassified" | cut -c1-300
  t1=$(date +%s.%N)
  python3 -c "print('   wall %.3f s' % ($t1 - $t0))"
done
