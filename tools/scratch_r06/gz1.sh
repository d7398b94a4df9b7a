cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/r06_gz_make.py 8000000 random | tail -1
D=/tmp/gzbench
BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.random.fq.gz 2>&1 | grep -E "gzip text" | cut -c1-900
