cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/r06_gz_make.py ${1:-32000000} binned | tail -1
D=/tmp/gzbench
for i in 1 2; do
BNS_CLI_TIMING=1 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.binned.fq.gz 2>&1 | grep -E "gzip text|process_dataset|since start|start-up" | cut -c1-1100
done
