cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/r06_gz_make.py ${1:-32000000} binned | tail -1
D=/tmp/gzbench
for w in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
BNS_CLI_TIMING=1 timeout 60 bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.binned.fq.gz 2>&1 | grep -E "took|workspaces|process_dataset" | tr '\n' ' '; echo
done
echo with Kraken lines
for w in 1 2 3 4; do
BNS_CLI_TIMING=1 timeout 60 bonsai_amd/bin/bonsai classify -a -o /dev/null $D/bns.db $D/nodes.dmp $D/r.binned.fq.gz 2>&1 | grep -E "took|workspaces|process_dataset" | tr '\n' ' '; echo
done
