cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/r06_gz_make.py ${1:-32000000} binned | tail -1
D=/tmp/gzbench
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
rm -rf /tmp/prof$i
BNS_CLI_TIMING=1 rocprofv3 --hip-trace --stats -d /tmp/prof$i -o t --output-format csv -- /root/repo/bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.binned.fq.gz 2>&1 | grep -E "process_dataset"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/prof$i/t_hip_api_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:9]: print("  %-28s calls %6s total %8.1f ms max %8.1f ms" % (r["Name"][:28], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["MaxNs"])/1e6))
PY
done
