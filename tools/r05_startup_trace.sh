#!/bin/bash
# round 5: what a `bonsai classify` process does in the HIP runtime before its first text kernel and at its end (rocprofv3 --hip-trace on a small plain file)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=/tmp/clibig; mkdir -p $D
python tools/make_fastq.py 4000000 $D/r.fq > /dev/null
O=/tmp/r05_start; rm -rf $O; mkdir -p $O
BNS_NORMAL_EXIT=1 BNS_CLI_TIMING=1 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O -o t -- bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.fq 2>&1 | grep -E "start-up|process_dataset|since start" | cut -c1-250
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/r05_start/**/*hip_api_trace.csv", recursive=True)[0]
rows = [(r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Thread_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
t0 = rows[0][1]
print("first HIP call at 0; calls longer than 2 ms, in order (start s: name, ms, thread):")
for n, a, b, t in rows:
    if b - a > 2e6: print("  %.3f: %-34s %8.1f ms  thread %s" % ((a - t0) / 1e9, n, (b - a) / 1e6, t))
print("last HIP call ends at %.3f s; %d calls" % ((max(r[2] for r in rows) - t0) / 1e9, len(rows)))
PY
