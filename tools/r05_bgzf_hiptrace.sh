#!/bin/bash
# round 5: where the host threads of the BGZF device path spend their time in the HIP runtime (rocprofv3 --hip-trace, no counters)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/r05_bgzf_make.py 24000000 | tail -1
D=/tmp/bgzfbench; O=/tmp/r05_bgzf_hip; rm -rf $O; mkdir -p $O
BNS_NORMAL_EXIT=1 BNS_CLI_TIMING=1 rocprofv3 --hip-trace --output-format csv -d $O -o t -- bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.bgzf.fq.gz 2>&1 | grep -E "BGZF text|process_dataset" | cut -c1-300
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/r05_bgzf_hip/**/*hip_api_trace.csv", recursive=True)[0]
rows = [(r.get("Thread_Id", "?"), r["Function"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
main = collections.Counter(t for t, *_ in rows).most_common(1)[0][0]
# the window of the data set itself: from the first call of a thread that is not the main one (readers page-locking their slots) to the last launch
t0 = min(a for t, n, a, b in rows if t != main); t1 = max(b for t, n, a, b in rows if n == "hipLaunchKernel")
print("window %.3f s (first call off the main thread .. last kernel launch); calls that START in it:" % ((t1 - t0) / 1e9))
tot = collections.defaultdict(lambda: collections.Counter()); cnt = collections.defaultdict(lambda: collections.Counter())
for t, n, a, b in rows:
    if a < t0 or a > t1: continue
    tot[t][n] += b - a; cnt[t][n] += 1
for t in sorted(tot, key=lambda x: -sum(tot[x].values()))[:6]:
    print("thread", t, "total %.3f s in HIP calls" % (sum(tot[t].values()) / 1e9))
    for n, d in tot[t].most_common(7):
        print("    %-34s %6d calls %.3f s  (%.1f us each)" % (n, cnt[t][n], d / 1e9, d / cnt[t][n] / 1e3))
PY
