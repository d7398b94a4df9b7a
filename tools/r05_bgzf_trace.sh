#!/bin/bash
# round 5: kernel timeline of the BGZF device path (who runs beside whom): rocprofv3 --kernel-trace of the CLI on a 24 M-read BGZF file
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/r05_bgzf_make.py ${1:-24000000} | tail -1
D=/tmp/bgzfbench; O=gpurun_out/${BNS_TRACE_NAME:-r05_bgzf_trace}; rm -rf $O; mkdir -p $O
BNS_NORMAL_EXIT=1 BNS_CLI_TIMING=1 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- bonsai_amd/bin/bonsai classify -a -K -o /dev/null $D/bns.db $D/nodes.dmp $D/r.bgzf.fq.gz 2>&1 | grep -E "BGZF text|process_dataset" | cut -c1-300
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/" + __import__("os").environ.get("BNS_TRACE_NAME", "r05_bgzf_trace") + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r["Kernel_Name"]; a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    kind = "inflate" if "inflate_" in n else ("classify" if "classify_kernel" in n else ("ingest" if "ingest" in n else ("runs" if "hit_runs" in n else "other")))
    ev.append((a, b, kind, n[:50], r.get("Queue_Id", "?")))
t0 = min(e[0] for e in ev); t1 = max(e[1] for e in ev)
print("kernels", len(ev), "span %.3f s" % ((t1 - t0) / 1e9))
tot = collections.Counter(); cnt = collections.Counter()
for a, b, k, n, q in ev: tot[k] += b - a; cnt[k] += 1
for k in tot: print("  %-9s %5d launches, %.3f s summed, %.1f us average" % (k, cnt[k], tot[k] / 1e9, tot[k] / cnt[k] / 1e3))
# classify / ingest kernels: average duration when an inflate kernel is running at their start vs not
inf = sorted((a, b) for a, b, k, n, q in ev if k == "inflate")
def busy(t): return any(a <= t < b for a, b in inf)
for kind in ("classify", "ingest"):
    d = {True: [], False: []}
    for a, b, k, n, q in ev:
        if k == kind: d[busy(a)].append(b - a)
    for flag in (True, False):
        if d[flag]: print("  %s kernels that start while an inflate kernel runs = %s: %d, average %.1f us" % (kind, flag, len(d[flag]), sum(d[flag]) / len(d[flag]) / 1e3))
# union of inflate busy time
u = 0; cur = None
for a, b in inf:
    if cur is None: cur = [a, b]
    elif a <= cur[1]: cur[1] = max(cur[1], b)
    else: u += cur[1] - cur[0]; cur = [a, b]
if cur: u += cur[1] - cur[0]
allk = sorted((a, b) for a, b, k, n, q in ev)
ub = 0; cur = None; gaps = []
for a, b in allk:
    if cur is None: cur = [a, b]
    elif a <= cur[1]: cur[1] = max(cur[1], b)
    else:
        ub += cur[1] - cur[0]
        if a - cur[1] > 2e6: gaps.append(((cur[1] - t0) / 1e9, (a - cur[1]) / 1e6))
        cur = [a, b]
if cur: ub += cur[1] - cur[0]
print("  some kernel runs during %.3f s of the span; idle stretches > 2 ms (at s: ms): %s" % (ub / 1e9, " ".join("%.2f:%.0f" % g for g in gaps[:40])))
print("  inflate kernels cover %.3f s of the span; queues used: %s" % (u / 1e9, sorted(set(e[4] for e in ev))))
PY

# per-kernel sums (round 6)
python - <<'PY'
import csv, glob, collections, os
f = glob.glob("gpurun_out/" + os.environ.get("BNS_TRACE_NAME", "r05_bgzf_trace") + "/**/*kernel_trace.csv", recursive=True)
if f:
    tot = collections.Counter(); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"].split("(")[0][-40:]; tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
    for n, t in tot.most_common(14): print("  %-42s %5d launches %.4f s  %.1f us avg" % (n, cnt[n], t / 1e9, t / cnt[n] / 1e3))
PY
find $O -name "*kernel_trace.csv" -size +30M -delete
