#!/bin/bash
# round 5: kernel + memory-copy timeline of the plain PAIR path (2 x 16 M mates): how busy are the link and the GPU
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/cli_bench.py 32000000 --paired > /dev/null 2>&1
D=/tmp/clibench; O=/tmp/r05_pairtrace; rm -rf $O; mkdir -p $O
BNS_NORMAL_EXIT=1 BNS_CLI_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- bonsai_amd/bin/bonsai classify -a -K -p 4 -o /dev/null $D/bns.db $D/nodes.dmp $D/r_1.fq $D/r_2.fq 2>&1 | grep -E "pair of files|process_dataset" | cut -c1-300
python - <<'PY'
import csv, glob
def load(pat, a="Start_Timestamp", b="End_Timestamp"):
    f = glob.glob("/tmp/r05_pairtrace/**/*" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = load("kernel_trace.csv"); m = load("memory_copy_trace.csv")
def union(iv):
    iv = sorted(iv); u = 0; cur = None
    for a, b in iv:
        if cur is None: cur = [a, b]
        elif a <= cur[1]: cur[1] = max(cur[1], b)
        else: u += cur[1] - cur[0]; cur = [a, b]
    if cur: u += cur[1] - cur[0]
    return u
ki = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "ingest" in r["Kernel_Name"] or "classify" in r["Kernel_Name"]]
h2d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in m if "HOST_TO_DEVICE" in r.get("Direction", "")]
big = [(a, b) for a, b in h2d if b - a > 200000]            # (pieces of 64 MiB are ~1.2 ms; the rest are words)
t0 = min(a for a, b in ki); t1 = max(b for a, b in ki)
print("text + classify kernels: span %.3f s, busy %.3f s" % ((t1 - t0) / 1e9, union(ki) / 1e9))
if big:
    busy = union(big)
    print("uploads longer than 0.2 ms: %d, link busy %.3f s of a span of %.3f s; average piece %.2f ms; gaps between pieces > 1 ms: %s" % (len(big), busy / 1e9, (max(b for a, b in big) - min(a for a, b in big)) / 1e9,
          sum(b - a for a, b in big) / len(big) / 1e6, " ".join("%.1f" % ((big[i + 1][0] - big[i][1]) / 1e6) for i in range(len(big) - 1) if big[i + 1][0] - big[i][1] > 1e6)[:400]))
PY
