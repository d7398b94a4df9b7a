#!/usr/bin/env python
"""Summarise a rocprofv3 output tree (tools/collect_profiles.sh) into profiles/<tag>_*:
   <tag>_kernel_stats.csv   copy of the --kernel-trace --stats table (bench.py command)
   <tag>_pmc.json           per-launch averages of the PMC passes for our kernels
   traffic.json             HBM bytes per classify launch (read by bench.py for roofline.traffic)
"""
import collections
import csv
import json
import os
import shutil
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = os.path.join(src, "kt", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        rows = list(csv.reader(open(ks)))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:25]:
                w.writerow([c[:160] for c in r])
    log = os.path.join(src, "bench_kt.log")
    bench_line = None
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                bench_line = json.loads(line)
    pmc = {}
    for d in sorted(os.listdir(src)):
        p = os.path.join(src, d, "bench_counter_collection.csv")
        if not d.startswith("pmc_") or not os.path.exists(p):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            kn = r["Kernel_Name"]
            short = None
            for name in ("classify_kernel", "pack_kernel", "probe_kernel", "rebucket_kernel", "build_kernel"):
                if name in kn:
                    short = name
            if not short:
                continue
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[short].add(r["Dispatch_Id"])
        for kname, cs in agg.items():
            pmc.setdefault(kname, {})
            for c, v in cs.items():
                pmc[kname][c] = v / len(disp[kname])
            pmc[kname]["_launches_" + d] = len(disp[kname])
    summary = {"bench": bench_line, "per_launch": pmc,
               "notes": ["FETCH_SIZE is in KiB = TCC_EA0_RDREQ_sum*64/1024 here (no 32B / 128B-bubble requests were counted)",
                         "MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced 128 B requests by 2x; for the "
                         "64-byte bucket gathers of classify_kernel the per-request size is uncalibrated, so "
                         "hbm_bytes_per_launch below is requests x 64 B (a lower bound if the fabric moves 128 B lines)"]}
    json.dump(summary, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
    ck = pmc.get("classify_kernel", {})
    if "FETCH_SIZE" in ck and bench_line:
        wr = ck.get("TCC_EA0_WRREQ_sum", 0.0) * 64
        tj = {"tag": tag, "reads_per_launch": bench_line["config"]["reads_per_gpu"], "layout": bench_line["config"]["layout"],
              "hbm_bytes_per_launch": ck["FETCH_SIZE"] * 1024 + wr, "fetch_bytes": ck["FETCH_SIZE"] * 1024, "write_bytes": wr,
              "source": "rocprofv3 --pmc FETCH_SIZE / TCC_EA0_WRREQ_sum, separate passes, averaged per classify_kernel launch"}
        json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: {c: v for c, v in vs.items() if not c.startswith("_")} for k, vs in pmc.items() if k in ("classify_kernel", "pack_kernel")}, indent=1)[:3000])


if __name__ == "__main__":
    main()
