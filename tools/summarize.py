#!/usr/bin/env python
"""Summarise one tools/measure.sh run (gpurun_out/<tag>/) into profiles/:
   <tag>_bench.json                 the bench line of that run
   <tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the bench command (top rows)
   <tag>_traffic_calibration.json   bytes per L2 memory-side read request for the three known-byte kernels of
                                    tools/micro/gather_calib.hip (probe pattern, 64-byte gather, streaming read)
   <tag>_pmc.json                   per-launch PMC averages of classify_kernel + derived figures
   traffic.json                     HBM bytes per classify_kernel launch with the calibrated request size (read by bench.py)
usage: python tools/summarize.py gpurun_out/r03m r03
"""
import collections
import csv
import json
import os
import sys


def agg(path, names):
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    d = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        s = [n for n in names if n in r["Kernel_Name"]]
        if not s:
            continue
        a[s[0]][r["Counter_Name"]] += float(r["Counter_Value"])
        d[s[0]].add(r["Dispatch_Id"])
    return {k: dict({c: v / len(d[k]) for c, v in cs.items()}, _launches=len(d[k])) for k, cs in a.items()}


def source_sha256(root):
    import hashlib
    h = hashlib.sha256()
    for f in ("bns_kernels.hip", "bns_device.hpp", "bns_kernels.hpp", "bns_api.hip"):
        h.update(open(os.path.join(root, "bonsai_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = source_sha256(root)
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
    json.dump(bench, open(os.path.join(out, tag + "_bench.json"), "w"), indent=1)
    ks = os.path.join(src, "kt", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        rows = list(csv.reader(open(ks)))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:25]:
                w.writerow([c[:160] for c in r])
    # ---- calibration
    known = {}
    for line in open(os.path.join(src, "calib_plain.log")):
        if line.startswith("{"):
            j = json.loads(line)
            known[j["kernel"]] = j                      # bytes per launch are the same for every repetition
    cal = {}
    for i in range(8):
        p = os.path.join(src, "calib_pmc%d" % i, "calib_counter_collection.csv")
        if os.path.exists(p):
            for kname, cs in agg(p, list(known)).items():
                cal.setdefault(kname, {}).update(cs)
    calib = {"source": "tools/micro/gather_calib.hip under rocprofv3 --pmc (TCC read-request counters in their own passes), table 68.7 GB",
             "kernels": {}}
    for kname, cs in cal.items():
        kb = known[kname]["bytes"]
        rd = cs.get("TCC_EA0_RDREQ_sum", 0.0)
        calib["kernels"][kname] = {
            "known_bytes_per_launch": kb, "known_fetches_per_launch": known[kname].get("fetches"),
            "TCC_EA0_RDREQ_sum": rd, "TCC_EA0_RDREQ_128B_sum": cs.get("TCC_EA0_RDREQ_128B_sum"),
            "TCC_EA0_RDREQ_64B_sum": cs.get("TCC_EA0_RDREQ_64B_sum"), "TCC_EA0_RDREQ_32B_sum": cs.get("TCC_EA0_RDREQ_32B_sum"),
            "FETCH_SIZE_KiB": cs.get("FETCH_SIZE"), "bytes_per_request": kb / rd if rd else None,
            "requests_per_fetch": (rd / known[kname]["fetches"]) if known[kname].get("fetches") else None,
            "FETCH_SIZE_over_known": (cs["FETCH_SIZE"] * 1024 / kb) if cs.get("FETCH_SIZE") else None,
            "plain_run": {k: known[kname][k] for k in ("ms", "GB_s", "Gfetch_s") if k in known[kname]}}
    g = calib["kernels"].get("calib_gather128", {})
    calib["conclusion"] = ("every L2 memory-side read request on gfx950 is a 128-byte request (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ; no "
                           "32/64-byte requests) for all three shapes, including the 64-byte gather, which therefore moves 128 bytes per "
                           "64-byte bucket; FETCH_SIZE (= RDREQ x 64 B in this rocprofv3) is exactly half the bytes moved.  "
                           "HBM read bytes = TCC_EA0_RDREQ_sum x 128.")
    json.dump(calib, open(os.path.join(out, tag + "_traffic_calibration.json"), "w"), indent=1)
    # ---- classify_kernel counters
    pmc = {}
    for pre in ("bench_pmc", "bench_sq"):
        for i in range(8):
            p = os.path.join(src, "%s%d" % (pre, i), "bench_counter_collection.csv")
            if os.path.exists(p):
                for kname, cs in agg(p, ["classify_kernel"]).items():
                    pmc.setdefault(kname, {}).update(cs)
    ck = pmc.get("classify_kernel", {})
    n_reads = bench["config"]["reads_per_gpu"]
    rd_bytes = ck.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + ck.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + ck.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    w64 = ck.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    wr_bytes = w64 * 64 + (ck.get("TCC_EA0_WRREQ_sum", 0.0) - w64) * 32
    kern_ms = bench["roofline"]["kernel_ms"]
    derived = {"hbm_read_bytes_per_launch": rd_bytes, "hbm_write_bytes_per_launch": wr_bytes,
               "hbm_bytes_per_read": (rd_bytes + wr_bytes) / n_reads,
               "alg_bytes_per_launch": bench["roofline"]["alg_bytes_per_read"] * n_reads,
               "traffic_over_algorithmic": (rd_bytes + wr_bytes) / (bench["roofline"]["alg_bytes_per_read"] * n_reads),
               "hbm_GBs_moved": (rd_bytes + wr_bytes) / (kern_ms * 1e-3) / 1e9, "kernel_ms_unprofiled": kern_ms}
    cnt = os.path.join(src, "bench_count.json")
    if os.path.exists(cnt):
        try:
            cj = json.loads([l for l in open(cnt) if l.startswith("{")][-1]).get("debug_fetch_count")
            if cj:
                derived["buckets_fetched_per_read"] = cj["buckets_fetched_per_launch"] / n_reads
                derived["probe_passes_per_read"] = cj["probe_passes_per_launch"] / n_reads
                derived["bucket_fetch_rate_G_per_s"] = cj["buckets_fetched_per_launch"] / (kern_ms * 1e-3) / 1e9
                if g.get("plain_run"):
                    derived["gather_ceiling_G_per_s"] = g["plain_run"].get("Gfetch_s")
        except Exception:
            pass
    if "GRBM_GUI_ACTIVE" in ck and "SQ_ACTIVE_INST_VALU" in ck:
        cyc = ck["GRBM_GUI_ACTIVE"] / 8.0                         # summed over the 8 XCDs
        slots = cyc / 4.0 * 1024                                   # quad-cycle issue slots of the 1024 SIMDs
        derived.update({"gpu_cycles": cyc, "valu_issue_utilisation": ck["SQ_ACTIVE_INST_VALU"] / slots,
                        "salu_issue_utilisation": ck.get("SQ_ACTIVE_INST_SCA", 0) / slots,
                        "wave_time_split": {"active": ck["SQ_ACTIVE_INST_ANY"] / ck["SQ_WAVE_CYCLES"],
                                            "issue_stalled": ck["SQ_WAIT_INST_ANY"] / ck["SQ_WAVE_CYCLES"],
                                            "waiting_on_counters": ck["SQ_WAIT_ANY"] / ck["SQ_WAVE_CYCLES"]}})
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM"):
        if c in ck:
            derived.setdefault("instructions_per_read", {})[c[9:]] = ck[c] / n_reads
    json.dump({"bench": bench, "classify_kernel_per_launch": ck, "derived": derived,
               "notes": ["counter passes ran the bench command with --steps 2 --warmup 1; kernel_ms is the un-profiled HIP-event mean of the default run",
                         "SQ_* cycle counters are in quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs"]},
              open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
    # ---- probe kernel: requests per lookup under the counters
    pp = os.path.join(src, "probe_pmc", "bench_counter_collection.csv")
    if os.path.exists(pp) and bench.get("probe_roofline"):
        pk = agg(pp, ["probe_kernel"]).get("probe_kernel", {})
        if pk:
            pr = dict(bench["probe_roofline"])
            pr.update({"TCC_EA0_RDREQ_sum_per_launch": pk.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_128B_sum_per_launch": pk.get("TCC_EA0_RDREQ_128B_sum"),
                       "read_requests_per_lookup": pk.get("TCC_EA0_RDREQ_sum", 0) / pr["keys"],
                       "hbm_read_bytes_per_lookup": pk.get("TCC_EA0_RDREQ_sum", 0) * 128 / pr["keys"],
                       "hbm_read_GBs": pk.get("TCC_EA0_RDREQ_sum", 0) * 128 / (pr["kernel_ms"] * 1e-3) / 1e9,
                       "frac_of_8TBs_moved": pk.get("TCC_EA0_RDREQ_sum", 0) * 128 / (pr["kernel_ms"] * 1e-3) / 8e12})
            json.dump(pr, open(os.path.join(out, tag + "_probe.json"), "w"), indent=1)
            json.dump({"tag": tag, "source_sha256": sha, "keys": pr["keys"], "hbm_read_bytes_per_lookup": pr["hbm_read_bytes_per_lookup"],
                       "source": "TCC_EA0_RDREQ_sum x 128 B per probe_kernel launch / keys (profiles/%s_probe.json)" % tag},
                      open(os.path.join(out, "probe_traffic.json"), "w"), indent=1)
    c2 = os.path.join(src, "bench_c2.json")
    if os.path.exists(c2):
        try:
            b2 = json.loads([l for l in open(c2) if l.startswith("{")][-1])
            pp2 = os.path.join(src, "c2_pmc", "bench_counter_collection.csv")
            if os.path.exists(pp2):                              # configs[2]'s own traffic, measured in the same pass
                k2 = agg(pp2, ["classify_kernel"]).get("classify_kernel", {})
                if k2:
                    w64_2 = k2.get("TCC_EA0_WRREQ_64B_sum", 0.0)
                    b2["roofline"]["traffic"] = k2.get("TCC_EA0_RDREQ_sum", 0) * 128 + w64_2 * 64 + (k2.get("TCC_EA0_WRREQ_sum", 0.0) - w64_2) * 32
                    b2["roofline"]["traffic_source"] = "rocprofv3 --pmc TCC_EA0_RDREQ_sum / WRREQ on the same command, per classify_kernel launch, 128 B per read request"
            json.dump(b2, open(os.path.join(out, tag + "_bench_configs2.json"), "w"), indent=1)
        except Exception:
            pass
    ks2 = os.path.join(src, "kt_c2", "bench_kernel_stats.csv")
    if os.path.exists(ks2):
        rows = list(csv.reader(open(ks2)))
        with open(os.path.join(out, tag + "_kernel_stats_configs2.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:12]:
                w.writerow([c[:160] for c in r])
    cfg = bench["config"]
    tj = {"tag": tag, "source_sha256": sha, "reads_per_launch": n_reads, "read_len": cfg["read_len"], "layout": cfg["layout"], "k": cfg["k"],
          "db_window": cfg.get("db_window", 0), "genome_len": cfg.get("genome_len", 1 << 18), "genomes": cfg.get("genomes", 1024),
          "genome_model": (cfg.get("genome_model") or {}).get("model", "uniform"), "table_buckets": cfg.get("table_buckets"),
          "identity_bits": cfg.get("table_identity_bits"),
          "hbm_bytes_per_launch": rd_bytes + wr_bytes, "read_bytes": rd_bytes, "write_bytes": wr_bytes,
          "request_bytes": 128, "FETCH_SIZE_bytes_uncorrected": ck.get("FETCH_SIZE", 0) * 1024,
          "source": "rocprofv3 --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B}_sum and TCC_EA0_WRREQ_{sum,64B}_sum (separate passes), per "
                    "classify_kernel launch; request size calibrated on known byte counts (profiles/%s_traffic_calibration.json): "
                    "128 B per read request, i.e. 2x FETCH_SIZE" % tag}
    json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
