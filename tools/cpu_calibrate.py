#!/usr/bin/env python
"""CPU-baseline calibration: is the oracle port -- what bench.py times as `cpu_baseline`
(kind "port") -- as fast as the REFERENCE's own functions?  BUILD CONTAINER ONLY (needs /root/reference).

Both sides classify the same reads against the same khash arrays (>= 200 MB: out of cache, as a real db is):
  port : oracle/bns_oracle.c  bo_classify_batch            (liboracle.so as `make -C oracle` builds it: gcc -O3)
  ref  : oracle/ref_harness.cpp ref_classify_batch          -- the reference's DNA4 / rhmask / mul / canonical_representation in
         the loop of encoder.h:246-271 with the hit lambda inlined, its kh_get, linear::counter and resolve_tree, batched inside
         the library (no per-read foreign call), compiled with the REFERENCE's flags (its Makefile: -O3 -funroll-loops
         -march=native -fno-strict-aliasing -fno-rtti -fopenmp -DNDEBUG) into oracle/_ref/libbns_ref_cal.so
at 1 thread and at N threads, whole path and cut after each phase (encode / + kh_get / + vote and resolve_tree), results
compared read by read.  Writes profiles/r04_cpu_calibration.json; bench.py copies `port_over_ref` and the reference's
reads/s/thread into its cpu_baseline block.

usage: python tools/cpu_calibrate.py [n_reads=400000] [threads=nproc]"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

REF = os.environ.get("BNS_REF", "/root/reference")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
K, L, NG, G = 31, 150, 64, 160_000


def build_cal():
    out = os.path.join(ROOT, "oracle", "_ref", "libbns_ref_cal.so")
    src = os.path.join(ROOT, "oracle", "ref_harness.cpp")
    if not os.path.isdir(os.path.join(REF, "include", "bonsai")):
        sys.exit("cpu_calibrate: the reference checkout is not at %s (this script runs in the build container only)" % REF)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    flags = ["-O3", "-funroll-loops", "-march=native", "-fno-strict-aliasing", "-fno-rtti", "-fopenmp", "-DNDEBUG", "-std=c++17"]
    subprocess.run(["g++"] + flags + ["-fPIC", "-shared", "-I" + REF, "-I" + REF + "/include", "-I" + os.path.join(ROOT, "oracle", "_ref"),
                                      "-o", out, src, "-lz"], check=True)
    return out, " ".join(flags)


def taxonomy(n):
    """4-ary tree over n leaves (bench.make_taxonomy's shape), as (child, parent) pairs + leaf ids"""
    sizes = [n]
    while sizes[-1] > 1:
        sizes.append((sizes[-1] + 3) // 4)
    sizes = sizes[::-1]
    base, nxt = [], 1
    for s in sizes:
        base.append(nxt); nxt += s
    pairs = [(1, 0)]
    for lvl in range(1, len(sizes)):
        for i in range(sizes[lvl]):
            pairs.append((base[lvl] + i, base[lvl - 1] + i // 4))
    return pairs, [base[-1] + i for i in range(n)]


def best_of(fn, reps=3):
    b = None
    for _ in range(reps):
        t = time.perf_counter(); fn(); e = time.perf_counter() - t
        b = e if b is None or e < b else b
    return b


def main():
    so, flags = build_cal()
    R = C.CDLL(so)
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    R.ref_khc_view.restype = C.c_void_p
    R.ref_khc_view.argtypes = [C.c_uint64] * 4 + [u32p, u64p, u32p]
    R.ref_khp_from_pairs.restype = C.c_void_p; R.ref_khp_from_pairs.argtypes = [u32p, u32p, C.c_uint32]
    R.ref_classify_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_char_p, u64p, C.c_uint64, u32p, C.c_int, C.c_int, u64p]
    L_ = O.lib()
    L_.bo_classify_batch_phase.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_char_p, u64p, C.c_uint64, C.c_void_p,
                                           C.c_int, C.c_int, u64p]
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genomes = acgt[rng.integers(0, 4, size=(NG, G))]
    # relatives share stretches, so that the db holds LCAs above the leaves and resolve_tree has ties to fold
    for g in range(NG):
        if g % 4:
            genomes[g, : G // 4] = genomes[g - g % 4, : G // 4]
        if g % 16:
            genomes[g, G // 4: G // 4 + G // 16] = genomes[g - g % 16, G // 4: G // 4 + G // 16]
    pairs, leaves = taxonomy(NG)
    tax = O.Taxonomy(pairs=pairs)
    table = O.Table()
    t0 = time.time()
    for g in range(NG):
        O.lca_map_add(table, tax, K, genomes[g].tobytes(), leaves[g])
    nb, size, nocc, ub = table.header()
    print("db: %d keys in %d khash buckets (%.0f MB of arrays), built in %.1f s" % (size, nb, nb * 12.25 / 1e6, time.time() - t0), flush=True)
    flat = genomes.reshape(-1)
    gi = rng.integers(0, NG, size=N); st = rng.integers(0, G - L, size=N)
    reads = flat[(gi * G + st)[:, None] + np.arange(L)[None, :]].copy()
    sub = rng.random((N, L)) < 0.01
    reads[sub] = acgt[rng.integers(0, 4, size=int(sub.sum()))]
    reads[rng.random((N, L)) < 0.001] = ord("N")
    bases = reads.reshape(-1).tobytes()
    offsets = (np.arange(N + 1, dtype=np.uint64) * L)
    c = table.h.contents
    rdb = R.ref_khc_view(nb, size, nocc, ub, c.flags, C.cast(c.keys, u64p), c.vals)
    ch = np.array([p[0] for p in pairs], dtype=np.uint32); pa = np.array([p[1] for p in pairs], dtype=np.uint32)
    rtax = R.ref_khp_from_pairs(ch.ctypes.data_as(u32p), pa.ctypes.data_as(u32p), ch.size)
    out_ref = np.zeros((N, 4), dtype=np.uint32)
    out_port = np.zeros(N, dtype=[("taxon", "<u4"), ("missing", "<u4"), ("ambig", "<u4"), ("n_hits", "<u4")])
    offp = offsets.ctypes.data_as(u64p)
    sink = C.c_uint64()

    def ref_run(nt, phase):
        R.ref_classify_batch(rdb, rtax, K, 1, bases, offp, N, out_ref.ctypes.data_as(u32p), nt, phase, C.byref(sink))

    def port_run(nt, phase):
        L_.bo_classify_batch_phase(table.h, C.byref(tax.t), K, None, 1, bases, offp, N,
                                   out_port.ctypes.data_as(C.c_void_p), nt, phase, C.byref(sink))

    ref_run(NT, 2); port_run(NT, 2)
    same = (np.array_equal(out_ref[:, 0], out_port["taxon"]) and np.array_equal(out_ref[:, 1], out_port["missing"])
            and np.array_equal(out_ref[:, 2], out_port["ambig"]) and np.array_equal(out_ref[:, 3], out_port["n_hits"]))
    print("results identical: %s; classified %.3f" % (same, float((out_ref[:, 0] != 0).mean())), flush=True)
    res = {"n_reads": N, "read_len": L, "k": K, "db_keys": int(size), "khash_buckets": int(nb), "khash_bytes": int(nb * 12.25),
           "results_identical": bool(same), "threads_n": NT, "ref_flags": flags, "port_flags": "gcc -O3 -std=gnu11 -fopenmp (oracle/Makefile)",
           "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown"),
           "where": "build container (the reference checkout does not travel to the GPU box)"}
    for side, run in (("ref", ref_run), ("port", port_run)):
        for nt in (1, NT):
            ph = [best_of(lambda: run(nt, p)) for p in (0, 1, 2)]
            key = "%s_%dt" % (side, nt)
            res[key] = {"reads_per_s": N / ph[2], "seconds": ph[2],
                        "phase_seconds": {"encode": ph[0], "probe": ph[1] - ph[0], "vote_resolve": ph[2] - ph[1]},
                        "phase_frac": {"encode": ph[0] / ph[2], "probe": (ph[1] - ph[0]) / ph[2], "vote_resolve": (ph[2] - ph[1]) / ph[2]}}
            print("%-9s %8.0f reads/s  (encode %.2f s, +probe %.2f s, +vote/resolve %.2f s)" % (key, N / ph[2], ph[0], ph[1] - ph[0], ph[2] - ph[1]), flush=True)
    res["ref_reads_per_s_per_thread"] = res["ref_1t"]["reads_per_s"]
    res["port_over_ref_1t"] = res["port_1t"]["reads_per_s"] / res["ref_1t"]["reads_per_s"]
    res["port_over_ref_nt"] = res["port_%dt" % NT]["reads_per_s"] / res["ref_%dt" % NT]["reads_per_s"]
    print("port / ref: %.3f at 1 thread, %.3f at %d threads" % (res["port_over_ref_1t"], res["port_over_ref_nt"], NT))
    with open(os.path.join(ROOT, "profiles", "r04_cpu_calibration.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
