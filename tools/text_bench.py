#!/usr/bin/env python
"""PCIe-inclusive rate of bns_classify_text: FASTQ TEXT in page-locked host memory -> H2D -> device parse + pack -> classify ->
results back (DESIGN.md "Host path").  usage: text_bench.py [n_reads] [taxon|full|runs]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bonsai_amd
    from bonsai_amd import _lib
    import oracle_lib as O
    import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    mode = sys.argv[2] if len(sys.argv) > 2 else "taxon"
    w = synth.make_world(O, seed=3, k=31, genome_len=20000)
    ctx = bonsai_amd.Context(0)
    ctx.set_encoder(31, None, True)
    ctx.load_table(w.n_buckets, w.flags, w.keys, w.vals)
    ctx.load_taxonomy(w.parent)
    rng = np.random.default_rng(1)
    g = np.concatenate(list(w.genomes.values()))
    st = rng.integers(0, g.size - 150, size=n)
    seqs = g[st[:, None] + np.arange(150)[None, :]]
    # "@r<9 digits>\n" + 150 + "\n+\n" + 150 + "\n" = 12 + 151 + 2 + 151 = 316 bytes per record
    rec = np.zeros((n, 316), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(n)
    for d in range(9):
        rec[:, 2 + d] = ord("0") + (idx // 10 ** (8 - d)) % 10
    rec[:, 11] = 10
    rec[:, 12:162] = seqs
    rec[:, 162] = 10; rec[:, 163] = ord("+"); rec[:, 164] = 10
    rec[:, 165:315] = ord("I")
    rec[:, 315] = 10
    text = rec.reshape(-1)
    L = ctx.L

    def pinned(nbytes, dtype):
        p = C.c_void_p()
        assert L.bns_host_alloc(ctx.h, nbytes, C.byref(p)) == 0
        return p, np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dtype)
    pt, ht = pinned(text.size + 64, np.uint8)
    ht[:text.size] = text
    cap = n + 16
    arrs = {k: pinned(4 * cap, np.uint32) for k in ("taxon", "missing", "ambig", "n_hits", "seq_len", "n_runs")}
    arrs["name_off"] = pinned(4 * (cap + 1), np.uint32)
    arrs["names"] = pinned(16 * cap, np.uint8)
    arrs["run_start"] = pinned(8 * cap, np.uint64)
    o = _lib.TextOut()
    o.taxon = arrs["taxon"][0].value
    if mode in ("full", "runs"):
        for k in ("missing", "ambig", "n_hits", "seq_len", "name_off", "names"):
            setattr(o, k, arrs[k][0].value)
        o.names_cap = 16 * cap
    if mode == "runs":
        o.run_start = arrs["run_start"][0].value; o.n_runs = arrs["n_runs"][0].value
    info = _lib.TextInfo()
    ptrs = (C.c_void_p * 1)(pt.value)
    sizes = np.array([text.size], dtype=np.uint64)
    ctx.set_timing(True)

    def call():
        rc = L.bns_classify_text(ctx.h, ptrs, sizes.ctypes.data_as(C.POINTER(C.c_uint64)), 1, 0xFFFFFFFFFFFFFFFF,
                                 _lib.TEXT_FINAL | _lib.TEXT_TRIM_READNO, cap, C.byref(o), C.byref(info))
        assert rc == 0 and info.status == 0 and info.n_records == n, (rc, info.status, info.why, info.n_records)
    call()
    t = []
    for _ in range(4):
        t0 = time.perf_counter()
        call()
        t.append(time.perf_counter() - t0)
    bases = seqs.reshape(-1).copy()
    offsets = np.arange(n + 1, dtype=np.uint64) * 150
    m = min(n, 200000)
    ref = ctx.classify(bases[:150 * m], offsets[:m + 1])
    assert np.array_equal(arrs["taxon"][1][:m], ref["taxon"])
    print(json.dumps({"entry": "bns_classify_text (pinned text, %s)" % mode, "reads": n, "text_bytes": int(text.size), "best_s": min(t),
                      "reads_per_s": n / min(t), "text_GBps": text.size / min(t) / 1e9, "slices": int(info.n_slices),
                      "ms_parse_kernels": info.ms_parse, "ms_classify": info.ms_classify}))


if __name__ == "__main__":
    main()
