#!/usr/bin/env python
"""PCIe-inclusive rate of the host-buffer entry point bns_classify_batch (DESIGN.md "Measurement"):
ASCII reads in host memory -> H2D -> pack -> classify -> D2H of taxon/missing/ambig."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bonsai_amd
    import oracle_lib as O
    import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    w = synth.make_world(O, seed=3, k=31, genome_len=20000)
    ctx = bonsai_amd.Context(0)
    ctx.set_encoder(31, None, True)
    ctx.load_table(w.n_buckets, w.flags, w.keys, w.vals)   # default layout: minbucket
    ctx.load_taxonomy(w.parent)
    rng = np.random.default_rng(1)
    g = np.concatenate(list(w.genomes.values()))
    st = rng.integers(0, g.size - 150, size=n)
    bases = g[st[:, None] + np.arange(150)[None, :]].reshape(-1).copy()
    offsets = np.arange(n + 1, dtype=np.uint64) * 150
    ctx.classify(bases, offsets)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.classify(bases, offsets)
        t.append(time.perf_counter() - t0)
    print(json.dumps({"entry": "bns_classify_batch (host buffers, pageable)", "reads": n, "best_s": min(t),
                      "reads_per_s": n / min(t), "h2d_bytes": int(bases.nbytes + offsets.nbytes), "d2h_bytes": 16 * n}))
    # the same call with every host buffer in pinned memory (bns_host_alloc), as the CLI's pipeline uses it
    import ctypes as C
    L = ctx.L

    def pinned(nbytes, dtype):
        p = C.c_void_p()
        assert L.bns_host_alloc(ctx.h, nbytes, C.byref(p)) == 0
        return p, np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dtype)
    pb, hb = pinned(bases.nbytes + 8, np.uint8)
    po, ho = pinned(offsets.nbytes, np.uint64)
    outs = [pinned(4 * n, np.uint32) for _ in range(4)]
    hb[:bases.size] = bases; ho[:] = offsets
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    call = lambda: L.bns_classify_batch(ctx.h, pb.value, C.cast(po, u64p), n, 0, *[C.cast(o[0], u32p) for o in outs], None)
    assert call() == 0
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        assert call() == 0
        t.append(time.perf_counter() - t0)
    ref = ctx.classify(bases, offsets)
    assert np.array_equal(outs[0][1], ref["taxon"])
    print(json.dumps({"entry": "bns_classify_batch (host buffers, pinned)", "reads": n, "best_s": min(t), "reads_per_s": n / min(t),
                      "h2d_GBps": (bases.nbytes + offsets.nbytes) / min(t) / 1e9}))
    # packed reads: the host packer's own rate, then the packed entry point with everything pinned (what a host that keeps
    # pre-packed shards, or packs on its reader threads, hands over): 40 + 8 bytes per read up, 16 down
    import bonsai_amd as A
    for th in (1, 16):
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            words, bw, bm = A.pack_reads(bases, offsets, threads=th)
            t.append(time.perf_counter() - t0)
        print(json.dumps({"entry": "bns_pack_reads (%d thread%s)" % (th, "s" if th > 1 else ""), "reads": n, "best_s": min(t),
                          "reads_per_s": n / min(t), "ascii_GBps": bases.nbytes / min(t) / 1e9, "invalid_words": int(bw.size)}))
    pw, hw = pinned(words.nbytes, np.uint64)
    hw[:] = words
    pbw, hbw = pinned(max(8, bw.nbytes), np.uint64); pbm, hbm = pinned(max(4, bm.nbytes), np.uint32)
    hbw[:bw.size] = bw; hbm[:bm.size] = bm
    callp = lambda: L.bns_classify_batch_packed(ctx.h, C.cast(pw, u64p), C.cast(pbw, u64p), C.cast(pbm, u32p), int(bw.size), C.cast(po, u64p), n, 0,
                                                *[C.cast(o[0], u32p) for o in outs], None)
    assert callp() == 0
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        assert callp() == 0
        t.append(time.perf_counter() - t0)
    assert np.array_equal(outs[0][1], ref["taxon"]) and np.array_equal(outs[1][1], ref["missing"])
    print(json.dumps({"entry": "bns_classify_batch_packed (host buffers, pinned)", "reads": n, "best_s": min(t), "reads_per_s": n / min(t),
                      "h2d_bytes": int(words.nbytes + offsets.nbytes + bw.nbytes + bm.nbytes), "d2h_bytes": 16 * n,
                      "h2d_GBps": (words.nbytes + offsets.nbytes) / min(t) / 1e9}))
    # taxon only back (4 bytes per read), as a caller that wants the classification alone asks for it
    callt = lambda: L.bns_classify_batch_packed(ctx.h, C.cast(pw, u64p), C.cast(pbw, u64p), C.cast(pbm, u32p), int(bw.size), C.cast(po, u64p), n, 0,
                                                C.cast(outs[0][0], u32p), None, None, None, None)
    assert callt() == 0
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        assert callt() == 0
        t.append(time.perf_counter() - t0)
    print(json.dumps({"entry": "bns_classify_batch_packed (pinned, taxon only)", "reads": n, "best_s": min(t), "reads_per_s": n / min(t)}))


if __name__ == "__main__":
    main()
