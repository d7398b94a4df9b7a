#!/usr/bin/env python
"""Where the time of one CLI-sized GPU call goes: the packed entry points on one chunk (2^24 bases of 150-bp reads), timed per
variant, with the kernels' own time (bns_set_timing) beside the call's wall time."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (initialises the HIP runtime the way the tests do)
import oracle_lib as O, synth
import bonsai_amd as B
from bonsai_amd.context import _p, u32p, u64p
n = int(sys.argv[1]) if len(sys.argv) > 1 else 111_848
w = synth.make_world(O, seed=3, k=31, genome_len=50000)
ctx = B.Context(0)
ctx.set_encoder(31)
ctx.load_table(w.n_buckets, w.flags, w.keys, w.vals)
ctx.load_taxonomy(w.parent)
g = np.concatenate(list(w.genomes.values()))
rng = np.random.default_rng(1)
st = rng.integers(0, g.size - 150, size=n)
bases = np.ascontiguousarray(g[st[:, None] + np.arange(150)[None, :]]).reshape(-1)
offsets = (np.arange(n + 1, dtype=np.uint64) * 150)
words, bad_word, bad_mask = B.pack_reads(bases, offsets)
taxon = np.zeros(n, np.uint32); missing = np.zeros(n, np.uint32); ambig = np.zeros(n, np.uint32); n_hits = np.zeros(n, np.uint32)
hits = np.zeros(n * 150, np.uint32); run_start = np.zeros(n, np.uint64); n_runs = np.zeros(n, np.uint32)
rt = u32p(); rl = u32p(); tot = C.c_uint64()
L = ctx.L


def plain(m, a, nh, h):
    return L.bns_classify_batch_packed(ctx.h, _p(words, u64p), None, None, 0, _p(offsets, u64p), n, 0, _p(taxon, u32p),
                                       _p(missing, u32p) if m else None, _p(ambig, u32p) if a else None, _p(n_hits, u32p) if nh else None,
                                       _p(hits, u32p) if h else None)


def runs():
    return L.bns_classify_batch_packed_runs(ctx.h, _p(words, u64p), None, None, 0, _p(offsets, u64p), n, 0, _p(taxon, u32p), _p(missing, u32p),
                                            _p(ambig, u32p), _p(n_hits, u32p), _p(run_start, u64p), _p(n_runs, u32p), C.byref(rt), C.byref(rl),
                                            C.cast(C.byref(tot), u64p))


ctx.set_timing(True)
for name, fn in (("taxon only", lambda: plain(0, 0, 0, 0)), ("taxon+missing+ambig+n_hits", lambda: plain(1, 1, 1, 0)),
                 ("+ hits to the host", lambda: plain(1, 1, 1, 1)), ("runs (the CLI's Kraken call)", runs)):
    for _ in range(3):
        assert fn() == 0
    ctx.timing_summary()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        fn()
    dt = (time.perf_counter() - t0) / R
    ts = ctx.timing_summary()
    print("%-32s %.3f ms per call of %d reads (%.1f M reads/s); kernels: %s; runs total %d" % (name, dt * 1e3, n, n / dt / 1e6, ts, tot.value))
