#!/usr/bin/env python
"""RollingHasher over long reads (10 kb synthetic reads; self-consistency only, F10): kernel time from
rocprofv3 --kernel-trace --stats around this script, end-to-end time printed here (dominated by the 8 B/base copy back)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bonsai_amd, oracle_lib as O
n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 20000), 10000
rng = np.random.default_rng(5)
bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n * L)
bases[rng.integers(0, n * L, size=n * L // 1000)] = ord("N")
offsets = np.arange(n + 1, dtype=np.uint64) * L
ctx = bonsai_amd.Context(0)
for k, canon, w in ((21, True, 0), (31, False, 0), (21, True, 50)):
    ctx.rolling_hash(bases[:L * 10], offsets[:11], k, canon, w=w)
    t0 = time.perf_counter()
    out = ctx.rolling_hash(bases, offsets, k, canon, w=w)
    dt = time.perf_counter() - t0
    exp = O.rolling_hash(bases[:L].tobytes(), k, canon, w=w)
    assert np.array_equal(out[0], exp)
    print("k %d canon %d w %d: %d x %d bp, %.2f s end to end (%.1f M bases/s), %d values" % (k, canon, w, n, L, dt, n * L / dt / 1e6, sum(a.size for a in out)))
