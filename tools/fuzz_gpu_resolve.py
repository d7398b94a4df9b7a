#!/usr/bin/env python
"""Randomised soak of resolve_tree / lca on random forests (deep chains, several roots, ids that are not keys of the
parent map, taxid 0, (tax_t)-1, big counts) against the oracle.  usage: tools/fuzz_gpu_resolve.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402   (the checker)
import bonsai_amd               # noqa: E402

O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = bonsai_amd.Context(0)
t0 = time.time()
it = 0
n_cases = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 100003 + it)
    n = int(rng.integers(3, 400))
    ids = np.unique(rng.integers(2, 3000, size=n)).tolist()
    shape = rng.random()
    pairs = [(1, 1)]
    known = [1]
    for x in ids:
        if shape < 0.3:   par = known[-1]                                  # one long chain
        elif shape < 0.6: par = known[int(rng.integers(len(known)))]        # random tree
        else:             par = known[int(rng.integers(max(0, len(known) - 3), len(known)))]   # deep and bushy
        if rng.random() < 0.03: par = int(rng.integers(3000, 3100))         # parent that is not a key itself (another root)
        pairs.append((x, par)); known.append(x)
    tax = O.Taxonomy(pairs=pairs)
    ctx.load_taxonomy(tax.parent)
    pool = np.array(known + [0, 0xFFFFFFFF, 5000, 3050], dtype=np.uint64)
    cases = []
    for _ in range(300):
        m = int(rng.integers(1, 12)) if rng.random() < 0.9 else int(rng.integers(12, 150))
        ks = rng.choice(pool, size=min(m, pool.size), replace=False)
        hi = int(rng.choice([2, 5, 60000, 65535]))
        cases.append((ks.astype(np.uint32), rng.integers(0, hi + 1, size=ks.size).astype(np.uint16)))
    keys = np.concatenate([c[0] for c in cases]); counts = np.concatenate([c[1] for c in cases])
    starts = np.zeros(len(cases) + 1, dtype=np.uint64); starts[1:] = np.cumsum([c[0].size for c in cases])
    got = ctx.resolve(keys, counts, starts)
    exp = np.array([tax.resolve(c[0], c[1]) for c in cases], dtype=np.uint32)
    if not np.array_equal(got, exp):
        bad = int(np.flatnonzero(got != exp)[0])
        print("RESOLVE MISMATCH seed", seed0 * 100003 + it, "case", bad, "keys", cases[bad][0].tolist(), "counts", cases[bad][1].tolist(),
              "got", int(got[bad]), "exp", int(exp[bad]))
        sys.exit(1)
    it += 1; n_cases += len(cases)
print("resolve fuzz ok: %d taxonomies, %d counters, %.0f s" % (it, n_cases, time.time() - t0))
