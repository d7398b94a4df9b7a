#!/usr/bin/env python
"""Randomised soak of encode (incl. windowed minimizers) and of the device db build against the oracle
(verification aid).  usage: tools/fuzz_gpu_build.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402   (the checker)
import synth                    # noqa: E402
import bonsai_amd               # noqa: E402
from test_gpu_build import device_build, present_pairs   # noqa: E402

O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = bonsai_amd.Context(0)
t0 = time.time()
it = 0
while time.time() - t0 < budget:
    seed = seed0 * 100003 + it
    rng = np.random.default_rng(seed)
    k = int(rng.choice([5, 9, 15, 16, 21, 27, 31, 31, 32]))
    spaced = rng.random() < 0.25
    gaps = [int(x) for x in rng.integers(0, 3, size=k - 1)] if spaced else None
    canon = True if spaced else bool(rng.random() < 0.7)
    comb = k + (sum(gaps) if gaps else 0)
    windowed = rng.random() < 0.5
    w = int(rng.integers(comb + 1, comb + 64)) if windowed else comb
    score = int(rng.integers(0, 3)) if (windowed and not spaced) else int(rng.integers(0, 2))   # 2 = the string overload's real entropy (contiguous seeds)
    if windowed and rng.random() < 0.15:
        w = int(rng.integers(comb + 64, comb + 1024))          # wide windows
    seqs = [b"", b"T" * 90, b"ACGT" * 40, b"A" * 33 + b"N" + b"C" * 70]
    for L in rng.integers(1, 5000, size=12):
        s = bytearray(synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(rng.choice([0, 0.01, 0.05])), 0.1).tobytes())
        if rng.random() < 0.4:                                 # T runs (the -C windowed path restarts at every 32nd T when k >= 31)
            a = int(rng.integers(0, len(s))); n = int(rng.choice([31, 32, 33, 64, 100, 2500]))
            s[a:a + n] = b"T" * len(s[a:a + n])
        seqs.append(bytes(s))
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    ctx.set_encoder(k, gaps, canonicalize=canon, spaced_intended=True)
    if windowed:
        ctx.set_window(w, score)
    got = ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        if windowed and score == 2:
            exp = O.encode_windowed_entropy_str(s, k, w, canon)
        else:
            exp = O.encode_windowed(s, k, w, score, gaps=gaps, canon=canon) if windowed else O.encode(s, k, gaps=gaps, canon=canon, spaced_intended=True)
        if not np.array_equal(g, exp):
            print("ENCODE MISMATCH seed", seed, "k", k, "gaps", gaps, "canon", canon, "w", w, "score", score, "len", len(s), g.size, exp.size)
            sys.exit(1)
    # RollingHasher: random k (also beyond the 64-bit word), both strands modes, random or default character tables
    rk = int(rng.choice([1, 2, 7, 21, 31, 32, 63, 64, 65, 97, 200]))
    rcanon = bool(rng.random() < 0.5)
    tabs = None if rng.random() < 0.3 else (rng.integers(0, 1 << 63, size=256, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=256, dtype=np.uint64),
                                            rng.integers(0, 1 << 63, size=256, dtype=np.uint64))
    rw = 0 if rng.random() < 0.5 else int(rk + rng.integers(0, 80))
    rgot = ctx.rolling_hash(bases, offsets, rk, rcanon, tabs, w=rw)
    for s, g in zip(seqs, rgot):
        if not np.array_equal(g, O.rolling_hash(s, rk, rcanon, tabs, w=rw)):
            print("ROLLING MISMATCH seed", seed, "k", rk, "canon", rcanon, "w", rw, "len", len(s)); sys.exit(1)
    # device build (optionally windowed) vs the oracle's sequential update_lca_map
    if (canon or (windowed and not spaced)) and k >= 9 and score != 2:
        wld = synth.make_world(O, seed=seed, k=k, genome_len=int(rng.choice([600, 2500])), gaps=gaps, canon=canon)
        exp_t = O.Table()
        for leaf, g in wld.genomes.items():
            if windowed:
                O.lca_map_add_windowed(exp_t, wld.tax, k, w, score, g.tobytes(), leaf, gaps=gaps, canon=canon)
            else:
                O.lca_map_add(exp_t, wld.tax, k, g.tobytes(), leaf, gaps=gaps, canon=canon)
        ef, ek, ev = exp_t.arrays()
        exp_keys, exp_vals = present_pairs(ef, ek, ev, exp_t.n_buckets)
        ctx.load_taxonomy(wld.parent)
        nb = 1 << 16
        hdr, flags, keys, vals = device_build(ctx, list(wld.genomes.values()), list(wld.genomes.keys()), nb)
        gk, gv = present_pairs(flags, keys, vals, nb)
        if not (np.array_equal(gk, exp_keys) and np.array_equal(gv, exp_vals)):
            print("BUILD MISMATCH seed", seed, "k", k, "w", w, "score", score, gk.size, exp_keys.size)
            sys.exit(1)
    ctx.set_encoder(k, gaps, canonicalize=canon, spaced_intended=True)     # clears the window
    it += 1
print("build/encode fuzz ok: %d configurations, %.0f s" % (it, time.time() - t0))
