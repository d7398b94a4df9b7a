#!/usr/bin/env python
"""Randomised soak of the GPU classify path against the oracle (profiling/verification aid, not part of the test tiers):
random k, canonical or not, spaced or not, layout, read lengths, N / lower-case / substitution rates, paired or not,
repetitive genomes (low-complexity and shared blocks).  usage: tools/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402   (the checker)
import synth                    # noqa: E402
import bonsai_amd               # noqa: E402

O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
start_it = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # resume at a given configuration (debugging)
ctx = bonsai_amd.Context(0)
t0 = time.time()
it = start_it
n_reads_total = 0
prev_cfg = None
while time.time() - t0 < budget:
    seed = seed0 * 100003 + it
    rng = np.random.default_rng(seed)
    k = int(rng.choice([9, 13, 19, 20, 21, 27, 28, 31, 31, 31, 32]))
    canon = bool(rng.random() < 0.8)
    gaps = None
    if rng.random() < 0.2:
        gaps = [int(x) for x in rng.integers(0, 3, size=k - 1)]
        canon = True
    layout = int(rng.choice([0, 1, 2, 2, 2]))
    glen = int(rng.choice([800, 3000, 12000]))
    w = synth.make_world(O, seed=seed, k=k, genome_len=glen, gaps=gaps, canon=canon)
    if rng.random() < 0.3:                                   # low-complexity extras: poly-X and short tandem repeats
        leaf = list(w.genomes)[0]
        for s in (b"A" * 200, b"T" * 150, b"ACACACACAC" * 30, b"ACGT" * 80, (b"G" * 40 + b"N" + b"C" * 60)):
            O.lca_map_add(w.table, w.tax, k, s, leaf, gaps=gaps, canon=canon)
            w.genomes[leaf] = np.concatenate([w.genomes[leaf], np.frombuffer(s.replace(b"N", b"A"), dtype=np.uint8)])
        w.flags, w.keys, w.vals = w.table.arrays(); w.n_buckets = w.table.n_buckets
    ctx.set_encoder(k, gaps, canonicalize=canon, spaced_intended=True)
    span = int(rng.choice([0, 0, 8, 11, 15]))                 # the clustered table's minimizer window: chosen by the loader or fixed
    ctx.set_minimizer_span(span)
    ii = np.arange(w.n_buckets)
    n_keys = int((((w.flags[ii >> 4] >> ((ii & 15) << 1)) & 3) == 0).sum())
    ctx.set_minimizer_identity(int(rng.choice([0, 0, 32, 52])))   # the clustered table's minimizer identity: chosen by the loader or fixed
    # its size: from the key count (default), crowded (95 % load: chains fill, keys overflow, the cooperative overflow lookup
    # runs), or some odd bucket count (the index is a multiply-high, not a mask)
    mode = int(rng.choice([0, 0, 1, 2, 3])) if layout == 2 else 0
    # how it is filled and looked up (round 4): the loader's choice, or arrival order / group by group forced (0x10 / 0x20), with the
    # crowded-table form of the lookup -- cooperative overflow lookup + tag bits -- forced on or off (0x8000 / 0x2000)
    ctx.debug_set(int(rng.choice([0, 0, 0x10, 0x20, 0x20 | 0x8000, 0x10 | 0x8000, 0x20 | 0x2000])) if layout == 2 else 0)
    ctx.set_bucket_slots_log2(0)
    ctx.set_table_buckets([0, 0, n_keys // 9 + 3, n_keys // 3 + 7, 2 * n_keys + 1][mode + 1] if mode else 0)
    ctx.load_table(w.n_buckets, w.flags, w.keys, w.vals, layout=layout)
    ctx.load_taxonomy(w.parent)
    paired = bool(rng.random() < 0.3)
    n = int(rng.integers(2, 1500)) & ~1
    length = int(rng.choice([40, 100, 150, 151, 250, 700, 2300]))
    reads = synth.simulate_reads(rng, w.genomes, n, length=length, sub_rate=float(rng.choice([0, 0.01, 0.05])),
                                 n_rate=float(rng.choice([0, 0.001, 0.02])), random_frac=0.1, var_len=bool(rng.random() < 0.5),
                                 lower_rate=float(rng.choice([0, 0.1])))
    if rng.random() < 0.3:
        reads[int(rng.integers(len(reads)))] = np.zeros(0, dtype=np.uint8)
    bases, offsets = synth.concat(reads)
    exp = O.classify_batch(w.table, w.tax, k, bases, offsets, paired=paired, gaps=gaps, canon=canon, spaced_intended=True)
    got = ctx.classify(bases, offsets, paired=paired, want_hits=True)
    gr = ctx.classify_runs(bases, offsets, paired=paired)
    pw, pbw, pbm = bonsai_amd.pack_reads(bases, offsets, threads=int(rng.integers(1, 4)))
    gp = ctx.classify_packed(pw, pbw, pbm, offsets, paired=paired, want_hits=True)      # the packed entry point: same answers
    if not all(np.array_equal(a, b) for a, b in zip(gp["hits"], got["hits"])):
        print("PACKED HITS MISMATCH seed", seed); sys.exit(1)
    for key in ("taxon", "missing", "ambig", "n_hits"):
        if not (np.array_equal(got[key], exp[key]) and np.array_equal(gr[key], exp[key]) and np.array_equal(gp[key], exp[key])):
            bad = np.flatnonzero(got[key] != exp[key])
            print("MISMATCH seed", seed, "span", span, "k", k, "canon", canon, "gaps", gaps, "layout", layout, "paired", paired, "len", length, key,
                  "units", bad[:5], "got", got[key][bad[:5]], "exp", exp[key][bad[:5]])
            # is it the configuration or something a previous configuration left behind in the context?
            c2 = bonsai_amd.Context(0)
            c2.set_encoder(k, gaps, canonicalize=canon, spaced_intended=True)
            c2.set_minimizer_span(span)
            c2.load_table(w.n_buckets, w.flags, w.keys, w.vals, layout=layout)
            c2.load_taxonomy(w.parent)
            g2 = c2.classify(bases, offsets, paired=paired)
            present = w.keys[[i for i in range(w.n_buckets) if ((int(w.flags[i >> 4]) >> ((i & 15) << 1)) & 3) == 0]]
            gv, gf = c2.probe(present)
            print("present keys not found by probe:", int((gf == 0).sum()), "of", present.size, c2.table_stats(), "genome_len", glen)
            inc = 2 if paired else 1
            for uu in bad[:2]:
                for m in range(inc):
                    rd = reads[int(uu) * inc + m]
                    ge = c2.encode(rd, np.array([0, rd.size], dtype=np.uint64))[0]
                    oe = O.encode(rd.tobytes(), k, gaps=gaps, canon=canon, spaced_intended=True)
                    print("unit", int(uu), "mate", m, "len", rd.size, "read", rd.tobytes()[:300], "| encode equal:", bool(np.array_equal(ge, oe)), ge.size, oe.size)
            print("fresh context agrees with oracle:", bool(np.array_equal(g2[key], exp[key])), "| previous configuration:", prev_cfg)
            sys.exit(1)
    for u, h in enumerate(got["hits"][:50]):
        cut = np.flatnonzero(np.r_[True, h[1:] != h[:-1]]) if h.size else np.zeros(0, np.int64)
        if not np.array_equal(gr["runs"][u][0], h[cut]):
            print("RUNS MISMATCH seed", seed); sys.exit(1)
    prev_cfg = dict(k=k, canon=canon, gaps=gaps, layout=layout, paired=paired, length=length)
    it += 1
    n_reads_total += len(reads)
print("fuzz ok: %d configurations, %d reads, %.0f s" % (it, n_reads_total, time.time() - t0))
