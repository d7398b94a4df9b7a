#!/usr/bin/env python
"""Stress of the minimizer-clustered layout: many genomes share a conserved core with random flanks, so thousands of
distinct k-mers share one minimizer (oversized groups -> long spill chains).  Checks parity and reports timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bonsai_amd, oracle_lib as O, synth
n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(5)
core = synth.rand_seq(rng, 60)
pairs = [(1, 1)] + [(10 + i, 1) for i in range(n_gen)]
tax = O.Taxonomy(pairs=pairs)
table = O.Table()
genomes = []
t0 = time.time()
for i in range(n_gen):
    g = np.concatenate([synth.rand_seq(rng, 45), core, synth.rand_seq(rng, 45)])
    genomes.append(g)
    O.lca_map_add(table, tax, 31, g.tobytes(), 10 + i)
print("oracle build %.1fs, keys %d" % (time.time() - t0, table.header()[1]))
flags, keys, vals = table.arrays()
reads = [genomes[int(rng.integers(n_gen))] for _ in range(20000)]
bases, offsets = synth.concat(reads)
exp = O.classify_batch(table, tax, 31, bases, offsets, nthreads=8)
for layout, name in ((2, "minbucket"), (1, "bucket")):
    ctx = bonsai_amd.Context(0)
    ctx.set_encoder(31, None, True)
    t0 = time.time(); ctx.load_table(table.n_buckets, flags, keys, vals, layout=layout); t1 = time.time()
    ctx.load_taxonomy(tax.parent)
    ctx.classify(bases, offsets)
    t2 = time.time(); got = ctx.classify(bases, offsets); t3 = time.time()
    ok = np.array_equal(got["taxon"], exp["taxon"]) and np.array_equal(got["missing"], exp["missing"])
    print("%-9s load %.3fs classify(20k reads) %.4fs parity %s" % (name, t1 - t0, t3 - t2, ok))
    ctx.close()
