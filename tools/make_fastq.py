#!/usr/bin/env python
"""A large synthetic FASTQ (150 bp, 314 bytes per record: the file of tools/cli_bench.py) written in pieces, for CLI runs that are long
enough for the steady state to show: make_fastq.py <n_reads> <path> [world_genome_len]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, synth
n = int(sys.argv[1]); path = sys.argv[2]
d = os.path.dirname(path)
w = synth.make_world(O, seed=3, k=31, genome_len=50000)
O.db_write(d + "/bns.db", 31, 31, None, w.table)
synth.write_nodes_dmp(d + "/nodes.dmp")
if os.path.exists(path) and os.path.getsize(path) == n * 314:
    sys.exit(0)
g = np.concatenate(list(w.genomes.values()))
rng = np.random.default_rng(1)
step = 8_000_000
with open(path, "wb") as f:
    for a in range(0, n, step):
        m = min(step, n - a)
        st = rng.integers(0, g.size - 150, size=m)
        rec = np.empty((m, 314), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        idx = (np.arange(m) + a) % 10_000_000
        for j in range(7):
            rec[:, 8 - j] = ord("0") + (idx // 10 ** j) % 10
        rec[:, 9] = 10
        rec[:, 10:160] = g[st[:, None] + np.arange(150)[None, :]]
        rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
        rec[:, 163:313] = ord("I"); rec[:, 313] = 10
        rec.tofile(f)
