#!/usr/bin/env python
"""End-to-end CLI throughput (host ingest + GPU + formatting) on a synthetic FASTQ.  IO-bound by design."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
extra = sys.argv[2:]
paired = "--paired" in extra                                 # two files of n / 2 mates each
extra = [a for a in extra if a != "--paired"]
d = "/tmp/clibench"; os.makedirs(d, exist_ok=True)
w = synth.make_world(O, seed=3, k=31, genome_len=50000)
O.db_write(d + "/bns.db", 31, 31, None, w.table)
synth.write_nodes_dmp(d + "/nodes.dmp")
g = np.concatenate(list(w.genomes.values()))
rng = np.random.default_rng(1)
st = rng.integers(0, g.size - 150, size=n)
seqs = g[st[:, None] + np.arange(150)[None, :]]
fq = d + "/r.fq"
if not os.path.exists(fq) or os.path.getsize(fq) != n * 314:
    rec = np.empty((n, 314), dtype=np.uint8)                     # "@r0000000\n" + 150 bases + "\n+\n" + 150 quals + "\n"
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(n)
    for j in range(7):
        rec[:, 8 - j] = ord("0") + (idx // 10 ** j) % 10
    rec[:, 9] = 10
    rec[:, 10:160] = seqs
    rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
    rec[:, 163:313] = ord("I"); rec[:, 313] = 10
    rec.tofile(fq)
    rec[: n // 2].tofile(d + "/r_1.fq"); rec[n // 2:].tofile(d + "/r_2.fq")
    del rec
for args in (["-p", "4"], ["-K", "-p", "4"]):
    t0 = time.time()
    p = subprocess.run(os.environ.get("BNS_CLI_PREFIX", "").split() + [ROOT + "/bonsai_amd/bin/bonsai", "classify", "-a"] + list(args) + extra + ["-o", d + "/out.txt", d + "/bns.db", d + "/nodes.dmp"] + ([d + "/r_1.fq", d + "/r_2.fq"] if paired else [fq]),
                       stderr=subprocess.PIPE, env=dict(os.environ, BNS_CLI_TIMING="1"))
    dt = time.time() - t0
    tl = [l for l in p.stderr.decode().splitlines() if l.startswith("[timing]")]
    print(tl)
    su = 0.0
    for l in tl:
        if "start-up" in l:
            su = float(l.split(")")[1].split()[0])
    print("args", args + extra, "rc", p.returncode, "%.2f s wall = %.2f M reads/s; without the %.2f s start-up %.2f M reads/s; out %.1f MB"
          % (dt, n / dt / 1e6, su, n / max(1e-9, dt - su) / 1e6, os.path.getsize(d + "/out.txt") / 1e6))
