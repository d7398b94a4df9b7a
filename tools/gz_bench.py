#!/usr/bin/env python
"""One plain gzip stream through the CLI (round 4: inflated on many threads, csrc/host/pgzip.hpp) against the zlib reader
(BNS_NO_PGZ=1): reads/s end to end (process wall time), same taxa.  Qualities random over 40 symbols: the file compresses and
inflates like real FASTQ.   usage (GPU box): python tools/gz_bench.py [n_reads=8000000]"""
import os, subprocess, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
BIN = ROOT + "/bonsai_amd/bin/bonsai"


def main():
    import oracle_lib as O, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    d = "/tmp/gzbench"; os.makedirs(d, exist_ok=True)
    w = synth.make_world(O, seed=3, k=31, genome_len=50000)
    O.db_write(d + "/bns.db", 31, 31, None, w.table)
    synth.write_nodes_dmp(d + "/nodes.dmp")
    g = np.concatenate(list(w.genomes.values()))
    rng = np.random.default_rng(1)
    cos = {1: zlib.compressobj(1, zlib.DEFLATED, 31), 6: zlib.compressobj(6, zlib.DEFLATED, 31)}
    fs = {l: open(d + "/r.l%d.fq.gz" % l, "wb") for l in cos}
    t0 = time.time()
    text = 0
    for s0 in range(0, n, 2_000_000):
        m = min(2_000_000, n - s0)
        st = rng.integers(0, g.size - 150, size=m)
        rec = np.empty((m, 314), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        idx = np.arange(s0, s0 + m)
        for j in range(8):
            rec[:, 9 - j] = ord("0") + (idx // 10 ** j) % 10
        rec[:, 1] = ord("r"); rec[:, 9] = 10
        rec[:, 10:160] = g[st[:, None] + np.arange(150)[None, :]]
        rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
        rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
        rec[:, 313] = 10
        b = rec.tobytes(); text += len(b)
        for l in cos:
            if l == 6 and s0 >= 4_000_000:
                continue                                   # (level 6 compresses at 30 MB/s: half the reads are plenty)
            fs[l].write(cos[l].compress(b))
    for l in cos:
        fs[l].write(cos[l].flush()); fs[l].close()
    n6 = min(n, 4_000_000)
    print("gzip -1 of %d reads: %.2f of the text; gzip -6 of %d reads: %.2f; made in %.0f s" % (
        n, os.path.getsize(d + "/r.l1.fq.gz") / text, n6, os.path.getsize(d + "/r.l6.fq.gz") / (n6 * 314), time.time() - t0), flush=True)

    def run(tag, inp, n_reads, env):
        outs = []
        for rep in range(2):
            t = time.time()
            p = subprocess.run([BIN, "classify", "-K", "-p", "4", "-b", d + "/t.bin", d + "/bns.db", d + "/nodes.dmp", inp], stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, env=dict(os.environ, BNS_CLI_TIMING="1", **env))
            dt = time.time() - t
            tl = [l for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and ("process_dataset" in l or "gzip reader" in l or "reader:" in l)]
            print("%-44s rc %d  %6.2f s wall = %6.2f M reads/s   %s" % (tag, p.returncode, dt, n_reads / dt / 1e6, " | ".join(x[9:] for x in tl)), flush=True)
            if p.returncode:
                print(p.stderr.decode()[-400:])
        return np.fromfile(d + "/t.bin", dtype="<u4")
    res = {}
    for l, nr in ((1, n), (6, n6)):
        inp = d + "/r.l%d.fq.gz" % l
        res[(l, "zlib")] = run("gzip -%d, zlib reader (BNS_NO_PGZ=1)" % l, inp, nr, {"BNS_NO_PGZ": "1"})
        res[(l, "pgz")] = run("gzip -%d, parallel reader (default threads)" % l, inp, nr, {})
        for t in (4, 8, 14):
            res[(l, "pgz%d" % t)] = run("gzip -%d, parallel reader, %d threads" % (l, t), inp, nr, {"BNS_GZ_THREADS": str(t)})
        res[(l, "pgz8m")] = run("gzip -%d, parallel reader, 8 MiB chunks" % l, inp, nr, {"BNS_PGZ_CHUNK": str(8 << 20)})
    ok = all(np.array_equal(res[(l, "zlib")], v) for (l, k), v in res.items())
    print("taxa identical, zlib reader vs parallel reader (every setting): %s (%d / %d reads)" % (ok, res[(1, "zlib")].size, res[(6, "zlib")].size))


main()
