#!/usr/bin/env python
"""(file maker) round 6: a FASTQ file as ONE gzip member, the way pigz writes it: pieces of 4 MiB of text deflated side by side (level 6), each
primed with the 32 KiB in front of it (so matches reach across the pieces, as in a stream gzip itself wrote) and ended with a sync flush;
one header, one trailer.  Qualities: `random` = uniform over 40 symbols (what the BGZF benchmarks use: DEFLATE works for its ratio, 0.51), or
`binned` = eight quality values in runs (what current instruments write: 0.27).
usage (GPU box): python tools/r06_gz_make.py [n_reads=64000000] [random|binned]  -> /tmp/gzbench/r.<qual>.fq.gz (+ bns.db, nodes.dmp)"""
import os, struct, subprocess, sys, time, zlib
from multiprocessing import Pool
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def piece(args):
    data, zdict, last = args
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, zdict) if zdict else zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(data) + (co.flush(zlib.Z_FINISH) if last else co.flush(zlib.Z_SYNC_FLUSH))
    return body, zlib.crc32(data) & 0xFFFFFFFF, len(data)


def main():
    import oracle_lib as O, synth
    import bonsai_amd
    lib = bonsai_amd.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
    qual = sys.argv[2] if len(sys.argv) > 2 else "random"
    d = "/tmp/gzbench"; os.makedirs(d, exist_ok=True)
    w = synth.make_world(O, seed=3, k=31, genome_len=50000)
    O.db_write(d + "/bns.db", 31, 31, None, w.table)
    synth.write_nodes_dmp(d + "/nodes.dmp")
    g = np.concatenate(list(w.genomes.values()))
    rng = np.random.default_rng(1)
    path = d + "/r.%s.fq.gz" % qual
    t0 = time.time()
    crc, total = 0, 0
    prev = b""
    PIECE = 4 << 20
    bins = np.frombuffer(b"#,5:AFIJ", dtype=np.uint8)
    with open(path, "wb") as f, Pool(min(16, os.cpu_count() or 1)) as p:
        f.write(b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03")
        carry = b""
        for s0 in range(0, n, 4_000_000):
            m = min(4_000_000, n - s0)
            st = rng.integers(0, g.size - 150, size=m)
            rec = np.empty((m, 314), dtype=np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            idx = np.arange(s0, s0 + m) % 100_000_000
            for j in range(8):
                rec[:, 9 - j] = ord("0") + (idx // 10 ** j) % 10
            rec[:, 1] = ord("r"); rec[:, 9] = 10
            rec[:, 10:160] = g[st[:, None] + np.arange(150)[None, :]]
            rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
            if qual == "random":
                rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
            else:
                # runs of ~6 equal values, mostly the top bins
                q = rng.choice(8, size=(m, 25), p=[0.02, 0.03, 0.05, 0.05, 0.1, 0.2, 0.25, 0.3])
                rec[:, 163:313] = bins[np.repeat(q, 6, axis=1)]
            rec[:, 313] = 10
            data = carry + rec.tobytes()
            final = s0 + m >= n
            cut = len(data) if final else len(data) - len(data) % PIECE
            jobs = []
            for i in range(0, cut, PIECE):
                chunk = data[i:i + PIECE]
                jobs.append((chunk, prev, final and i + PIECE >= cut))
                prev = chunk[-32768:]
            for body, c, ln in p.imap(piece, jobs, chunksize=4):
                f.write(body)
                crc = lib.bns_crc32_combine(crc, c, ln); total += ln
            carry = data[cut:]
        f.write(struct.pack("<II", crc, total & 0xFFFFFFFF))
    print("%d reads, %.1f GB of text, gzip %.2f of it (%.0f s to write): %s" % (n, total / 1e9, os.path.getsize(path) / total, time.time() - t0, path), flush=True)
    subprocess.run(["cat", path], stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    main()
