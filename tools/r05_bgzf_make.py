#!/usr/bin/env python
"""(file maker only) round 5: `bonsai classify` on a BGZF FASTQ with the members' text left on the device (process_bgzf_gpu) against the host reader
(BNS_TEXT_GPU=0: CPU inflaters, and the device inflating beside them with the text copied back).  Random qualities over 40 symbols
(what makes DEFLATE work for its ratio).  usage (GPU box): python tools/r05_bgzf.py [n_reads=96000000]"""
import os, struct, subprocess, sys, time, zlib
from multiprocessing import Pool
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
BIN = ROOT + "/bonsai_amd/bin/bonsai"


def member(chunk):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(chunk) + co.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body
            + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def main():
    import oracle_lib as O, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96_000_000
    d = "/tmp/bgzfbench"; os.makedirs(d, exist_ok=True)
    w = synth.make_world(O, seed=3, k=31, genome_len=50000)
    O.db_write(d + "/bns.db", 31, 31, None, w.table)
    synth.write_nodes_dmp(d + "/nodes.dmp")
    g = np.concatenate(list(w.genomes.values()))
    rng = np.random.default_rng(1)
    bg = d + "/r.bgzf.fq.gz"
    t0 = time.time()
    carry = b""
    with open(bg, "wb") as f, Pool(min(16, os.cpu_count() or 1)) as p:
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
            rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
            rec[:, 313] = 10
            data = carry + rec.tobytes()
            cut = len(data) - len(data) % 65280 if s0 + m < n else len(data)
            f.write(b"".join(p.map(member, [data[i:i + 65280] for i in range(0, cut, 65280)], chunksize=64)))
            carry = data[cut:]
        f.write(member(b""))
    print("%d reads, %.1f GB of text, bgzf %.2f of it (%.0f s to write)" % (n, n * 314 / 1e9, os.path.getsize(bg) / (n * 314), time.time() - t0), flush=True)
    subprocess.run(["cat", bg], stdout=subprocess.DEVNULL)



if __name__ == "__main__":
    main()
