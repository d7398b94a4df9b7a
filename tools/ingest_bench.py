#!/usr/bin/env python
"""Ingest paths of the CLI side by side: the same reads as plain FASTQ, BGZF, one-stream gzip and as a
`bonsai pack` container -> `bonsai classify`, reads/s end to end (process wall time).  Qualities are random over 40 symbols, so
the gzip files compress (and inflate) like real FASTQ, not like a constant string.
usage (GPU box): python tools/ingest_bench.py [n_reads=16000000]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    d = "/tmp/ingestbench"; os.makedirs(d, exist_ok=True)
    w = synth.make_world(O, seed=3, k=31, genome_len=50000)
    O.db_write(d + "/bns.db", 31, 31, None, w.table)
    synth.write_nodes_dmp(d + "/nodes.dmp")
    g = np.concatenate(list(w.genomes.values()))
    rng = np.random.default_rng(1)
    fq = d + "/r.fq"
    t0 = time.time()
    with open(fq, "wb") as f:
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
            rec.tofile(f)
    data = open(fq, "rb").read()
    print("fastq %d reads, %.1f GB in %.0f s" % (n, len(data) / 1e9, time.time() - t0), flush=True)
    t0 = time.time()
    with Pool(min(16, os.cpu_count() or 1)) as p:
        ms = p.map(member, [data[i:i + 65280] for i in range(0, len(data), 65280)], chunksize=256)
    with open(d + "/r.bgzf.fq.gz", "wb") as f:
        f.write(b"".join(ms) + member(b""))
    del ms
    print("bgzf: %.2f of the text, %.0f s" % (os.path.getsize(d + "/r.bgzf.fq.gz") / len(data), time.time() - t0), flush=True)
    n_gz = min(n, 4_000_000)                          # one-stream gzip: a quarter is plenty (it runs at ~1 M reads/s)
    t0 = time.time()
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(d + "/r.plain.fq.gz", "wb") as f:
        f.write(co.compress(data[:n_gz * 314])); f.write(co.flush())
    print("gzip -1 of %d reads: %.0f s" % (n_gz, time.time() - t0), flush=True)
    del data

    def run(tag, args, n_reads):
        t = time.time()
        p = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, BNS_CLI_TIMING="1"))
        dt = time.time() - t
        tl = [l for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and ("process_dataset" in l or "pack +" in l)]
        print("%-34s rc %d  %6.2f s wall = %7.2f M reads/s   %s" % (tag, p.returncode, dt, n_reads / dt / 1e6, " | ".join(x[9:] for x in tl)), flush=True)
        if p.returncode:
            print(p.stderr.decode()[-500:])
    cls = [BIN, "classify", "-a", "-p", "4", "-o", d + "/out.txt", d + "/bns.db", d + "/nodes.dmp"]
    clsK = [BIN, "classify", "-K", "-p", "4", d + "/bns.db", d + "/nodes.dmp"]
    t = time.time()
    p = subprocess.run([BIN, "pack", "-p", "8", "-o", d + "/r.bnsp", fq], stderr=subprocess.PIPE)
    dt = time.time() - t
    print("bonsai pack (plain FASTQ -> container, %.2f of the text): %.2f s = %.1f M reads/s" % (os.path.getsize(d + "/r.bnsp") / (n * 314), dt, n / dt / 1e6), flush=True)
    t = time.time()
    subprocess.run([BIN, "pack", "-p", "8", "-o", d + "/r2.bnsp", d + "/r.bgzf.fq.gz"], stderr=subprocess.PIPE)
    dt = time.time() - t
    print("bonsai pack (BGZF -> container): %.2f s = %.1f M reads/s" % (dt, n / dt / 1e6), flush=True)
    os.remove(d + "/r2.bnsp")
    # steady state of the container path: the same chunks eight times over (chunks are self-contained)
    body = open(d + "/r.bnsp", "rb").read()
    with open(d + "/r8.bnsp", "wb") as f:
        f.write(body[:32])
        for _ in range(8):
            f.write(body[32:])
    del body
    for rep in range(2):
        run("container x8, -K", clsK + [d + "/r8.bnsp"], 8 * n)
        run("container x8, -K -b taxa.bin", clsK[:3] + ["-b", d + "/taxa.bin"] + clsK[3:] + [d + "/r8.bnsp"], 8 * n)
    run("container x8, Kraken lines", cls + [d + "/r8.bnsp"], 8 * n)
    for rep in range(2):
        run("plain FASTQ, Kraken lines", cls + [fq], n)
        run("plain FASTQ, -K", clsK + [fq], n)
        run("BGZF, Kraken lines", cls + [d + "/r.bgzf.fq.gz"], n)
        run("BGZF, -K", clsK + [d + "/r.bgzf.fq.gz"], n)
        run("container, Kraken lines", cls + [d + "/r.bnsp"], n)
        run("container, -K", clsK + [d + "/r.bnsp"], n)
        run("container, -K -b taxa.bin", clsK[:3] + ["-b", d + "/taxa.bin"] + clsK[3:] + [d + "/r.bnsp"], n)
    run("one gzip stream, -K", clsK + [d + "/r.plain.fq.gz"], n_gz)
    # same answers whatever the input form
    outs = {}
    for tag, inp in (("plain", fq), ("bgzf", d + "/r.bgzf.fq.gz"), ("pack", d + "/r.bnsp")):
        subprocess.run([BIN, "classify", "-K", "-p", "4", "-b", d + "/t_%s.bin" % tag, d + "/bns.db", d + "/nodes.dmp", inp], stderr=subprocess.DEVNULL)
        outs[tag] = np.fromfile(d + "/t_%s.bin" % tag, dtype="<u4")
    print("taxa identical across plain / BGZF / container: %s (%d reads, %.3f classified)" % (
        bool(np.array_equal(outs["plain"], outs["bgzf"]) and np.array_equal(outs["plain"], outs["pack"])), outs["plain"].size, float((outs["plain"] != 0).mean())))


if __name__ == "__main__":
    main()
