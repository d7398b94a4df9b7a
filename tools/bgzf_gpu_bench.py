#!/usr/bin/env python
"""BGZF input of `bonsai classify`: members inflated on CPU threads against members inflated on the GPU as well (BNS_BGZF_GPU=1:
bns_inflate_members), on the box's CPU quota and on four CPUs; reads/s end to end (process wall time) and the reader's own timers; same taxa either way.
usage (GPU box): python tools/bgzf_gpu_bench.py [n_reads=16000000]"""
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
    d = "/tmp/bgzfbench"; os.makedirs(d, exist_ok=True)
    w = synth.make_world(O, seed=3, k=31, genome_len=50000)
    O.db_write(d + "/bns.db", 31, 31, None, w.table)
    synth.write_nodes_dmp(d + "/nodes.dmp")
    g = np.concatenate(list(w.genomes.values()))
    rng = np.random.default_rng(1)
    fq = d + "/r.fq"
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
    t0 = time.time()
    with Pool(min(16, os.cpu_count() or 1)) as p:
        ms = p.map(member, [data[i:i + 65280] for i in range(0, len(data), 65280)], chunksize=256)
    bg = d + "/r.bgzf.fq.gz"
    with open(bg, "wb") as f:
        f.write(b"".join(ms) + member(b""))
    del ms, data
    print("%d reads, %.1f GB of text, bgzf %.2f of it (%.0f s to write)" % (n, n * 314 / 1e9, os.path.getsize(bg) / (n * 314), time.time() - t0), flush=True)

    def run(tag, args, env):
        e = dict(os.environ, BNS_CLI_TIMING="1"); e.update(env)
        t = time.time()
        p = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
        dt = time.time() - t
        tl = [l[9:] for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and ("process_dataset" in l or "BGZF on" in l or "reader:" in l)]
        print("%-40s rc %d  %6.2f s wall = %6.2f M reads/s | %s" % (tag, p.returncode, dt, n / dt / 1e6, " | ".join(tl)), flush=True)
        if p.returncode:
            print(p.stderr.decode()[-600:])
    GPU = {"BNS_BGZF_GPU": "1"}
    cls = [BIN, "classify", "-a", "-p", "4", "-o", d + "/out.txt", d + "/bns.db", d + "/nodes.dmp"]
    clsK = [BIN, "classify", "-K", "-p", "4", d + "/bns.db", d + "/nodes.dmp"]
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
    if len(sys.argv) > 2 and sys.argv[2] == "cpus":                # where does the device start to pay?  the same on 2 / 4 / 6 / 8 / 12 CPUs
        for ncpu in (2, 4, 6, 8, 12):
            k = ["taskset", "-c", "0-%d" % (ncpu - 1)] + clsK[:2] + ["-N"] + clsK[2:]
            for rep in range(2):
                run("%2d CPUs: BGZF, -K, CPU inflaters" % ncpu, k + [bg], {})
                run("%2d CPUs: BGZF, -K, CPU + GPU" % ncpu, k + [bg], GPU)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "few":                 # GPU threads x batch on a host of 4 and of 8 CPUs
        for ncpu in (4, 8, 12):
            k = ["taskset", "-c", "0-%d" % (ncpu - 1)] + clsK[:2] + ["-N"] + clsK[2:]
            run("%d CPUs: plain FASTQ, -K" % ncpu, k + [fq], {})
            run("%d CPUs: BGZF, -K, CPU inflaters" % ncpu, k + [bg], {"BNS_BGZF_GPU": "0"})
            for rep in range(2):
                run("%d CPUs: plain FASTQ, -K, spinning waits" % ncpu, k + [fq], {"BNS_BLOCKING_SYNC": "0"})
                run("%d CPUs: plain FASTQ, -K, blocking waits" % ncpu, k + [fq], {"BNS_BLOCKING_SYNC": "1"})
                run("%d CPUs: BGZF, -K, device (default), spinning waits" % ncpu, k + [bg], dict(GPU, BNS_BLOCKING_SYNC="0"))
                run("%d CPUs: BGZF, -K, device (default), blocking waits" % ncpu, k + [bg], dict(GPU, BNS_BLOCKING_SYNC="1"))
        return
    if len(sys.argv) > 2 and sys.argv[2] == "big":                 # the device alone on all the box's CPUs: larger batches (more members in flight per call)
        for rep in range(2):
            run("BGZF, -K, CPU inflaters", clsK + [bg], {"BNS_BGZF_GPU": "0"})
            run("BGZF, -K, GPU alone (2 x 128)", clsK + [bg], dict(GPU, BNS_GZ_THREADS="0"))
            run("BGZF, -K, CPU (12) + GPU (2 x 128)", clsK + [bg], GPU)
            run("BGZF, Kraken lines, CPU inflaters", cls + [bg], {"BNS_BGZF_GPU": "0"})
            run("BGZF, Kraken lines, CPU (12) + GPU (2 x 128)", cls + [bg], GPU)
            k4 = ["taskset", "-c", "0-3"] + clsK[:2] + ["-N"] + clsK[2:]
            run("4 CPUs: BGZF, -K, device (default)", k4 + [bg], {})
        return
    if len(sys.argv) > 2 and sys.argv[2] == "threads":             # a large file: how many GPU threads beside the CPU inflaters
        for rep in range(2):
            for thr, b, cpu in ((2, 128, None), (3, 128, None), (4, 128, None), (3, 128, 10), (4, 128, 8), (3, 256, None)):
                e = dict(GPU, BNS_BGZF_GPU_THREADS=str(thr), BNS_BGZF_GPU_BATCH=str(b))
                if cpu is not None: e["BNS_GZ_THREADS"] = str(cpu)
                run("BGZF, -K, CPU (%s) + GPU (%d x %d)" % ("12" if cpu is None else cpu, thr, b), clsK + [bg], e)
            run("BGZF, Kraken lines, CPU inflaters", cls + [bg], {"BNS_BGZF_GPU": "0"})
            run("BGZF, Kraken lines, CPU (12) + GPU (2 x 128)", cls + [bg], GPU)
            run("plain FASTQ, -K", clsK + [fq], {})
        return
    scan = len(sys.argv) > 2 and sys.argv[2] == "scan"
    if scan:                                                       # how many CPU inflaters beside how many GPU threads
        for rep in range(2):
            run("BGZF, -K, CPU inflaters", clsK + [bg], {})
            for cpu, thr, b in ((8, 2, 128), (8, 3, 128), (6, 3, 128), (6, 3, 64), (4, 3, 128), (8, 3, 64), (10, 2, 128)):
                run("BGZF, -K, CPU (%d) + GPU (%d x %d)" % (cpu, thr, b), clsK + [bg], dict(GPU, BNS_GZ_THREADS=str(cpu), BNS_BGZF_GPU_THREADS=str(thr), BNS_BGZF_GPU_BATCH=str(b)))
        return
    for rep in range(2):
        run("plain FASTQ, -K", clsK + [fq], {})
        run("BGZF, -K, CPU inflaters", clsK + [bg], {})
        run("BGZF, -K, CPU + GPU", clsK + [bg], GPU)
        run("BGZF, -K, CPU + GPU, one parser", clsK + [bg], dict(GPU, BNS_BGZF_ONE_PARSER="1"))
        run("BGZF, -K, GPU only (3 threads)", clsK + [bg], dict(GPU, BNS_GZ_THREADS="0", BNS_BGZF_GPU_THREADS="3"))
        run("BGZF, Kraken lines, CPU inflaters", cls + [bg], {})
        run("BGZF, Kraken lines, CPU + GPU", cls + [bg], GPU)
        if quick:
            continue
        # a host short of CPUs: the same on four of them (no binding to the GPU's CPUs: -N)
        t4 = ["taskset", "-c", "0-3"]
        k4 = t4 + clsK[:2] + ["-N"] + clsK[2:]
        run("4 CPUs: plain FASTQ, -K", k4 + [fq], {})
        run("4 CPUs: BGZF, -K, CPU inflaters", k4 + [bg], {})
        run("4 CPUs: BGZF, -K, CPU + GPU", k4 + [bg], GPU)
    outs = {}
    for tag, inp, env in (("plain", fq, {}), ("bgzf_cpu", bg, {}), ("bgzf_gpu", bg, {"BNS_BGZF_GPU": "1"})):
        subprocess.run([BIN, "classify", "-K", "-p", "4", "-b", d + "/t_%s.bin" % tag, d + "/bns.db", d + "/nodes.dmp", inp], stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
        outs[tag] = np.fromfile(d + "/t_%s.bin" % tag, dtype="<u4")
    subprocess.run(cls[:6] + [d + "/o_cpu.txt"] + cls[7:] + [bg], stderr=subprocess.DEVNULL)
    subprocess.run(cls[:6] + [d + "/o_gpu.txt"] + cls[7:] + [bg], stderr=subprocess.DEVNULL, env=dict(os.environ, BNS_BGZF_GPU="1"))
    same_text = subprocess.run(["cmp", "-s", d + "/o_cpu.txt", d + "/o_gpu.txt"]).returncode == 0
    print("taxa identical across plain / BGZF on CPU / BGZF on GPU: %s (%d reads, %.3f classified); Kraken output byte-identical CPU vs GPU inflate: %s (%d bytes)" % (
        bool(np.array_equal(outs["plain"], outs["bgzf_cpu"]) and np.array_equal(outs["plain"], outs["bgzf_gpu"])), outs["plain"].size,
        float((outs["plain"] != 0).mean()), same_text, os.path.getsize(d + "/o_gpu.txt")))


if __name__ == "__main__":
    main()
