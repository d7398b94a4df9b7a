#!/usr/bin/env python
"""The read-container path of the CLI against the container's chunk size, the loader threads and the contexts per device: a chunk
below the host entry point's slice size (64 MiB of words) goes up, is classified and comes back one step after the other; a chunk of
several slices overlaps the three inside the call.
usage (GPU box): python tools/container_scan.py [n_reads=8000000] [copies=16]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
BIN = ROOT + "/bonsai_amd/bin/bonsai"


def main():
    import oracle_lib as O, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    d = "/tmp/containerscan"; os.makedirs(d, exist_ok=True)
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
    print("fastq: %d reads" % n, flush=True)

    def run(tag, args, n_reads, env=None):
        e = dict(os.environ, BNS_CLI_TIMING="1")
        e.update(env or {})
        t = time.time()
        p = subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
        dt = time.time() - t
        tl = [l for l in p.stderr.decode().splitlines() if l.startswith("[timing]") and ("process_dataset" in l or "pack +" in l)]
        print("%-44s rc %d  %6.2f s wall = %7.2f M reads/s   %s" % (tag, p.returncode, dt, n_reads / dt / 1e6, " | ".join(x[9:] for x in tl)), flush=True)
        if p.returncode:
            print(p.stderr.decode()[-500:])

    ref = None
    for lg in (27, 29, 30, 31):
        pk = d + "/r_%d.bnsp" % lg
        t = time.time()
        subprocess.run([BIN, "pack", "-n", "-p", "8", "-c", str(1 << lg), "-o", pk, fq], stderr=subprocess.PIPE, check=True)
        print("pack -c 2^%d: %.2f s" % (lg, time.time() - t), flush=True)
        body = open(pk, "rb").read()
        big = d + "/big_%d.bnsp" % lg
        with open(big, "wb") as f:
            f.write(body[:32])
            for _ in range(copies):
                f.write(body[32:])
        del body
        clsK = [BIN, "classify", "-K", "-p", "4", d + "/bns.db", d + "/nodes.dmp", big]
        for rep in range(2):
            run("chunks of 2^%d bases, -K" % lg, clsK, copies * n)
        for np_ in (2, 6, 8):
            run("chunks of 2^%d bases, -K, %d loaders" % (lg, np_), clsK, copies * n, {"BNS_CLI_PACKERS": str(np_)})
        run("chunks of 2^%d bases, -K, -g 0,0" % lg, clsK[:2] + ["-g", "0,0"] + clsK[2:], copies * n)
        # same answers whatever the chunk size
        subprocess.run([BIN, "classify", "-K", "-p", "4", "-b", d + "/t.bin", d + "/bns.db", d + "/nodes.dmp", pk], stderr=subprocess.DEVNULL, check=True)
        t_ = np.fromfile(d + "/t.bin", dtype="<u4")
        if ref is None:
            ref = t_
        print("  taxa identical to the 2^27 container's: %s (%d reads, %.3f classified)" % (bool(np.array_equal(ref, t_)), t_.size, float((t_ != 0).mean())), flush=True)
        os.remove(big)


if __name__ == "__main__":
    main()
