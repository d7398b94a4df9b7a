#!/usr/bin/env python
"""Randomised end-to-end soak of `bonsai classify` (reader + GPU + device run encoding + formatter) against lines built from
the oracle: FASTQ / FASTA / multi-line / CRLF / .gz / BGZF inputs and `bonsai pack` containers, single and paired, -a, chunk sizes that cut the input into many
bseq_read chunks, all three layouts, -P stretches of a few KB parsed side by side.  usage: tools/fuzz_cli.py [seconds] [seed]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402   (the checker)
import synth                    # noqa: E402

O.build()
BIN = os.path.join(ROOT, "bonsai_amd", "bin", "bonsai")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = tempfile.mkdtemp(prefix="fuzzcli")
w = synth.make_world(O, seed=3, k=31, genome_len=6000)
db = os.path.join(d, "bns.db"); nodes = os.path.join(d, "nodes.dmp")
O.db_write(db, 31, 31, None, w.table, spacing_width=1)
synth.write_nodes_dmp(nodes)


def write_reads(path, names, reads, rng, mate):
    fastq = rng.random() < 0.7
    eol = b"\r\n" if rng.random() < 0.2 else b"\n"
    width = int(rng.choice([0, 0, 50, 70]))
    parts = []
    for nm, r in zip(names, reads):
        s = r.tobytes()
        hdr = nm + (b"/%d" % mate if rng.random() < 0.5 else b"") + (b" some comment" if rng.random() < 0.3 else b"")
        if fastq:
            # (round 6: quality over several lines, its lines starting with whatever -- '@', '+', '>' among them -- and sequences over several lines)
            q = bytes(rng.choice(np.frombuffer(b"I@+>F#5", dtype=np.uint8), size=len(s)).astype(np.uint8)) if rng.random() < 0.5 else b"I" * len(s)
            qw = int(rng.choice([0, 0, 0, 37, 60]))
            body = eol.join(s[j:j + width] for j in range(0, len(s), width)) if width and s and qw else s
            qual = eol.join(q[j:j + qw] for j in range(0, len(q), qw)) if qw and q else q
            parts.append(b"@" + hdr + eol + body + eol + b"+" + eol + qual + eol)
        else:
            body = eol.join(s[j:j + width] for j in range(0, len(s), width)) if width and s else s
            parts.append(b">" + hdr + eol + body + eol)
    data = b"".join(parts)
    form = rng.random()
    if form < 0.2:
        path += ".gz"
        with gzip.open(path, "wb", compresslevel=int(rng.choice([1, 6, 9]))) as f:
            f.write(data)
    elif form < 0.4:                                                 # BGZF: members of assorted sizes, inflated side by side
        path += ".bgzf.gz"
        synth.write_bgzf(path, data, member_sizes=[int(x) for x in rng.integers(1, 4000, size=7)] + [65280], level=int(rng.choice([1, 6])))
    else:
        with open(path, "wb") as f:
            f.write(data)
    return path


t0 = time.time()
it = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 100003 + it)
    paired = rng.random() < 0.4
    n = int(rng.integers(1, 400))
    reads1 = synth.simulate_reads(rng, w.genomes, n, length=int(rng.choice([60, 150, 300])), var_len=True, n_rate=0.004, lower_rate=0.05)
    reads1 = [r for r in reads1 if r.size > 0]                      # an empty FASTQ sequence line ends kseq's record early
    reads2 = [synth.simulate_reads(rng, w.genomes, 1, length=150, var_len=True)[0] for _ in reads1] if paired else None
    names = [b"r%d" % i for i in range(len(reads1))]
    p1 = write_reads(os.path.join(d, "a_%d" % it), names, reads1, rng, 1)
    args = ["classify"]
    emit_all = rng.random() < 0.6
    if emit_all: args.append("-a")
    args += ["-c", str(int(rng.choice([200, 5000, 1 << 20])))]
    args += ["-L", str(rng.choice(["minbucket", "minbucket", "bucket", "khash"]))]
    args += ["-P", str(rng.choice(["1", "2", "2:2000", "3:5000", "2:20000", "4:700"])), "-p", str(int(rng.integers(1, 5)))]   # stretches parsed side by side
    inputs = [p1]
    if paired:
        inputs.append(write_reads(os.path.join(d, "b_%d" % it), names, reads2, rng, 2))
    if rng.random() < 0.3:                                           # through `bonsai pack`: the container instead of the text
        pk = os.path.join(d, "a_%d.bnsp" % it)
        pp = subprocess.run([BIN, "pack", "-o", pk, "-c", str(int(rng.choice([300, 9000, 1 << 27]))), "-p", str(int(rng.integers(1, 4)))] + inputs,
                            stderr=subprocess.PIPE, timeout=120)
        if pp.returncode != 0:
            print("PACK FAILED seed", seed0 * 100003 + it, pp.stderr.decode()[-300:]); sys.exit(1)
        inputs = [pk]
    devices = str(rng.choice(["0", "0", "0,0", "0,0,0"]))            # round 6: several contexts on the one device (blocks cut in order / guessed and verified)
    if devices != "0": args += ["-g", devices]
    args += [db, nodes] + inputs
    env = dict(os.environ)
    # the device text paths: blocks of a few KB, BGZF batches of a few members, little room for what a batch leaves, the two-device copy path
    if rng.random() < 0.6: env["BNS_TEXT_BLOCK_BYTES"] = str(int(rng.choice([700, 3000, 50000])))
    if rng.random() < 0.6: env["BNS_BGZF_BATCH_MEMBERS"] = str(int(rng.choice([1, 2, 5])))
    if rng.random() < 0.3: env["BNS_BGZF_HEAD_BYTES"] = str(int(rng.choice([4096, 20000])))
    if rng.random() < 0.3: env["BNS_PEER_VIA_HOST"] = "1"
    if rng.random() < 0.15: env["BNS_TEXT_GPU"] = "0"
    # one plain gzip stream on the device (round 6): chunks of a few KB, calls of a few dozen KB, little room for text, chunks with too little
    # room for a block's symbols (the device gives up: the host reader's), the host reader outright
    if rng.random() < 0.7: env["BNS_GZ_CHUNK_KB"] = str(int(rng.choice([4, 8, 64])))
    if rng.random() < 0.8: env["BNS_GZ_RATIO_CAP"] = str(int(rng.choice([2, 16, 400, 400])))
    if rng.random() < 0.3: env["BNS_GZ_PIECE_BYTES"] = str(int(rng.choice([65536, 100000])))
    if rng.random() < 0.3: env["BNS_GZ_TEXT_BYTES"] = str(int(rng.choice([70000, 300000])))
    if rng.random() < 0.1: env["BNS_GZ_GPU"] = "0"
    if rng.random() < 0.3: env["BNS_GZ_ROOM_RETRY"] = "0"
    if any(x.endswith(".bgzf.gz") for x in inputs):                  # BGZF: small text blocks (many tasks, stretches of the inflated text), the device inflating
        if rng.random() < 0.7: env["BNS_READER_BLOCK"] = str(int(rng.choice([3000, 20000, 70000])))
        if rng.random() < 0.6:
            env["BNS_BGZF_GPU"] = "1"
            env["BNS_BGZF_GPU_BATCH"] = str(int(rng.choice([1, 3, 128])))
            env["BNS_BGZF_GPU_THREADS"] = str(int(rng.integers(1, 4)))
            if rng.random() < 0.5: env["BNS_GZ_THREADS"] = str(int(rng.choice([0, 0, 2])))
        if rng.random() < 0.2: env["BNS_BGZF_NO_MMAP"] = "1"
    p = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    exp = []
    for i, r in enumerate(reads1):
        t, m, a, hits = O.classify_seq(w.table, w.tax, 31, r.tobytes(), reads2[i].tobytes() if paired else None)
        if t or emit_all:
            exp.append(O.kraken_line(names[i].decode(), t, r.size, m, a, hits))
    if p.returncode != 0 or p.stdout != b"".join(exp):
        print("CLI MISMATCH seed", seed0 * 100003 + it, "args", args, "rc", p.returncode, "out bytes", len(p.stdout), "expected", len(b"".join(exp)),
              "env", {k: v for k, v in env.items() if k.startswith("BNS_")})
        print(p.stderr.decode()[-400:])
        sys.exit(1)
    for f in os.listdir(d):
        if f.startswith(("a_", "b_")):
            os.remove(os.path.join(d, f))
    it += 1
print("cli fuzz ok: %d invocations, %.0f s" % (it, time.time() - t0))
