#!/usr/bin/env python
"""Randomised soak of the host reader's BGZF paths WITHOUT a GPU: member sizes, text-block sizes, stretches on 1-4 parser threads, the
GPU-inflate threads against the CPU stand-in of tests/helpers/inflater_shim.cpp (LD_PRELOAD, random answer latency), queue shares,
mapped / pread splitter -- records against those of the plain text.  usage: tools/dev/bgzf_reader_stress.py [seconds] [seed]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import synth
from bonsai_amd import hostio

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = tempfile.mkdtemp(prefix="bgzfstress")
so = os.path.join(d, "libshim.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", ROOT + "/tests/helpers/inflater_shim.cpp", "-o", so], check=True)
CODE = ("import sys; sys.path.insert(0, %r); from bonsai_amd import hostio\n"
        "import hashlib, os\n"
        "if os.environ.get('STRESS_GPU') == '1': hostio.lib().bnsh_set_bgzf_device(0)\n"
        "r, n, fell = hostio.read_fastx_par(sys.argv[1], chunk_size=int(sys.argv[2]), parser_threads=int(sys.argv[3]), segment_bytes=int(sys.argv[4]))\n"
        "print(len(r), hashlib.sha256(repr(r).encode()).hexdigest(), n, fell)\n" % ROOT)
import hashlib
t0 = time.time(); it = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 7919 + it)
    kind = "fastq" if rng.random() < 0.7 else "fasta"
    n = int(rng.integers(1, 6000))
    parts = []
    for i in range(n):
        L = int(rng.integers(1, 400))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L))
        if kind == "fastq":
            parts.append(b"@r%d c\n" % i + seq + b"\n+\n" + bytes(rng.choice(np.frombuffer(b"@>+I#5", dtype=np.uint8), size=L)) + b"\n")
        else:
            w = int(rng.integers(20, 90))
            parts.append(b">g%d d\n" % i + b"\n".join(seq[j:j + w] for j in range(0, L, w)) + b"\n")
    doc = b"".join(parts)
    plain = os.path.join(d, "p.fx"); open(plain, "wb").write(doc)
    want, _ = hostio.read_fastx(plain)
    bg = os.path.join(d, "p.gz")
    synth.write_bgzf(bg, doc, member_sizes=[int(x) for x in rng.integers(1, 70000, size=5)] + [65280], level=int(rng.choice([1, 6])))
    env = dict(os.environ, LD_PRELOAD=so)
    env["BNS_READER_BLOCK"] = str(int(rng.choice([300, 3000, 20000, 70000, 1 << 22])))
    if rng.random() < 0.7:
        env["STRESS_GPU"] = "1"
        env["BNS_BGZF_GPU_BATCH"] = str(int(rng.choice([1, 2, 5, 128])))
        env["BNS_BGZF_GPU_THREADS"] = str(int(rng.integers(1, 4)))
        env["BNS_SHIM_LATENCY_MS"] = str(int(rng.choice([0, 3, 30])))
        if rng.random() < 0.6: env["BNS_GZ_THREADS"] = str(int(rng.choice([0, 1, 3])))
    if rng.random() < 0.3: env["BNS_BGZF_NO_MMAP"] = "1"
    if rng.random() < 0.15 and kind == "fastq": env["BNS_BGZF_FORCE_BAD_CUT"] = str(int(rng.integers(0, 4)))
    args = [str(int(rng.choice([200, 5000, 1 << 20]))), str(int(rng.integers(1, 5))), str(int(rng.choice([1, 5000, 100000, 1 << 24])))]
    p = subprocess.run([sys.executable, "-c", CODE, bg] + args, env=env, capture_output=True, text=True, timeout=120)
    ok = p.returncode == 0 and p.stdout.split()[:2] == [str(len(want)), hashlib.sha256(repr(want).encode()).hexdigest()]
    if not ok:
        print("MISMATCH seed", seed0 * 7919 + it, "args", args, "env", {k: v for k, v in env.items() if k.startswith(("BNS_", "STRESS"))}, "rc", p.returncode, p.stdout[:200], p.stderr[-400:])
        sys.exit(1)
    it += 1
print("bgzf reader stress ok: %d configurations, %.0f s" % (it, time.time() - t0))
