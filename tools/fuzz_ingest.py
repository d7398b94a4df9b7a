#!/usr/bin/env python
"""Randomised soak of the device text parser (bns_classify_text, parse only) against the character-level kseq restatement tests/kseq_py.py:
random FASTA / FASTQ text -- CRLF records, quality over several lines whose lines start with '@' / '+' / '>', blank lines, text between
records, truncated and overlong quality -- one call or pieces of 8 KiB, one classify batch or many; what every call took must be what kseq
takes, and kseq restarted where the call stopped must read the rest; texts kseq reads cleanly must be taken whole.
usage: tools/fuzz_ingest.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bonsai_amd                # noqa: E402
from bonsai_amd import _lib      # noqa: E402
import ingest_fuzz               # noqa: E402
import kseq_py                   # noqa: E402
from test_gpu_ingest import check_call   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = bonsai_amd.Context(0)
ctx.set_encoder(31, None, canonicalize=True)
t0 = time.time()
it = n_ok = n_irr = n_slow = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 1000003 + it)
    wild = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    doc = ingest_fuzz.make_doc(rng, int(rng.integers(1, 300)), wild=wild, kinds=(("fastq",), ("fasta",), ("fastq", "fasta"))[it % 3],
                               final_newline=bool(rng.integers(0, 2)), crlf=float(rng.choice([0.0, 0.0, 0.3, 1.0])), wrapq=float(rng.choice([0.0, 0.5, 1.0])),
                               max_len=int(rng.choice([20, 120, 400])))
    if rng.random() < 0.1: doc += b"\n" + (b"@" if rng.random() < 0.5 else b">")
    if rng.random() < 0.1: doc = b"\r\n\n" + doc
    dbg = int(rng.choice([0, 0x4000, 0x4040]))
    ctx.debug_set(dbg)
    final = bool(rng.random() < 0.7)
    try:
        res, recs = check_call(ctx, doc, final=final, trim=bool(rng.integers(0, 2)), limit=(int(rng.integers(0, len(doc) + 1)) if rng.random() < 0.2 else None), want_words=dbg == 0)
    except AssertionError:
        print("INGEST MISMATCH seed", seed0 * 1000003 + it, "final", final, "bytes", len(doc)); open("/tmp/fuzz_ingest_fail.txt", "wb").write(doc)
        raise
    if final and kseq_py.reads_cleanly(doc) and res["status"] != _lib.TEXT_OK:
        print("NOT TAKEN seed", seed0 * 1000003 + it, res["status"], res["why"]); sys.exit(1)
    n_ok += res["status"] == _lib.TEXT_OK; n_irr += res["status"] == _lib.TEXT_IRREGULAR
    it += 1
print("ingest fuzz ok: %d texts in %.0f s (%d taken whole or up to a limit, %d handed back)" % (it, time.time() - t0, n_ok, n_irr))
