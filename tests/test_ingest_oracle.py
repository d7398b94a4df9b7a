"""tests/kseq_py.py -- the character-level restatement of kseq_read / bseq_read that checks the device text parser -- against the
records the reference's own reader returned (tests/golden/ingest_ref.npz, made in the build container by
tests/golden/make_golden_tree.py with klib/kseq.h + kseq_declare.h compiled where they lie)."""
import gzip
import os

import numpy as np

import kseq_py

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_text(IN, f):
    raw = IN["text_" + f].tobytes()
    return gzip.decompress(raw) if f.endswith("_gz") else raw


def test_kseq_restatement_matches_reference_vectors():
    IN = np.load(os.path.join(GOLD, "ingest_ref.npz"))
    n = int(IN["n_cases"])
    assert n >= 17
    for ci in range(n):
        f1, f2 = str(IN["case%d_file1" % ci]), str(IN["case%d_file2" % ci])
        fields = IN["case%d_fields" % ci].tobytes().split(b"\0")[:-1]
        exp = [tuple(fields[4 * i:4 * i + 4]) for i in range(len(fields) // 4)]
        got = kseq_py.read_all(case_text(IN, f1), case_text(IN, f2) if f2 else None, chunk_size=int(IN["case%d_chunk" % ci]))
        assert got == exp, (ci, f1, f2)


def test_kseq_restatement_live_fuzz(oracle):
    """when the reference build is here: random texts -- regular and wild (CRLF, wrapped quality, stray text, wrong quality
    lengths, a bare header byte at the end) -- through the reference's own kseq_read / bseq_read and through the restatement"""
    import ctypes as C
    import tempfile
    import pytest
    import ingest_fuzz
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_bseq_read_all"):
        pytest.skip("oracle/_ref not built (no reference checkout on this box): the committed goldens above are the pin")
    rng = np.random.default_rng(1)
    with tempfile.TemporaryDirectory() as td:
        for it in range(400):
            doc = ingest_fuzz.make_doc(rng, int(rng.integers(1, 30)), wild=(0.0 if it % 3 else 0.8), final_newline=bool(it % 5), fastq_comments=False)
            if it % 17 == 0:
                doc += b"@"
            chunk = int(rng.choice([1 << 20, 300, 1]))
            p = os.path.join(td, "d.txt")
            open(p, "wb").write(doc)
            cap = len(doc) * 4 + 1024
            blob = C.create_string_buffer(cap); ls = (C.c_int32 * 4096)(); ck = (C.c_int32 * 4096)()
            n = R.ref_bseq_read_all(p.encode(), None, chunk, blob, cap, ls, ck, 4096)
            assert n >= 0
            fields = blob.raw.split(b"\0")[:4 * n]
            exp = [tuple(fields[4 * i:4 * i + 4]) for i in range(n)]
            assert kseq_py.read_all(doc, chunk_size=chunk) == exp, it
