"""The C++ Template-API shims driven the way a C++ caller of the reference drives them: a compiled program
(tests/helpers/cpp_api_harness.cpp -> bonsai_amd/bin/bns_api_check, linked against libbns_host.so) constructs bns::Encoder /
bns::RollingHasher<u64 | u128> and calls the PATH overloads -- Encoder::for_each(func, path, kseq_t*) (encoder.h:511, dispatch
:448-463), for_each_canon / for_each_uncanon (:479-494), for_each_hash(func, path) (:408), RollingHasher(k, canon, enc, wsz, seed1,
seed2) (:672), for_each_hash(func, s, l) (:810) and its path overload (:821).  What the functor saw is compared with the
reference-made k-mer streams (tests/golden/stream_ref.npz: the reference's own LUT / mask / canonical form) and with the oracle,
record by record."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bonsai_amd", "bin", "bns_api_check")
PHIX = os.path.join(ROOT, "tests", "golden", "phix.fa")


def call(mode, path, tmp_path, k=31, canon=1, w=0, score=0, gaps="-", seeds=(), expect_rc=0):
    out = str(tmp_path / "api.bin")
    if os.path.exists(out):
        os.remove(out)
    pr = subprocess.run([BIN, mode, str(path), out, str(k), str(int(canon)), str(w), str(score), gaps] + [str(s) for s in seeds],
                        stderr=subprocess.PIPE, timeout=300)
    assert pr.returncode == expect_rc, pr.stderr.decode()
    if expect_rc:
        return pr.stderr.decode()
    return np.fromfile(out, dtype=np.uint64)


def write_records(path, recs, kind="fasta", wrap=0):
    with (gzip.open(path, "wb") if str(path).endswith(".gz") else open(path, "wb")) as f:
        for i, s in enumerate(recs):
            s = bytes(s)
            if kind == "fasta":
                body = b"\n".join(s[j:j + wrap] for j in range(0, len(s), wrap)) if wrap else s
                f.write(b">r%d some comment\n" % i + body + b"\n")
            else:
                f.write(b"@r%d/1\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")


def records(seed, n=40):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = bytearray(synth.rand_seq(rng, int(rng.integers(5, 900))).tobytes())
        if i % 5 == 0 and len(s) > 100:
            s[50] = ord("N"); s[51] = ord("n"); s[len(s) - 3] = ord("R")
        if i % 7 == 0:
            s = bytearray(bytes(s).lower())
        out.append(bytes(s))
    out.append(b"")                                     # an empty record
    out.append(b"ACGT")                                 # shorter than k
    return out


def test_encoder_path_overload_phix_reference_stream(tmp_path):
    """phiX through Encoder::for_each(func, path): the reference-made streams (5356 canonical / forward 31-mers)"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "stream_ref.npz"))
    assert np.array_equal(call("enc_path", PHIX, tmp_path, k=31, canon=1), z["phix_cn31"])
    assert np.array_equal(call("enc_path_ks", PHIX, tmp_path, k=31, canon=0), z["phix_fw31"])
    assert np.array_equal(call("enc_path_string", PHIX, tmp_path, k=31, canon=1), z["phix_cn31"])
    assert np.array_equal(call("enc_paths", PHIX, tmp_path, k=31, canon=1), np.concatenate([z["phix_cn31"]] * 2))
    # for_each_canon / for_each_uncanon pick their side whatever canonicalize() says
    assert np.array_equal(call("enc_canon_path", PHIX, tmp_path, k=31, canon=0), z["phix_cn31"])
    assert np.array_equal(call("enc_uncanon_path", PHIX, tmp_path, k=31, canon=1), z["phix_fw31"])


@pytest.mark.parametrize("kind,suffix,wrap", [("fasta", ".fa", 60), ("fastq", ".fq", 0), ("fasta", ".fa.gz", 0)])
def test_encoder_path_overload_records(oracle, tmp_path, kind, suffix, wrap):
    """many records (N runs, lowercase, IUPAC, empty, shorter than k; wrapped FASTA, FASTQ, gzip): the functor sees every record's
    stream in file order -- unwindowed, windowed Lex, the entropy PATH rule (F8) where the string overload computes the string
    rule, a spaced seed through for_each_uncanon_spaced where the string overload emits nothing (F7)"""
    recs = records(5)
    p = tmp_path / ("recs" + suffix)
    write_records(p, recs, kind, wrap)
    for k, canon in ((31, 1), (31, 0), (21, 1), (32, 1)):
        exp = np.concatenate([oracle.encode(r, k, canon=bool(canon)) for r in recs])
        assert np.array_equal(call("enc_path", p, tmp_path, k=k, canon=canon), exp), (k, canon)
    # windowed, Lex score: the same function for string and path
    exp = np.concatenate([oracle.encode_windowed(r, 31, 50, oracle.SCORE_LEX, canon=True) for r in recs])
    assert np.array_equal(call("enc_path", p, tmp_path, k=31, canon=1, w=50, score=0), exp)
    exp = np.concatenate([oracle.encode_windowed(r, 31, 50, oracle.SCORE_LEX, canon=False) for r in recs])
    assert np.array_equal(call("enc_path", p, tmp_path, k=31, canon=0, w=50, score=0), exp)
    # Encoder<score::Entropy>: the path overload scores by the path rule whichever of the two entropy constants built it
    exp = np.concatenate([oracle.encode_windowed(r, 31, 50, oracle.SCORE_ENTROPY_PATH, canon=True) for r in recs])
    for score in (1, 2):
        assert np.array_equal(call("enc_path", p, tmp_path, k=31, canon=1, w=50, score=score), exp), score
    # spaced seed (BASELINE configs[2]'s mask): the path overloads reach for_each_uncanon_spaced
    gaps = [1] * 15 + [0] * 15
    exp = np.concatenate([oracle.encode(r, 31, gaps=gaps, canon=False, spaced_intended=True) for r in recs])
    got = call("enc_path", p, tmp_path, k=31, canon=1, gaps="1x15,0x15")
    assert len(exp) > 0 and np.array_equal(got, exp)
    exp = np.concatenate([oracle.encode_windowed(r, 31, 60, oracle.SCORE_LEX, gaps=gaps) for r in recs])
    assert np.array_equal(call("enc_path", p, tmp_path, k=31, canon=1, w=60, gaps="1x15,0x15"), exp)


def test_encoder_string_overload_keeps_the_string_rules(oracle, tmp_path):
    """the same object's string overload: F7 (a spaced seed emits nothing) and the string form of the entropy score"""
    seq = synth.rand_seq(np.random.default_rng(9), 700).tobytes()
    p = tmp_path / "one.txt"
    p.write_bytes(seq)
    assert call("enc_str", p, tmp_path, k=31, canon=1, gaps="1x15,0x15").size == 0
    assert np.array_equal(call("enc_str", p, tmp_path, k=31, canon=1), oracle.encode(seq, 31))
    exp = oracle.encode_windowed_entropy_str(seq, 31, 50, canon=True)
    for score in (1, 2):
        assert np.array_equal(call("enc_str", p, tmp_path, k=31, canon=1, w=50, score=score), exp)


def test_encoder_for_each_hash_path(oracle, tmp_path):
    recs = records(6, n=25)
    p = tmp_path / "h.fa"
    write_records(p, recs, "fasta", wrap=70)
    for k, canon in ((31, 1), (25, 0)):
        exp = np.concatenate([oracle.for_each_hash(r, k, canon=bool(canon)) for r in recs])
        assert exp.size and np.array_equal(call("enc_hash_path", p, tmp_path, k=k, canon=canon), exp), (k, canon)
    # a spaced encoder is refused, as encoder.h:363-364 refuses it
    err = call("enc_hash_path", p, tmp_path, k=31, gaps="1x15,0x15", expect_rc=3)
    assert "bns::Error" in err


@pytest.mark.parametrize("bits", [64, 128])
def test_rolling_hasher_class(oracle, tmp_path, bits):
    """RollingHasher<u64 / u128>(k, canon, DNA, wsz, seed1, seed2): string and path overloads, window and seeds"""
    recs = records(7, n=30)
    p = tmp_path / "r.fq.gz"
    write_records(p, recs, "fastq")
    one = tmp_path / "one.txt"
    one.write_bytes(recs[1])
    ref = oracle.rolling_hash if bits == 64 else oracle.rolling_hash128
    tabs = oracle.rolling_tables if bits == 64 else oracle.rolling_tables128
    mode = "roll%d" % bits
    for k, canon, w, seeds in ((21, 0, -1, ()), (21, 1, -1, ()), (31, 1, 50, ()), (17, 0, 40, (99, 12345)), (31, 1, 10, (7, 3))):
        t = tabs(*seeds) if seeds else None
        ww = w if w > k else 0
        exp = [np.asarray(ref(r, k, canon=bool(canon), tables=t, w=ww)).reshape(-1) for r in recs]
        got = call(mode + "_path", p, tmp_path, k=k, canon=canon, w=w, seeds=seeds)
        assert np.array_equal(got, np.concatenate(exp)), (k, canon, w)
        got1 = call(mode + "_str", one, tmp_path, k=k, canon=canon, w=w, seeds=seeds)
        assert np.array_equal(got1, exp[1]), (k, canon, w)
    if bits == 64:
        # for_each(args...) forwards to for_each_hash; canonicalize(bool) flips the side
        e0 = np.concatenate([oracle.rolling_hash(r, 21, canon=False) for r in recs])
        e1 = np.concatenate([oracle.rolling_hash(r, 21, canon=True) for r in recs])
        assert np.array_equal(call("roll64_path_each", p, tmp_path, k=21, canon=0, w=-1), e0)
        assert np.array_equal(call("roll64_path_flip", p, tmp_path, k=21, canon=0, w=-1), e1)
    # the reference's one RollingHasher test (test/encoding.cpp:152-156): u128, k = 100 windowed on phiX: len - w + 1 values
    if bits == 128:
        n = call("roll128_path", PHIX, tmp_path, k=100, canon=0, w=200).size // 2
        assert n == 5386 - 200 + 1


def test_path_overloads_errors_and_compressors(oracle, tmp_path):
    err = call("enc_path", tmp_path / "missing.fa", tmp_path, expect_rc=3)
    assert "Could not open" in err
    recs = records(8, n=10)
    p = tmp_path / "c.fa"
    write_records(p, recs, "fasta")
    exp = np.concatenate([oracle.encode(r, 31) for r in recs])
    for tool, suf in (("xz", ".xz"), ("bzip2", ".bz2"), ("zstd", ".zst")):
        if not shutil.which(tool):
            continue
        q = str(p) + suf
        with open(q, "wb") as f:
            subprocess.run([tool, "-c", str(p)], stdout=f, check=True)
        assert np.array_equal(call("enc_path", q, tmp_path, k=31), exp), tool
        e64 = np.concatenate([oracle.rolling_hash(r, 21) for r in recs])
        assert np.array_equal(call("roll64_path", q, tmp_path, k=21, canon=0, w=-1), e64), tool
