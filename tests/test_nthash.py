"""Encoder::for_each_hash (encoder.h:355-394, ntHash) -- PARITY UNPINNED: NTC64 lives in the un-vendored bcgsc/ntHash submodule.
The oracle restates the reference's loop as written (labels, goto) over a rolling NTC64; here it is checked against an
independent closed form (maximal A/C/G/T runs, every window hashed directly from the definition), and the HIP kernel and the
C++ host class against the oracle.  The table geometry (complement at letter & 7) is pinned by make_nthash_lut, encoder.h:93-103."""
import os

import numpy as np
import pytest

import synth

M64 = (1 << 64) - 1


def rol(v, s):
    s &= 63
    return ((v << s) | (v >> (64 - s))) & M64 if s else v


def py_for_each_hash(seq: bytes, k: int, canon: bool, T):
    """closed form: every window of every maximal valid run of length >= k; a run whose FIRST window ends at the end of the
    string emits nothing; a NUL byte ends the string."""
    z = seq.find(b"\0")
    if z >= 0:
        seq = seq[:z]
    l = len(seq)
    out = []
    valid = [c in b"ACGTacgt" for c in seq]
    i = 0
    while i < l:
        if not valid[i]:
            i += 1
            continue
        j = i
        while j < l and valid[j]:
            j += 1
        if j - i >= k and i + k != l:
            for w in range(i, j - k + 1):
                f = r = 0
                for t in range(k):
                    c = seq[w + t]
                    f ^= rol(int(T[c]), k - 1 - t)
                    r ^= rol(int(T[c & 7]), t)
                out.append(min(f, r) if canon else f)
        i = j
    return np.array(out, dtype=np.uint64)


def cases(rng):
    seqs = [b"", b"A", b"ACGT", b"ACGTA", b"ACGTAC", b"NNACGTA", b"NNACGTAC", b"ACGTANACGTAC", b"ACGTAN", b"ACGTACN", b"ACGTAC\0ACGTACGT",
            b"acgtACGTnACGTacgtAC", b"N" * 40, b"ACGT" * 40 + b"N" + b"ACGT" * 8, b"ACGT" * 8 + bytes([200]) + b"TTGACCATTGACCA" * 5,
            b"A" * 31 + b"N" + b"C" * 31 + b"N" + b"G" * 32]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
             for L, r in zip(rng.integers(1, 400, size=60), rng.choice([0, 0.01, 0.05], size=60))]
    return seqs


def test_table_geometry(oracle):
    """make_nthash_lut (encoder.h:93-103): ret[4] = a, ret[7] = c, ret[3] = g, ret[1] = t -- the complement of letter x sits at x & 7."""
    t = oracle.nthash_tables((11, 22, 33, 44))
    for ch, own, comp in ((b"A", 11, 44), (b"C", 22, 33), (b"G", 33, 22), (b"T", 44, 11), (b"a", 11, 44), (b"t", 44, 11)):
        assert int(t[ch[0]]) == own and int(t[ch[0] & 7]) == comp
    assert int(t[ord("N")]) == 0 and np.count_nonzero(t) == 12            # 8 letters + indices 1, 3, 4, 7


def test_for_each_hash_restatement(oracle):
    rng = np.random.default_rng(5)
    custom = oracle.nthash_tables([int(x) for x in rng.integers(1, 1 << 63, size=4)])
    for s in cases(rng):
        for k in (1, 2, 5, 21, 31, 32, 33, 64, 65, 100):
            for canon in (False, True):
                for T in (None, custom):
                    got = oracle.for_each_hash(s, k, canon, T)
                    exp = py_for_each_hash(s, k, canon, oracle.nthash_tables() if T is None else T)
                    assert np.array_equal(got, exp), (s[:40], k, canon)
    # a clean sequence: l - k + 1 values (l > k), none when l == k
    clean = synth.rand_seq(rng, 300).tobytes()
    assert oracle.for_each_hash(clean, 21).size == 280 and oracle.for_each_hash(clean[:21], 21).size == 0
    # canonical value is strand-symmetric
    rc = synth.revcomp(np.frombuffer(clean, dtype=np.uint8)).tobytes()
    assert np.array_equal(oracle.for_each_hash(clean, 31), oracle.for_each_hash(rc, 31)[::-1])


@pytest.mark.gpu
def test_for_each_hash_gpu(gpu_ctx, oracle):
    rng = np.random.default_rng(6)
    seqs = cases(rng) + [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.001, 0.02).tobytes() for L in (5000, 10000, 10000, 66000)]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    custom = oracle.nthash_tables([int(x) for x in rng.integers(1, 1 << 63, size=4)])
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    for k in (1, 5, 21, 31, 32, 33, 64, 65, 130):
        for canon in (0, 1):
            for T in (None, custom):
                got = gpu_ctx.for_each_hash(bases, offsets, k=k, canon=canon, table=T)
                for s, g in zip(seqs, got):
                    assert np.array_equal(g, oracle.for_each_hash(s, k, bool(canon), T)), (k, canon, len(s), s[:30])
    # k = 0 / canon = -1 take the encoder's values; a spaced or windowed encoder is refused (encoder.h:363-364)
    got = gpu_ctx.for_each_hash(bases, offsets)
    assert all(np.array_equal(g, oracle.for_each_hash(s, 31, True)) for s, g in zip(seqs, got))
    import bonsai_amd
    gpu_ctx.set_encoder(31, [1] * 15 + [0] * 15, canonicalize=True)
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        gpu_ctx.for_each_hash(bases, offsets)
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    gpu_ctx.set_window(50)
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        gpu_ctx.for_each_hash(bases, offsets)
    gpu_ctx.set_window(0)


@pytest.mark.gpu
def test_host_encoder_for_each_hash(oracle):
    """bns::Encoder::for_each_hash (C++ host class over the C ABI)."""
    from bonsai_amd import hostio
    rng = np.random.default_rng(8)
    s = synth.mutate(rng, synth.rand_seq(rng, 3000), 0.0, 0.004, 0.05).tobytes()
    assert np.array_equal(hostio.encoder_hash_from_str(s, 31, canon=True), oracle.for_each_hash(s, 31, True))
    assert np.array_equal(hostio.encoder_hash_from_str(s, 31, canon=False, hash_k=70), oracle.for_each_hash(s, 70, False))
    with pytest.raises(hostio.HostIOError):
        hostio.encoder_hash_from_str(s, 31, gaps=[1] * 30, canon=True)


# ---------------------------------------------------------------- RollingHasher<__uint128_t> (SURVEY 8a row 11, 128-bit word)
M128 = (1 << 128) - 1


def rol128(v, s):
    s &= 127
    return ((v << s) | (v >> (128 - s))) & M128 if s else v


def py_rolling128(seq: bytes, k: int, canon: bool, tf, tr):
    """the two loops of RollingHasher::for_each_canon / for_each_uncanon (encoder.h:692-796) over Python ints, 128-bit rotations"""
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    T = lambda t, i: int(t[2 * i]) | (int(t[2 * i + 1]) << 64)      # noqa: E731
    rcc = lambda c: 3 - code[c] if c in code else 255               # noqa: E731
    l, out, i, myr = len(seq), [], 0, k % 128
    if l < k:
        return out
    while True:
        h = g = nf = 0
        while nf < k and i < l:
            c = seq[i]
            if c not in code:
                if canon and i + 2 * k >= l:
                    return out
                i += k; nf = 0; h = g = 0
            else:
                h = rol128(h, 1) ^ T(tf, code[c])
                if canon:
                    g = rol128(g, 1) ^ T(tr, rcc(seq[i - nf + k - 1]))
                nf += 1
            i += 1
        if nf < k:
            return out
        out.append(min(h, g) if canon else h)
        restart = False
        while i < l:
            c = seq[i]
            if c not in code:
                restart = True
                break
            h = rol128(h, 1) ^ rol128(T(tf, code[seq[i - k]]), myr) ^ T(tf, code[c])
            if canon:
                g ^= rol128(T(tr, rcc(c)), myr) ^ T(tr, rcc(seq[i - k]))
                g = rol128(g, 127)
            out.append(min(h, g) if canon else h)
            i += 1
        if not restart:
            return out
        if canon and i + 2 * k >= l:
            return out
        i += k + 1


def as_ints(a):
    return [int(lo) | (int(hi) << 64) for lo, hi in a]


def test_rolling128_restatement(oracle):
    rng = np.random.default_rng(15)
    tf, tr = oracle.rolling_tables128()
    assert not tf[1::2].any() and not tr[1::2].any() and tf[0::2].all()     # default entries keep the low word only (characterhash.h:82-97)
    full = (rng.integers(0, 1 << 63, size=512, dtype=np.uint64) * np.uint64(2) + np.uint64(1), rng.integers(0, 1 << 63, size=512, dtype=np.uint64))
    seqs = [b"", b"ACGT", b"A" * 150, b"ACGTN" * 30, b"N" * 50 + b"ACGT" * 60, b"acgtACGT" * 20 + b"N" + b"TTGACCA" * 40]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
             for L, r in zip(rng.integers(1, 600, size=25), rng.choice([0, 0.005, 0.03], size=25))]
    for s in seqs:
        for k in (1, 21, 63, 64, 65, 100, 127, 128, 129, 200):
            for canon in (False, True):
                for tabs in ((tf, tr), full):
                    got = as_ints(oracle.rolling_hash128(s, k, canon, tabs))
                    assert got == py_rolling128(s, k, canon, tabs[0], tabs[1]), (len(s), k, canon)
    clean = synth.rand_seq(rng, 5386).tobytes()
    assert oracle.rolling_hash128(clean, 100).shape[0] == 5386 - 100 + 1


def _frev64(x):
    M = (1 << 64) - 1
    x ^= 0x533f8c2151b20f97
    x = (x * 0x9a98567ed20c127d) & M
    x = ((x << 31) | (x >> 33)) & M
    return x ^ 0x691a9d706391077a


def py_window128(stream_pairs, ws):
    """QueueMap over a finished stream (qmap.h:79-87 as RollingHasher drives it): minimum by (score, value) of every ws
    consecutive entries, one flushed minimum for a stream shorter than the window; score as the oracle restates it"""
    sc = lambda v: _frev64((v & ((1 << 64) - 1)) ^ _frev64(v >> 64))
    if not stream_pairs:
        return []
    if len(stream_pairs) < ws:
        return [min(stream_pairs, key=lambda v: (sc(v), v))]
    return [m for m in (min(stream_pairs[i:i + ws], key=lambda v: (sc(v), v)) for i in range(len(stream_pairs) - ws + 1)) if m != (1 << 128) - 1]


def test_rolling128_windowed_restatement(oracle):
    """the windowed 128-bit hasher = the unwindowed stream (both strands as separate entries on the canonical path) through the
    queue; and the reference's own test of this instantiation (test/encoding.cpp:152-156): k = 100, w = 200 over phiX gives
    len - w + 1 values -- pinned on the reference's phiX file (tests/golden/phix.fa, DNA alphabet)"""
    rng = np.random.default_rng(19)
    tf, tr = oracle.rolling_tables128()
    seqs = [b"", b"ACGT" * 10, b"ACGTN" * 30, b"N" * 50 + b"ACGT" * 60] + [
        synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
        for L, r in zip(rng.integers(1, 700, size=20), rng.choice([0, 0.004, 0.03], size=20))]
    for s in seqs:
        for k, w in ((21, 30), (21, 22), (64, 100), (100, 200), (5, 70)):
            fwd = py_rolling128(s, k, False, tf, tr)
            assert as_ints(oracle.rolling_hash128(s, k, False, None, w=w)) == py_window128(fwd, w - k + 1), (len(s), k, w)
            # canonical: py_rolling128 returns min(h, g); the queue wants both -- rebuild them from two uncanonical views
            both = py_rolling128_both(s, k, tf, tr)
            assert as_ints(oracle.rolling_hash128(s, k, True, None, w=w)) == py_window128(both, w - k + 1), (len(s), k, w, "canon")
    phix = b"".join(l.strip() for l in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "phix.fa"), "rb") if not l.startswith(b">"))
    assert len(phix) == 5386
    assert oracle.rolling_hash128(phix, 100, False, None, w=200).shape[0] == 5386 - 200 + 1          # REQUIRE(total_hash == 5386 - 200 + 1)
    assert oracle.rolling_hash128(phix, 100, False, None, w=0).shape[0] == 5386 - 100 + 1


def py_rolling128_both(seq, k, tf, tr):
    """the canonical path's two hash values per position, forward then reverse, as the windowed hasher queues them"""
    out = []
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    M = (1 << 128) - 1
    rol = lambda x, r: ((x << (r % 128)) | (x >> (128 - r % 128))) & M if r % 128 else x
    T = lambda t, i: int(t[2 * i]) | (int(t[2 * i + 1]) << 64)
    rcc = lambda c: 3 - code[c]
    l, myr, i = len(seq), k % 128, 0
    if l < k or k == 0:
        return out
    while True:
        h = g = 0
        nf = 0
        while nf < k and i < l:
            c = seq[i]
            if c not in code:
                if i + 2 * k >= l:
                    return out
                i += k; nf = 0; h = g = 0
            else:
                h = rol(h, 1) ^ T(tf, code[c])
                g = rol(g, 1) ^ T(tr, rcc(seq[i - nf + k - 1])) if seq[i - nf + k - 1] in code else rol(g, 1)
                nf += 1
            i += 1
        if nf < k:
            return out
        out += [h, g]
        restart = False
        while i < l:
            c = seq[i]
            if c not in code:
                restart = True
                break
            h = rol(h, 1) ^ rol(T(tf, code[seq[i - k]]), myr) ^ T(tf, code[c])
            g ^= rol(T(tr, rcc(c)), myr) ^ T(tr, rcc(seq[i - k]))
            g = rol(g, 127)
            out += [h, g]
            i += 1
        if not restart:
            return out
        if i + 2 * k >= l:
            return out
        i += k + 1


@pytest.mark.gpu
def test_rolling128_windowed_gpu(gpu_ctx, oracle):
    """bns_rolling_hash128_windowed_batch == the oracle, forward-only and canonical, windows from 2 to 150 values, sequences
    shorter than the window, N restarts; and the reference's count on phiX"""
    rng = np.random.default_rng(23)
    seqs = [b"", b"ACGT", b"A" * 300, b"ACGTN" * 60, b"N" * 50 + b"ACGT" * 80]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
             for L, r in zip(rng.integers(1, 3000, size=20), rng.choice([0, 0.002, 0.02], size=20))]
    phix = b"".join(l.strip() for l in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "phix.fa"), "rb") if not l.startswith(b">"))
    seqs.append(phix)
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    for k, w in ((21, 22), (21, 50), (64, 100), (100, 200), (5, 154)):
        for canon in (False, True):
            got = gpu_ctx.rolling_hash128(bases, offsets, k, canon, None, w=w)
            for s, g in zip(seqs, got):
                assert np.array_equal(g, oracle.rolling_hash128(s, k, canon, None, w=w)), (k, w, canon, len(s))
    assert gpu_ctx.rolling_hash128(bases, offsets, 100, False, None, w=200)[-1].shape[0] == 5386 - 200 + 1


@pytest.mark.gpu
def test_rolling128_gpu(gpu_ctx, oracle):
    rng = np.random.default_rng(16)
    seqs = [b"", b"ACGT", b"A" * 300, b"ACGTN" * 60, b"N" * 50 + b"ACGT" * 80, b"ACGT" * 40 + b"N"]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
             for L, r in zip(rng.integers(1, 5000, size=30), rng.choice([0, 0.002, 0.02], size=30))]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, 10000), 0.0, 0.0005, 0.02).tobytes()]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    full = (rng.integers(0, 1 << 63, size=512, dtype=np.uint64) * np.uint64(2) + np.uint64(1), rng.integers(0, 1 << 63, size=512, dtype=np.uint64))
    for k in (1, 21, 64, 65, 100, 127, 128, 129, 200):
        for canon in (False, True):
            for tabs in (None, full):
                got = gpu_ctx.rolling_hash128(bases, offsets, k, canon, tabs)
                for s, g in zip(seqs, got):
                    assert np.array_equal(g, oracle.rolling_hash128(s, k, canon, tabs)), (k, canon, len(s), tabs is None)
