"""Encoder::for_each_hash (encoder.h:355-394, ntHash) -- PARITY UNPINNED: NTC64 lives in the un-vendored bcgsc/ntHash submodule.
The oracle restates the reference's loop as written (labels, goto) over a rolling NTC64; here it is checked against an
independent closed form (maximal A/C/G/T runs, every window hashed directly from the definition), and the HIP kernel and the
C++ host class against the oracle.  The table geometry (complement at letter & 7) is pinned by make_nthash_lut, encoder.h:93-103."""
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
