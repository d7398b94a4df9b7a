"""Windowed minimizers and device db construction (8f-1) against the oracle."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def present_pairs(flags, keys, vals, nb):
    i = np.arange(nb)
    st = (flags[i >> 4] >> ((i & 15) << 1)) & 3
    m = st == 0
    order = np.argsort(keys[m], kind="stable")
    return keys[m][order], vals[m][order]


@pytest.mark.parametrize("score", [0, 1])
@pytest.mark.parametrize("w", [31, 32, 50, 94, 95, 160, 700, 31 + 1023])
def test_encode_windowed(gpu_ctx, oracle, w, score):
    k = 31
    rng = np.random.default_rng(w * 7 + score)
    seqs = [b"", b"ACGT" * 10, b"ACGT" * 12 + b"A", synth.rand_seq(rng, w).tobytes(), synth.rand_seq(rng, w - 1).tobytes(),
            b"ACGTNACGT" * 30]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.01, 0.1).tobytes() for L in rng.integers(1, 6000, size=25)]
    import os
    seqs.append(oracle.read_fasta(os.path.join(os.path.dirname(__file__), "golden", "phix.fa"))[0][1])
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    gpu_ctx.set_encoder(k, None, canonicalize=True)
    gpu_ctx.set_window(w, score)
    got = gpu_ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        exp = oracle.encode_windowed(s, k, w, score)          # w <= k: the unwindowed canonical stream
        assert np.array_equal(g, exp), (len(s), g.size, exp.size)
    if w == 50:
        assert got[-1].size == 5386 - 50 + 1          # reference test "qmap" / SURVEY F8: len - w + 1 windows on phiX
    gpu_ctx.set_encoder(k, None, canonicalize=True)   # resets the window
    assert np.array_equal(gpu_ctx.encode(bases, offsets)[-1], oracle.encode(seqs[-1], k))


def test_encode_windowed_k32_entropy_overflow(gpu_ctx, oracle):
    """k = 32: a k-mer above 2^63 * 0.9999 scores below -2^63, where the reference's double -> integer conversion (x86
    cvttsd2si) yields 0x8000000000000000; the GPU must return the same (found by tools/fuzz_gpu_build.py)."""
    k, w = 32, 80
    rng = np.random.default_rng(4)
    seqs = [b"T" * 200, b"G" * 100 + b"T" * 100, synth.rand_seq(rng, 3000).tobytes(), b"TTTTGTTT" * 40]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    gpu_ctx.set_encoder(k, None, canonicalize=True)
    gpu_ctx.set_window(w, 1)
    for s, g in zip(seqs, gpu_ctx.encode(bases, offsets)):
        assert np.array_equal(g, oracle.encode_windowed(s, k, w, 1))
    gpu_ctx.set_encoder(k, None, canonicalize=True)


@pytest.mark.parametrize("gaps", [[1] * 15 + [0] * 15, [0, 2, 1] * 10, [0] * 29 + [40]])
@pytest.mark.parametrize("score", [0, 1])
def test_encode_windowed_spaced(gpu_ctx, oracle, gaps, score):
    """The path overloads' spaced + windowed stream (for_each_uncanon_spaced -> next_minimizer, encoder.h:233-239,615-620):
    what `bonsai build -S ... -w 50` feeds the db with (BASELINE configs[2])."""
    k = 31
    comb = k + sum(gaps)
    rng = np.random.default_rng(comb + score)
    seqs = [b"", b"ACGT" * 40, b"T" * 150, b"ACGTNACGT" * 30]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.01, 0.1).tobytes() for L in rng.integers(1, 5000, size=20)]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    for w in (comb, comb + 4, comb + 37, comb + 63, comb + 64, comb + 300):
        gpu_ctx.set_encoder(k, gaps, canonicalize=True, spaced_intended=True)
        gpu_ctx.set_window(w, score)
        for s, g in zip(seqs, gpu_ctx.encode(bases, offsets)):
            assert np.array_equal(g, oracle.encode_windowed(s, k, w, score, gaps=gaps)), (w, len(s))
    gpu_ctx.set_encoder(k, None, canonicalize=True)


@pytest.mark.parametrize("score", [0, 1])
@pytest.mark.parametrize("k,w", [(31, 50), (31, 32), (32, 40), (32, 95), (30, 45), (15, 31), (4, 10), (31, 94), (31, 95), (31, 231), (32, 32 + 1023)])
def test_encode_windowed_uncanon(gpu_ctx, oracle, k, w, score):
    """Encoder::for_each_uncanon_unspaced_windowed (`-C -w`): windows over the emitted forward k-mers (across N gaps), the
    partial-window flush, and the k >= 31 restart at every 32nd T of a T run -- also across the 2048-base chunks a
    wavefront works in (T runs of several thousand bases, at every phase)."""
    rng = np.random.default_rng(k * 1000 + w * 7 + score)
    seqs = [b"", b"ACGT" * 10, b"ACGT" * 12 + b"A", b"T" * 200, b"T" * 32, b"T" * 31, b"A" + b"T" * 31 + b"ACGT" * 20,
            b"ACGTNACGT" * 30, b"N" * 100, b"ACGTN" * 30 + b"ACGT" * 40, synth.rand_seq(rng, w).tobytes(),
            synth.rand_seq(rng, w - 1).tobytes(), b"t" * 5000, b"ACG" + b"T" * 4321 + b"ACGT" * 30,
            b"ACGT" * 700 + b"N" + b"ACGT" * 9]
    for L in rng.integers(1, 9000, size=30):
        s = bytearray(synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.01, 0.1).tobytes())
        for _ in range(int(rng.integers(0, 4))):           # T runs: short, around 32, and longer than a chunk
            a = int(rng.integers(0, len(s))); n = int(rng.choice([5, 31, 32, 33, 63, 64, 65, 200, 2100, 4200]))
            s[a:a + n] = b"T" * len(s[a:a + n])
        seqs.append(bytes(s))
    import os
    seqs.append(oracle.read_fasta(os.path.join(os.path.dirname(__file__), "golden", "phix.fa"))[0][1])
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    gpu_ctx.set_encoder(k, None, canonicalize=False)
    gpu_ctx.set_window(w, score)
    got = gpu_ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        exp = oracle.encode_windowed(s, k, w, score, canon=False)
        assert np.array_equal(g, exp), (len(s), g.size, exp.size, s[:80])
    gpu_ctx.set_encoder(31, None, canonicalize=True)


@pytest.mark.parametrize("canon", [True, False])
@pytest.mark.parametrize("k,w", [(31, 50), (31, 32), (32, 45), (15, 40), (4, 9), (21, 84), (21, 85), (31, 31 + 400)])
def test_encode_windowed_entropy_string(gpu_ctx, oracle, k, w, canon):
    """The string overload's real-entropy score (row 9): GPU == oracle restatement, bit for bit (the entropy terms come from
    one host libm table; homopolymers and the all-T 32-mer go through the x86 conversion's overflow branches)."""
    rng = np.random.default_rng(k * 100 + w + int(canon))
    seqs = [b"", b"ACGT" * 10, b"A" * 80, b"T" * 80, b"C" * 40 + b"ACGTTGCA" * 10, b"G" * 33 + b"N" + b"T" * 70, b"ACGTN" * 30,
            b"ACGT" * 7 + b"N" + b"GATTACA" * 12, b"acgtacgtac" * 9, synth.rand_seq(rng, w).tobytes(), synth.rand_seq(rng, w - 1).tobytes()]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.01, 0.1).tobytes() for L in rng.integers(1, 7000, size=25)]
    low = bytearray(synth.rand_seq(rng, 3000).tobytes()); low[500:900] = b"AT" * 200; low[1500:1700] = b"C" * 200
    seqs.append(bytes(low))
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    gpu_ctx.set_encoder(k, None, canonicalize=canon)
    gpu_ctx.set_window(w, oracle.SCORE_ENTROPY_STRING)
    got = gpu_ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        exp = oracle.encode_windowed_entropy_str(s, k, w, canon)
        assert np.array_equal(g, exp), (len(s), g.size, exp.size, s[:60])
    gpu_ctx.set_encoder(31, None, canonicalize=True)


def test_window_argument_checks(gpu_ctx):
    import bonsai_amd
    gpu_ctx.set_encoder(31, [1] * 15 + [0] * 15, canonicalize=True)
    gpu_ctx.set_window(60, 0)                         # spaced + windowed: for_each_uncanon_spaced through a window
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        gpu_ctx.set_window(60, 2)                     # the real-entropy score is the contiguous string overload
    gpu_ctx.set_encoder(31, None, canonicalize=False)
    gpu_ctx.set_window(50, 1)                         # -C windowed: for_each_uncanon_unspaced_windowed
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        gpu_ctx.set_window(31 + 1024, 0)              # > 1024 k-mers per window
    gpu_ctx.set_window(31 + 1023, 0)
    gpu_ctx.set_encoder(31, None, canonicalize=False)
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        gpu_ctx.set_window(31 + 1024, 0)
    gpu_ctx.set_window(31 + 1023, 0)
    gpu_ctx.set_encoder(31, None, canonicalize=True)


def device_build(ctx, genomes, taxids, nb):
    seqs = [np.ascontiguousarray(g) for g in genomes]
    bases, offsets = synth.concat(seqs)
    pad = (-bases.size) % 8 + 8
    d_bases = ctx.dev_alloc(bases.size + pad); ctx.dev_upload(d_bases, bases)
    d_off = ctx.dev_alloc(offsets.nbytes); ctx.dev_upload(d_off, offsets)
    tx = np.ascontiguousarray(taxids, dtype=np.uint32)
    d_tx = ctx.dev_alloc(tx.nbytes); ctx.dev_upload(d_tx, tx)
    fs = max(1, nb >> 4)
    d_f = ctx.dev_alloc(fs * 4); d_k = ctx.dev_alloc(nb * 8); d_v = ctx.dev_alloc(nb * 4)
    hdr = ctx.build_table_device(d_bases, d_off, len(seqs), int(offsets[-1]), d_tx, nb, d_f, d_k, d_v)
    flags = np.zeros(fs, dtype=np.uint32); keys = np.zeros(nb, dtype=np.uint64); vals = np.zeros(nb, dtype=np.uint32)
    ctx.dev_download(d_f, flags); ctx.dev_download(d_k, keys); ctx.dev_download(d_v, vals)
    for p in (d_bases, d_off, d_tx, d_f, d_k, d_v):
        ctx.dev_free(p)
    return hdr, flags, keys, vals


@pytest.mark.parametrize("w,score,canon", [(31, 0, True), (50, 1, True), (40, 0, True), (50, 1, False), (45, 0, False), (31, 0, False)])
def test_build_table_device(gpu_ctx, oracle, small_world, w, score, canon):
    """update_lca_map on device: same key -> lca map as the oracle's sequential build, and valid khash arrays."""
    wld = small_world
    k = 31
    exp_t = oracle.Table()
    for leaf, g in wld.genomes.items():
        if w > k:
            oracle.lca_map_add_windowed(exp_t, wld.tax, k, w, score, g.tobytes(), leaf, canon=canon)
        else:
            oracle.lca_map_add(exp_t, wld.tax, k, g.tobytes(), leaf, canon=canon)
    ef, ek, ev = exp_t.arrays()
    exp_keys, exp_vals = present_pairs(ef, ek, ev, exp_t.n_buckets)
    gpu_ctx.set_encoder(k, None, canonicalize=canon)
    gpu_ctx.set_window(w, score)
    gpu_ctx.load_taxonomy(wld.parent)
    nb = 1 << 17
    hdr, flags, keys, vals = device_build(gpu_ctx, list(wld.genomes.values()), list(wld.genomes.keys()), nb)
    got_keys, got_vals = present_pairs(flags, keys, vals, nb)
    assert np.array_equal(got_keys, exp_keys) and np.array_equal(got_vals, exp_vals)
    assert int(hdr[0]) == nb and int(hdr[1]) == exp_keys.size and int(hdr[2]) == exp_keys.size
    # the arrays are a valid khash for the reference's kh_get (probe-path invariant), and empty slots are zeroed
    t = oracle.Table.wrap(int(hdr[0]), int(hdr[2]), int(hdr[1]), int(hdr[3]), flags, keys, vals)
    qv, qf = t.get_batch(exp_keys)
    assert qf.all() and np.array_equal(qv, exp_vals)
    i = np.arange(nb)
    st = (flags[i >> 4] >> ((i & 15) << 1)) & 3
    assert set(np.unique(st).tolist()) <= {0, 2} and not keys[st == 2].any() and not vals[st == 2].any()
    # and it classifies like the oracle's own table
    gpu_ctx.set_encoder(k, None, canonicalize=canon)
    gpu_ctx.load_table(nb, flags, keys, vals)
    reads = synth.simulate_reads(np.random.default_rng(3), wld.genomes, 500)
    b, o = synth.concat(reads)
    exp = oracle.classify_batch(exp_t, wld.tax, k, b, o, canon=canon)
    got = gpu_ctx.classify(b, o)
    assert np.array_equal(got["taxon"], exp["taxon"]) and np.array_equal(got["missing"], exp["missing"])
    gpu_ctx.set_encoder(k, None, canonicalize=True)


def test_build_rejects_overfull(gpu_ctx, oracle, small_world):
    import bonsai_amd
    wld = small_world
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    gpu_ctx.load_taxonomy(wld.parent)
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        device_build(gpu_ctx, list(wld.genomes.values()), list(wld.genomes.keys()), 1 << 15)   # 26k keys > 0.77 * 32k


@pytest.mark.parametrize("nb", [4, 64, 4096, 1 << 14])
def test_build_undersized_table_returns(gpu_ctx, small_world, nb):
    """Fewer buckets than distinct keys (26k): pass 1's probe is bounded, so the call comes back with BNS_ERR_TABLE instead of
    spinning on a full table -- what bns::lca_map's `nb <<= 1` retry relies on.  (Runs under the test timeout: a hang fails.)"""
    import time
    import bonsai_amd
    wld = small_world
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    gpu_ctx.load_taxonomy(wld.parent)
    t0 = time.time()
    with pytest.raises(bonsai_amd.BonsaiAmdError) as e:
        device_build(gpu_ctx, list(wld.genomes.values()), list(wld.genomes.keys()), nb)
    assert "too small" in str(e.value) and time.time() - t0 < 30
    hdr, flags, keys, vals = device_build(gpu_ctx, list(wld.genomes.values()), list(wld.genomes.keys()), 1 << 17)   # and the context still works
    assert int(hdr[2]) > 20000


def test_host_encoder_class(oracle):
    """bns::Encoder (C++ host mirror of Encoder<ScoreType>(Spacer(k, w, gaps), canonicalize)): every stream the string overload
    of for_each can take."""
    import os
    from bonsai_amd import hostio
    name, seq = oracle.read_fasta(os.path.join(os.path.dirname(__file__), "golden", "phix.fa"))[0]
    s = seq[:1500] + b"N" + seq[1500:2400] + b"T" * 70 + seq[2400:3000]
    assert np.array_equal(hostio.encoder_from_str(s, 31), oracle.encode(s, 31, canon=True))
    assert np.array_equal(hostio.encoder_from_str(s, 21, canon=False), oracle.encode(s, 21, canon=False))
    assert np.array_equal(hostio.encoder_from_str(s, 31, w=50), oracle.encode_windowed(s, 31, 50, 0))
    assert np.array_equal(hostio.encoder_from_str(s, 31, canon=False, w=50), oracle.encode_windowed(s, 31, 50, 0, canon=False))
    for canon in (True, False):
        assert np.array_equal(hostio.encoder_from_str(s, 31, canon=canon, w=45, score=oracle.SCORE_ENTROPY_STRING),
                              oracle.encode_windowed_entropy_str(s, 31, 45, canon))
    assert hostio.encoder_from_str(s, 31, gaps=[1] * 15 + [0] * 15).size == 0                  # string overload + spaced seed: SURVEY F7


def test_python_bns_surface(oracle, tmp_path):
    """python/bns.cpp names over the GPU encoder."""
    import os
    from bonsai_amd import bns
    name, seq = oracle.read_fasta(os.path.join(os.path.dirname(__file__), "golden", "phix.fa"))[0]
    a = bns.from_str(seq.decode(), k=31)
    assert np.array_equal(a, oracle.encode(seq, 31, canon=True)) and a.dtype == np.uint64
    assert np.array_equal(bns.from_str(seq.decode(), k=21, canon=False), oracle.encode(seq, 21, canon=False))
    assert bns.from_str(seq.decode(), k=31, spacing="1x15,0x15").size == 0                     # string overload, SURVEY F7
    assert np.array_equal(bns.from_str(seq.decode(), k=31, w=50), oracle.encode_windowed(seq, 31, 50, 0))
    assert np.array_equal(bns.from_str(seq.decode(), k=31, w=50, canon=False), oracle.encode_windowed(seq, 31, 50, 0, canon=False))
    fa = tmp_path / "two.fa"
    fa.write_bytes(b">r1 x\n" + seq[:300] + b"\n>r2\n" + seq[1000:1400] + b"\n")
    l = bns.seqlist(str(fa), k=31)
    assert len(l) == 2 and np.array_equal(l[1], oracle.encode(seq[1000:1400], 31))
    u = bns.from_fasta(str(fa), k=31, unique=True)
    assert np.array_equal(u, np.unique(np.concatenate(l)))
    sp = bns.from_fasta(str(fa), k=31, spacing="1x15,0x15")                                     # path overload: spaced works
    assert np.array_equal(sp, np.concatenate([oracle.encode(s, 31, gaps=[1] * 15 + [0] * 15, spaced_intended=True)
                                              for s in (seq[:300], seq[1000:1400])]))
    d = bns.seqdict(str(fa), k=31)
    assert sorted(d) == ["r1", "r2"] and np.array_equal(d["r1"], l[0])
    # rolling variants (RollingHasher; self-consistent, reference-unverified: SURVEY F10)
    r = bns.seqlist(str(fa), k=21, rolling=True)
    assert len(r) == 2 and np.array_equal(r[0], oracle.rolling_hash(seq[:300], 21, False))
    fr = bns.from_fasta_r(str(fa), k=21, canon=True)
    assert np.array_equal(fr, np.concatenate([oracle.rolling_hash(s, 21, True) for s in (seq[:300], seq[1000:1400])]))
    assert np.array_equal(bns.from_fasta_r(str(fa), k=21, canon=True, unique=True), np.unique(fr))
    dr = bns.seqdict_r(str(fa), k=21)
    assert sorted(dr) == ["r1", "r2"] and np.array_equal(dr["r2"], r[1])


def test_rolling_hash(gpu_ctx, oracle):
    """RollingHasher on the GPU against the restatement, custom and default tables, k below / at / above
    the word size, invalid characters in every phase of the reference's loops."""
    rng = np.random.default_rng(12)
    seqs = [b"", b"ACGT", b"A" * 100, b"ACGTN" * 30, b"N" * 50 + b"ACGT" * 40, b"ACGT" * 20 + b"N", b"acgtACGT" * 10 + b"N" + b"TTGACCA" * 40]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
             for L, r in zip(rng.integers(1, 5000, size=40), rng.choice([0, 0.002, 0.02], size=40))]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, 10000), 0.0, r, 0.02).tobytes() for r in (0.0, 0.0005, 0.003)]   # 10 kb reads
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    custom = (rng.integers(0, 1 << 63, size=256, dtype=np.uint64) * np.uint64(2) + np.uint64(1),
              rng.integers(0, 1 << 63, size=256, dtype=np.uint64))
    for k in (1, 5, 21, 31, 64, 65, 130):
        for canon in (False, True):
            for tables in (None, custom):
                got = gpu_ctx.rolling_hash(bases, offsets, k, canon, tables)
                for s, g in zip(seqs, got):
                    assert np.array_equal(g, oracle.rolling_hash(s, k, canon, tables)), (k, canon, len(s), tables is None)
    # with a window (minimizers of the hash stream; the canonical path queues both strands' hashes)
    for k, w in ((5, 6), (21, 40), (31, 50), (31, 31), (64, 200), (21, 6000)):
        for canon in (False, True):
            got = gpu_ctx.rolling_hash(bases, offsets, k, canon, custom, w=w)
            for s, g in zip(seqs, got):
                assert np.array_equal(g, oracle.rolling_hash(s, k, canon, custom, w=w)), (k, w, canon, len(s))
