"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Bit-exact: integer work."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

LAYOUTS = [0, 1, 2]   # BNS_LAYOUT_KHASH, BNS_LAYOUT_BUCKET, BNS_LAYOUT_MINBUCKET


def load_world(ctx, w, layout, spaced_intended=True):
    ctx.set_encoder(w.k, w.gaps, canonicalize=w.canon, spaced_intended=spaced_intended)
    ctx.load_table(w.n_buckets, w.flags, w.keys, w.vals, layout=layout)
    ctx.load_taxonomy(w.parent)


CRAFTED = [
    b"",                                                   # empty read
    b"ACGT",                                               # shorter than k
    b"ACGTACGTACGTACGTACGTACGTACGTAC",                     # k-1
    b"ACGTACGTACGTACGTACGTACGTACGTACG",                    # exactly k
    b"ACGTACGTACGTACGTACGTACGTACGTACGN",                   # trailing N
    b"NACGTACGTACGTACGTACGTACGTACGTACG",                   # leading N
    b"acgtacgtacgtacgtacgtacgtacgtacgtacgt",               # lower case
    b"ACGTACGTACGTACGTRYKMACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT",   # IUPAC run
    b"ACGUACGUACGUACGUACGUACGUACGUACGUACGUACGU",           # U is not T (alphabet.h alias defect)
    b"A" * 40, b"T" * 40,                                  # key 0 (poly-A / canonical poly-T)
    b"N" * 50,
    bytes([200, 65, 67, 71, 84] * 20),                     # bytes >= 128 are non-ACGT (defined behaviour)
    b"ACGT" * 600,                                         # > one 2048-base chunk
]


@pytest.mark.parametrize("canon", [True, False])
def test_encode_crafted(gpu_ctx, oracle, canon):
    k = 31
    gpu_ctx.set_encoder(k, None, canonicalize=canon)
    rng = np.random.default_rng(5)
    seqs = list(CRAFTED) + [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.02, 0.2).tobytes()
                            for L in rng.integers(1, 5000, size=40)]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    got = gpu_ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        exp = oracle.encode(s, k, canon=canon)
        assert np.array_equal(g, exp), (s[:60], len(g), len(exp))


@pytest.mark.parametrize("k", [1, 5, 16, 21, 31, 32])
def test_encode_k_sweep(gpu_ctx, oracle, k):
    gpu_ctx.set_encoder(k, None, canonicalize=True)
    rng = np.random.default_rng(k)
    seqs = [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.01).tobytes() for L in rng.integers(1, 400, size=64)]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    got = gpu_ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        assert np.array_equal(g, oracle.encode(s, k, canon=True))


def test_encode_phix_pins(gpu_ctx, oracle):
    """reference test/encoding.cpp:122 (5356 k-mers) + survey-probed digests of the phiX stream."""
    import os
    name, seq = oracle.read_fasta(os.path.join(os.path.dirname(__file__), "golden", "phix.fa"))[0]
    b, o = synth.concat([np.frombuffer(seq, dtype=np.uint8)])
    gpu_ctx.set_encoder(31, None, canonicalize=False)
    fw = gpu_ctx.encode(b, o)[0]
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    cn = gpu_ctx.encode(b, o)[0]
    assert fw.size == 5356 and np.unique(fw).size == 5356 and cn.size == 5356
    assert int(fw[0]) == 0x22ff367d4e1920bc
    assert int(np.bitwise_xor.reduce(fw)) == 0x2accfc81a096f1a3
    assert int(np.bitwise_xor.reduce(cn)) == 0x015ba3c1a0eb8419


SPACED_GAPS = [
    [1, 2] + [0] * 28,                 # the mask of reference test/encoding.cpp:20-23
    [1] * 15 + [0] * 15,               # "1x15,0x15": comb 46 (BASELINE config 3)
    [0] * 29 + [40],                   # comb 71 > 64
]


@pytest.mark.parametrize("gaps", SPACED_GAPS)
def test_encode_spaced(gpu_ctx, oracle, gaps):
    k = 31
    rng = np.random.default_rng(3)
    seqs = list(CRAFTED) + [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.02).tobytes()
                            for L in rng.integers(1, 3000, size=30)]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    gpu_ctx.set_encoder(k, gaps, canonicalize=True, spaced_intended=True)
    got = gpu_ctx.encode(bases, offsets)
    for s, g in zip(seqs, got):
        assert np.array_equal(g, oracle.encode(s, k, gaps=gaps, spaced_intended=True))
    # reference behaviour through the string for_each: nothing
    gpu_ctx.set_encoder(k, gaps, canonicalize=True, spaced_intended=False)
    assert all(g.size == 0 for g in gpu_ctx.encode(bases, offsets))


def test_encode_spaced_k32_drops_all_t(gpu_ctx, oracle):
    """encoder.h:236-238: the spaced loop skips a k-mer that EQUALS the overflow marker ~0 -- with k = 32 that is the
    genuine all-T k-mer (found by tools/fuzz_gpu.py).  The contiguous path emits it (its 'filled' counter, not the value)."""
    k = 32
    seqs = [b"T" * 100, b"A" * 20 + b"T" * 90 + b"NN" + b"T" * 70, b"ACGT" * 30 + b"T" * 80]
    bases, offsets = synth.concat([np.frombuffer(s, dtype=np.uint8) for s in seqs])
    for gaps in ([1, 2, 0] * 10 + [1], [0] * 30 + [3], [2] * 31):          # run-decomposable and generic extraction paths
        gpu_ctx.set_encoder(k, gaps, canonicalize=True, spaced_intended=True)
        got = gpu_ctx.encode(bases, offsets)
        for s, g in zip(seqs, got):
            exp = oracle.encode(s, k, gaps=gaps, spaced_intended=True)
            assert np.array_equal(g, exp)
            assert not (g == np.uint64(0xFFFFFFFFFFFFFFFF)).any()
    assert got[0].size == 0                                                # poly-T: every window is the marker
    gpu_ctx.set_encoder(k, None, canonicalize=False)
    assert (gpu_ctx.encode(bases, offsets)[0] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_probe(gpu_ctx, oracle, small_world, layout):
    w = small_world
    load_world(gpu_ctx, w, layout)
    rng = np.random.default_rng(9)
    present = w.keys[(w.flags[np.arange(w.n_buckets) >> 4] >> ((np.arange(w.n_buckets) & 15) << 1)) & 3 == 0]
    q = np.concatenate([present, rng.integers(0, 1 << 62, size=5000, dtype=np.uint64),
                        np.array([0, 1, (1 << 62) - 1], dtype=np.uint64)])
    rng.shuffle(q)
    ev, ef = w.table.get_batch(q)
    gv, gf = gpu_ctx.probe(q)
    assert np.array_equal(gf, ef)
    assert np.array_equal(gv, ev)
    if layout >= 1:
        assert gpu_ctx.table_info()["n_keys"] == present.size


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("name", ["t128", "r50k", "del"])
def test_probe_reference_golden(gpu_ctx, layout, name):
    """khash arrays BUILT BY THE REFERENCE's kh_put/kh_del (tests/golden/make_golden_ref.py) probed on the GPU
    must give the reference's own kh_get answers."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "khash_ref.npz"))
    if name == "del":
        hdr, f, k, v, q, qv, qf = g["del_hdr"], g["del_flags"], g["del_keys_arr"], g["del_vals"], g["r50k_q"], g["del_qv"], g["del_qf"]
    else:
        hdr, f, k, v, q, qv, qf = (g[name + "_hdr"], g[name + "_flags"], g[name + "_keys"], g[name + "_vals"], g[name + "_q"],
                                   g[name + "_qv"], g[name + "_qf"])
    gpu_ctx.set_encoder(31, None, canonicalize=True)
    gpu_ctx.load_table(int(hdr[0]), f, k, v, layout=layout)
    gv, gf = gpu_ctx.probe(q)
    assert np.array_equal(gf, qf) and np.array_equal(gv, qv)


def check_classify(ctx, oracle, w, reads, paired=False, spaced_intended=True):
    bases, offsets = synth.concat(reads)
    exp = oracle.classify_batch(w.table, w.tax, w.k, bases, offsets, paired=paired, gaps=w.gaps, canon=w.canon,
                                spaced_intended=spaced_intended)
    got = ctx.classify(bases, offsets, paired=paired, want_hits=True)
    assert np.array_equal(got["taxon"], exp["taxon"])
    assert np.array_equal(got["missing"], exp["missing"])
    assert np.array_equal(got["ambig"], exp["ambig"])
    assert np.array_equal(got["n_hits"], exp["n_hits"])
    # the ordered hit stream (the reference's `taxa` vector) on a sample
    inc = 2 if paired else 1
    for u in range(0, len(reads) // inc, max(1, len(reads) // inc // 50)):
        s1 = reads[u * inc].tobytes()
        s2 = reads[u * inc + 1].tobytes() if paired else None
        t, m, a, hits = oracle.classify_seq(w.table, w.tax, w.k, s1, s2, gaps=w.gaps, canon=w.canon,
                                            spaced_intended=spaced_intended)
        assert np.array_equal(got["hits"][u], hits)
    # the same stream run-length encoded on the device (bns_classify_batch_runs) against an RLE of the full hit lists
    gr = ctx.classify_runs(bases, offsets, paired=paired)
    for key in ("taxon", "missing", "ambig", "n_hits"):
        assert np.array_equal(gr[key], got[key])
    total = 0
    for u, h in enumerate(got["hits"]):
        cut = np.flatnonzero(np.r_[True, h[1:] != h[:-1]]) if h.size else np.zeros(0, np.int64)
        lens = np.diff(np.r_[cut, h.size]) if h.size else np.zeros(0, np.int64)
        assert np.array_equal(gr["runs"][u][0], h[cut]) and np.array_equal(gr["runs"][u][1], lens)
        total += cut.size
    assert gr["n_runs_total"] == total
    # the same batch through the PACKED entry point (2-bit words + sparse invalid-base list, bns_pack_reads ->
    # bns_classify_batch_packed): identical per-unit results and hit streams
    import bonsai_amd
    words, bw, bm = bonsai_amd.pack_reads(bases, offsets, threads=2)
    gp = ctx.classify_packed(words, bw, bm, offsets, paired=paired, want_hits=True)
    for key in ("taxon", "missing", "ambig", "n_hits"):
        assert np.array_equal(gp[key], got[key]), "packed " + key
    assert all(np.array_equal(a, b) for a, b in zip(gp["hits"], got["hits"])), "packed hits"
    return got


@pytest.mark.parametrize("layout", LAYOUTS)
def test_classify_single(gpu_ctx, oracle, small_world, layout):
    w = small_world
    load_world(gpu_ctx, w, layout)
    rng = np.random.default_rng(21)
    reads = synth.simulate_reads(rng, w.genomes, 3000, length=150, sub_rate=0.01, n_rate=0.001, lower_rate=0.05)
    got = check_classify(gpu_ctx, oracle, w, reads)
    assert (got["taxon"] != 0).mean() > 0.5
    assert np.unique(got["taxon"]).size > 6          # leaves and internal nodes both occur


@pytest.mark.parametrize("paired", [False, True])
def test_classify_sliced_upload(gpu_ctx, oracle, small_world, paired):
    """bns_classify_batch uploads a large batch in slices on a second stream while earlier slices are classified; a tiny
    slice size (debug switch BNS_DBG_SLICE_8K) forces that path (up to 16 slices, ragged reads, pairs kept together)."""
    w = small_world
    load_world(gpu_ctx, w, 2)
    rng = np.random.default_rng(29)
    reads = synth.simulate_reads(rng, w.genomes, 2001, length=150, sub_rate=0.01, n_rate=0.002)
    reads += [synth.rand_seq(rng, int(L)) for L in rng.integers(0, 700, size=60)]
    if paired and len(reads) % 2:
        reads.append(synth.rand_seq(rng, 77))
    gpu_ctx.debug_set(0x4000)
    try:
        if paired:
            bases, offsets = synth.concat(reads)
            got = gpu_ctx.classify(bases, offsets, paired=True, want_hits=True)
            gpu_ctx.debug_set(0)
            ref = gpu_ctx.classify(bases, offsets, paired=True, want_hits=True)
            for key in ("taxon", "missing", "ambig", "n_hits"):
                assert np.array_equal(got[key], ref[key])
            assert all(np.array_equal(a, b) for a, b in zip(got["hits"], ref["hits"]))
            gpu_ctx.debug_set(0x4000)
            check_classify(gpu_ctx, oracle, w, reads, paired=True)
        else:
            check_classify(gpu_ctx, oracle, w, reads)
    finally:
        gpu_ctx.debug_set(0)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_classify_paired(gpu_ctx, oracle, small_world, layout):
    w = small_world
    load_world(gpu_ctx, w, layout)
    rng = np.random.default_rng(22)
    reads = synth.simulate_reads(rng, w.genomes, 2000, length=150, n_rate=0.002)
    check_classify(gpu_ctx, oracle, w, reads, paired=True)


@pytest.mark.parametrize("layout", [1, 2])
def test_classify_ragged_and_edge(gpu_ctx, oracle, small_world, layout):
    w = small_world
    load_world(gpu_ctx, w, layout)
    rng = np.random.default_rng(23)
    reads = [np.frombuffer(s, dtype=np.uint8) for s in CRAFTED]
    reads += synth.simulate_reads(rng, w.genomes, 500, length=150, var_len=True, n_rate=0.01)
    reads += [w.genomes[1001][:5000], synth.revcomp(w.genomes[2002])]          # long reads: several chunks
    check_classify(gpu_ctx, oracle, w, reads)
    if len(reads) % 2:
        reads = reads[:-1]
    check_classify(gpu_ctx, oracle, w, reads, paired=True)                     # short mates: u32 ambig wrap (F: A:4294967238)


@pytest.mark.parametrize("layout", [1, 2])
def test_classify_noncanonical(gpu_ctx, oracle, layout):
    w = synth.make_world(oracle, seed=12, k=31, genome_len=3000, canon=False)
    load_world(gpu_ctx, w, layout)
    reads = synth.simulate_reads(np.random.default_rng(24), w.genomes, 800)
    check_classify(gpu_ctx, oracle, w, reads)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("k", [9, 15, 16, 21, 32])
def test_classify_other_k(gpu_ctx, oracle, k, layout):
    w = synth.make_world(oracle, seed=13, k=k, genome_len=3000)
    load_world(gpu_ctx, w, layout)
    reads = synth.simulate_reads(np.random.default_rng(25), w.genomes, 800)
    check_classify(gpu_ctx, oracle, w, reads)


def test_minbucket_unhashable_buckets(gpu_ctx, oracle, small_world):
    """Buckets for which no perfect-hash multiplier is found (two keys with one fold: about one bucket in 10^8) have their
    keys moved to the overflow table.  The debug switch makes every 61st bucket pretend to be one."""
    w = small_world
    gpu_ctx.debug_set(0x100)
    try:
        load_world(gpu_ctx, w, 2)
        st = gpu_ctx.table_stats()
        assert st["n_overflow_keys"] > 20
        present = w.keys[[i for i in range(w.n_buckets) if ((int(w.flags[i >> 4]) >> ((i & 15) << 1)) & 3) == 0]]
        gv, gf = gpu_ctx.probe(present)
        assert gf.all()
        reads = synth.simulate_reads(np.random.default_rng(5), w.genomes, 1500, length=150, sub_rate=0.01)
        check_classify(gpu_ctx, oracle, w, reads)
    finally:
        gpu_ctx.debug_set(0)


@pytest.mark.parametrize("dbg", [0x20 | 0x8000, 0x20 | 0x2000, 0x10 | 0x8000, 0x20 | 0x8000 | 0x100])
@pytest.mark.parametrize("load_pct", [45, 80])
def test_group_fill_reads_with_errors(gpu_ctx, oracle, small_world, dbg, load_pct):
    """Crowded clustered table, reads with substitutions: a k-mer over an error is (nearly always) not in the db and its minimizer
    is one nobody put into the table -- with the tag bits such a lookup ends at its home bucket whatever the bucket's other groups
    did.  Group-aware and arrival-order fill, tag bits on and off, and (0x100) buckets whose keys were moved to the overflow
    table: the oracle's answers every time, single and paired."""
    w = small_world
    n_keys = int(w.table.header()[1])
    gpu_ctx.debug_set(dbg)
    gpu_ctx.set_table_buckets(max(16, n_keys * 10 // load_pct))        # 10 slots per bucket
    try:
        load_world(gpu_ctx, w, 2)
        st = gpu_ctx.table_stats()
        assert st["n_keys"] == n_keys
        geo = gpu_ctx.table_geometry()
        assert geo["group_fill"] == (1 if dbg & 0x20 else 0) and geo["spilled_keys"] > 0
        rng = np.random.default_rng(77)
        check_classify(gpu_ctx, oracle, w, synth.simulate_reads(rng, w.genomes, 1200, length=150, sub_rate=0.02))
        check_classify(gpu_ctx, oracle, w, synth.simulate_reads(rng, w.genomes, 800, length=101, sub_rate=0.05), paired=True)
        present = w.keys[(w.flags[np.arange(w.n_buckets) >> 4] >> ((np.arange(w.n_buckets) & 15) << 1)) & 3 == 0]
        gv, gf = gpu_ctx.probe(present)
        ev, ef = w.table.get_batch(present)
        assert gf.all() and np.array_equal(gv, ev)
    finally:
        gpu_ctx.debug_set(0)
        gpu_ctx.set_table_buckets(0)


def test_minbucket_equal_fold_keys(gpu_ctx, oracle):
    """Two distinct keys with the same 32-bit fold can never get different perfect-hash slots: a bucket holding such a pair
    must take the overflow-table route for real (no debug switch).  Spaced seed => the bucket is a hash of the key itself,
    so in a 4-bucket table some of 12 crafted pairs share a bucket."""
    k = 32
    gaps = [1] + [0] * 30
    rng = np.random.default_rng(2024)
    a = rng.integers(0, 1 << 63, size=12, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=12, dtype=np.uint64)
    # flip bit 20 of the low word and bit (20 - 15) of the high word: lo ^ rotl(hi, 15) is unchanged
    b = a ^ np.uint64(1 << 20) ^ np.uint64(1 << (32 + 5))
    lo = (a & np.uint64(0xFFFFFFFF)).astype(np.uint32); hi = (a >> np.uint64(32)).astype(np.uint32)
    fold = lambda lo_, hi_: lo_ ^ ((hi_ << np.uint32(15)) | (hi_ >> np.uint32(17)))
    assert np.array_equal(fold(lo, hi), fold((b & np.uint64(0xFFFFFFFF)).astype(np.uint32), (b >> np.uint64(32)).astype(np.uint32)))
    keys = np.concatenate([a, b])
    keys = keys[keys != np.uint64(0xFFFFFFFFFFFFFFFF)]
    vals = (np.arange(keys.size) % 5 + 1).astype(np.uint32)
    table = oracle.Table()
    table.insert_many(keys, vals)
    flags, tk, tv = table.arrays()
    tax = oracle.Taxonomy(pairs=[(1, 1)] + [(i, 1) for i in range(2, 7)])
    gpu_ctx.set_encoder(k, gaps, canonicalize=True, spaced_intended=True)
    gpu_ctx.set_bucket_slots_log2(5)
    try:
        gpu_ctx.load_table(table.n_buckets, flags, tk, tv, layout=2)
    finally:
        gpu_ctx.set_bucket_slots_log2(0)
    gpu_ctx.load_taxonomy(tax.parent)
    st = gpu_ctx.table_stats()
    assert st["n_keys"] == keys.size and st["n_overflow_keys"] >= 2      # at least one pair had to go to the overflow table
    gv, gf = gpu_ctx.probe(keys)
    assert gf.all() and np.array_equal(gv, vals)
    miss = keys ^ np.uint64(1 << 3)                                        # neighbours of the keys: not present
    miss = miss[~np.isin(miss, keys)]
    gv, gf = gpu_ctx.probe(miss)
    assert not gf.any()


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_classify_all_ones_key(gpu_ctx, oracle, layout):
    """k = 32, not canonical: the 32-mer TTTT...T is the key 0xFFFF...F, the value the minimizer buckets pad their unused
    slots with.  It must be found when it is in the db and must NOT be found (in the padding) when it is not."""
    k = 32
    tax = oracle.Taxonomy(pairs=[(1, 1), (2, 1), (3, 1)])
    rng = np.random.default_rng(77)
    polyt = np.frombuffer(b"T" * 60, dtype=np.uint8)
    other = synth.rand_seq(rng, 200)
    reads = [polyt.copy(), np.concatenate([other[:80], polyt[:40], other[80:120]]), other.copy()]
    for with_polyt in (True, False):
        table = oracle.Table()
        oracle.lca_map_add(table, tax, k, other.tobytes(), 2, canon=False)
        if with_polyt:
            oracle.lca_map_add(table, tax, k, polyt.tobytes(), 3, canon=False)
        w = synth.World()
        w.k, w.gaps, w.canon, w.tax, w.table = k, None, False, tax, table
        w.flags, w.keys, w.vals = table.arrays()
        w.n_buckets, w.parent = table.n_buckets, tax.parent
        load_world(gpu_ctx, w, layout)
        got = check_classify(gpu_ctx, oracle, w, reads)
        assert got["taxon"][0] == (3 if with_polyt else 0)


@pytest.mark.parametrize("layout", [1, 2])
def test_classify_spaced(gpu_ctx, oracle, layout):
    gaps = [1] * 15 + [0] * 15
    w = synth.make_world(oracle, seed=14, k=31, genome_len=3000, gaps=gaps)
    load_world(gpu_ctx, w, layout)
    reads = synth.simulate_reads(np.random.default_rng(26), w.genomes, 800, n_rate=0.003)
    got = check_classify(gpu_ctx, oracle, w, reads, paired=True)
    assert (got["taxon"] != 0).mean() > 0.3
    # reference (F7) behaviour: every read unclassified, ambig = l - c + 1 arithmetic kept
    load_world(gpu_ctx, w, layout, spaced_intended=False)
    check_classify(gpu_ctx, oracle, w, reads, spaced_intended=False)


@pytest.mark.parametrize("layout", [1, 2])
def test_classify_many_taxa_overflow(gpu_ctx, oracle, layout):
    """More than LDS_CAP (128) distinct taxa in one read: the overflow kernel must agree."""
    rng = np.random.default_rng(31)
    k = 31
    n_leaves = 700
    pairs = [(1, 1)] + [(10 + i, 1) for i in range(10)] + [(1000 + i, 10 + i % 10) for i in range(n_leaves)]
    tax = oracle.Taxonomy(pairs=pairs)
    table = oracle.Table()
    segs = []
    for i in range(n_leaves):
        s = synth.rand_seq(rng, 40)
        segs.append(s)
        oracle.lca_map_add(table, tax, k, s.tobytes(), 1000 + i)
    w = synth.World()
    w.k, w.gaps, w.canon, w.tax, w.table = k, None, True, tax, table
    w.flags, w.keys, w.vals = table.arrays()
    w.n_buckets, w.parent = table.n_buckets, tax.parent
    load_world(gpu_ctx, w, layout)
    long_read = np.concatenate(segs)                        # 700 taxa x 10 hits each
    tie_read = np.concatenate(segs[:300])
    reads = [long_read, tie_read, segs[0], np.concatenate(segs[:5]), long_read[::-1].copy()]
    # the counter keeps 64 entries in registers, the next 64 in LDS, the rest goes through the overflow kernel: walk the edges
    reads += [np.concatenate(segs[a:a + n]) for a, n in ((0, 63), (3, 64), (5, 65), (7, 100), (11, 127), (13, 128), (17, 129))]
    got = check_classify(gpu_ctx, oracle, w, reads)
    assert got["taxon"][0] == 1                              # 700-way tie folds to the root


def test_resolve_known_answers(gpu_ctx, oracle, small_world):
    w = small_world
    gpu_ctx.load_taxonomy(w.parent)
    cases = [
        ([1001], [5]), ([1001, 1002], [3, 3]), ([1001, 1002], [4, 3]), ([1001, 1003], [2, 2]),
        ([1001, 2001], [1, 1]), ([101, 1001], [7, 1]), ([1001, 101], [1, 7]), ([1, 1001, 2001], [9, 1, 1]),
        ([0], [4]), ([0, 1001], [9, 1]), ([777777], [3]), ([777777, 1001], [3, 3]), ([5], [2]),
        ([1001, 1002, 1003], [2, 2, 2]), ([1001, 1002, 1004], [1, 1, 1]), ([11, 12], [5, 5]),
        ([1001], [0]), ([1001, 1002], [0, 0]), ([4294967295, 1001], [2, 2]),
    ]
    rng = np.random.default_rng(41)
    ids = np.array([0, 1, 2, 3, 11, 12, 21, 101, 102, 111, 201, 1001, 1002, 1003, 1004, 2001, 2002, 5, 999999])
    for _ in range(400):
        n = int(rng.integers(1, 9))
        ks = rng.choice(ids, size=n, replace=False)
        cases.append((ks.tolist(), rng.integers(0, 5, size=n).tolist()))
    keys = np.concatenate([np.asarray(c[0], dtype=np.uint32) for c in cases])
    counts = np.concatenate([np.asarray(c[1], dtype=np.uint16) for c in cases])
    starts = np.zeros(len(cases) + 1, dtype=np.uint64)
    starts[1:] = np.cumsum([len(c[0]) for c in cases])
    got = gpu_ctx.resolve(keys, counts, starts)
    exp = np.array([w.tax.resolve(c[0], c[1]) for c in cases], dtype=np.uint32)
    assert np.array_equal(got, exp)


def test_taxonomy_cycle_rejected(gpu_ctx):
    import bonsai_amd
    parent = np.full(8, 0xFFFFFFFF, dtype=np.uint32)
    parent[1] = 0
    parent[2], parent[3], parent[4] = 3, 4, 2
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        gpu_ctx.load_taxonomy(parent)


def test_minbucket_oversized_groups(gpu_ctx, oracle):
    """Thousands of distinct k-mers sharing one minimizer (a conserved core with random flanks in every genome): the
    chain cap sends the excess to the overflow table; lookups stay exact and bounded."""
    rng = np.random.default_rng(5)
    n_gen = 1500
    core = synth.rand_seq(rng, 60)
    pairs = [(1, 1)] + [(10 + i, 1) for i in range(n_gen)]
    tax = oracle.Taxonomy(pairs=pairs)
    table = oracle.Table()
    genomes = []
    for i in range(n_gen):
        g = np.concatenate([synth.rand_seq(rng, 45), core, synth.rand_seq(rng, 45)])
        genomes.append(g)
        oracle.lca_map_add(table, tax, 31, g.tobytes(), 10 + i)
    w = synth.World()
    w.k, w.gaps, w.canon, w.tax, w.table = 31, None, True, tax, table
    w.flags, w.keys, w.vals = table.arrays()
    w.n_buckets, w.parent = table.n_buckets, tax.parent
    load_world(gpu_ctx, w, 2)
    st = gpu_ctx.table_stats()
    assert st["n_keys"] == table.header()[1] and st["n_overflow_keys"] > 1000
    reads = [genomes[int(rng.integers(n_gen))] for _ in range(600)] + synth.simulate_reads(rng, dict(enumerate(genomes[:50])), 200, length=100)
    check_classify(gpu_ctx, oracle, w, reads)
    present = w.keys[(w.flags[np.arange(w.n_buckets) >> 4] >> ((np.arange(w.n_buckets) & 15) << 1)) & 3 == 0]
    gv, gf = gpu_ctx.probe(present)
    ev, ef = table.get_batch(present)
    assert gf.all() and np.array_equal(gv, ev)


@pytest.mark.parametrize("layout", [1, 2])
def test_classify_u16_wrap_decides_winner(gpu_ctx, oracle, small_world, layout):
    """linear::counter<tax_t, u16>::add wraps at 65 536 (linear.h:229-240): a ~70 kb read with more than 65 536 hits of one taxon
    and a few hundred of an unrelated one must classify as the SECOND -- the first one's count has wrapped to a small number
    (SURVEY 5 "long-context").  Through bns_classify_batch, both table layouts, next to ordinary reads in the same batch."""
    w = small_world
    k = 31

    def private_window(leaf, n):                     # a stretch whose k-mers all map to the leaf itself (no shared segment)
        g = w.genomes[leaf]
        for st in range(0, g.size - n, 100):
            t, m, a, hits = oracle.classify_seq(w.table, w.tax, k, g[st:st + n].tobytes())
            if m == 0 and hits.size == n - k + 1 and (hits == leaf).all():
                return g[st:st + n]
        raise AssertionError("no private window")

    seg_a = private_window(1001, 400)                # 370 hits of 1001 per copy (junction k-mers between copies are misses)
    seg_b = private_window(2001, 420)                # 390 hits of 2001, an unrelated lineage: more than any remainder n_a mod 65 536 < 370
    per = seg_a.size - k + 1
    copies = -(-65536 // per)                        # first count >= 65 536
    read = np.concatenate([seg_a] * copies + [seg_b])
    t, m, a, hits = oracle.classify_seq(w.table, w.tax, k, read.tobytes())
    n_a, n_b = int((hits == 1001).sum()), int((hits == 2001).sum())
    assert n_a >= 65536 and (n_a & 0xFFFF) < n_b, (n_a, n_b)       # the wrap really decides
    assert t == 2001
    control = np.concatenate([seg_a] * (copies - 2) + [seg_b])      # two copies fewer: no wrap, 1001 wins
    assert oracle.classify_seq(w.table, w.tax, k, control.tobytes())[0] == 1001
    reads = synth.simulate_reads(np.random.default_rng(4), w.genomes, 20) + [read, control, read[::-1].copy()]
    bases, offsets = synth.concat(reads)
    gpu_ctx.set_encoder(k, None, canonicalize=True)
    gpu_ctx.load_table(w.n_buckets, w.flags, w.keys, w.vals, layout=layout)
    gpu_ctx.load_taxonomy(w.parent)
    got = gpu_ctx.classify(bases, offsets, want_hits=True)
    exp = oracle.classify_batch(w.table, w.tax, k, bases, offsets)
    for f in ("taxon", "missing", "ambig", "n_hits"):
        assert np.array_equal(got[f], exp[f]), f
    assert got["taxon"][20] == 2001 and got["taxon"][21] == 1001 and got["n_hits"][20] > 65536


@pytest.mark.parametrize("k,gaps", [(31, [1] * 15 + [0] * 15),            # run of 16 at the low end of the key (configs[2])
                                    (31, [0] * 14 + [2] * 16),            # run of 15 at the high end
                                    (31, [1] * 8 + [0] * 12 + [3] * 10),  # run of 13 in the middle
                                    (25, [2] * 6 + [0] * 11 + [1] * 7),   # run of 12: the shortest that clusters
                                    (31, [1, 0] * 15)])                   # no run longer than 2: no clustering (m = k)
@pytest.mark.parametrize("m_force", [None, 11, 12, 14])
def test_classify_spaced_minimizer_runs(gpu_ctx, oracle, k, gaps, m_force):
    """Spaced seeds whose mask has a long run of adjacent sampled bases take their table minimizer inside that run (so that
    neighbouring spaced k-mers share buckets); the key -> value map and therefore every result must be unchanged, whatever the
    minimizer length the loader picks (forced here through the profiling override) and wherever the run sits in the key."""
    w = synth.make_world(oracle, seed=31 + k, k=k, genome_len=4000, gaps=gaps)
    gpu_ctx.debug_set((m_force or 0) << 24)              # BNS_DBG_SPACED_M
    try:
        load_world(gpu_ctx, w, 2)
    finally:
        gpu_ctx.debug_set(0)
    rng = np.random.default_rng(5)
    reads = synth.simulate_reads(rng, w.genomes, 600, n_rate=0.003, var_len=True) + [w.genomes[1001][:1500], synth.revcomp(w.genomes[2001][200:900])]
    got = check_classify(gpu_ctx, oracle, w, reads)
    assert (got["taxon"] != 0).mean() > 0.2
    check_classify(gpu_ctx, oracle, w, reads[:400], paired=True)
    # and the probe entry point on the same table: present keys and misses
    f, kk, v = w.table.arrays()
    i = np.arange(w.n_buckets)
    pres = ((f[i >> 4] >> ((i & 15) << 1)) & 3) == 0
    q = np.concatenate([kk[pres][:4000], kk[pres][:4000] ^ np.uint64(1 << 20)])
    gv, gf = gpu_ctx.probe(q)
    ev, ef = w.table.get_batch(q)
    assert np.array_equal(gv, ev) and np.array_equal(gf, ef)


@pytest.mark.parametrize("k", [9, 13, 16])
def test_wide_identity_asked_for_a_windowless_k(gpu_ctx, oracle, k):
    """bns_set_minimizer_identity(52) with a k so small that the clustered table has no minimizer window (m = k): there is nothing
    to carry the wide identity through, the loader must fall back to the narrow form (found by tools/fuzz_gpu.py: the candidate
    list came out empty and the load crashed)."""
    w = synth.make_world(oracle, seed=5 + k, k=k, genome_len=2000)
    gpu_ctx.set_minimizer_identity(52)
    try:
        load_world(gpu_ctx, w, 2)
        assert gpu_ctx.table_geometry()["identity_bits"] == 32 and gpu_ctx.table_geometry()["m"] == k
    finally:
        gpu_ctx.set_minimizer_identity(0)
    reads = synth.simulate_reads(np.random.default_rng(k), w.genomes, 400, length=100, sub_rate=0.01, n_rate=0.002)
    check_classify(gpu_ctx, oracle, w, reads)
