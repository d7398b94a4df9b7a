"""CPU tests of the oracle (oracle/bns_oracle.c) against the reference's own vectors:
  * test/encoding.cpp:122 phiX count + survey-probed stream digests
  * golden vectors produced by the reference's khash64.h / linear.h (tests/golden/make_golden_ref.py)
  * the live oracle/_ref library when it is present (build container)
  * hand-derived known answers for resolve_tree / lca (the reference has no tests for them)
"""
import os

import numpy as np
import pytest

import synth

G = os.path.join(os.path.dirname(__file__), "golden")


def slot_state(flags, n_buckets):
    i = np.arange(n_buckets)
    return (flags[i >> 4] >> ((i & 15) << 1)) & 3


# ------------------------------------------------------------------ encoder (rows 1-2)
def test_phix_reference_pins(oracle):
    name, seq = oracle.read_fasta(os.path.join(G, "phix.fa"))[0]
    assert len(seq) == 5386
    fw = oracle.encode(seq, 31, canon=False)
    cn = oracle.encode(seq, 31, canon=True)
    assert fw.size == 5356 and np.unique(fw).size == 5356          # reference test/encoding.cpp:122
    assert cn.size == 5356
    assert int(fw[0]) == 0x22ff367d4e1920bc                         # SURVEY 8c probed values
    assert int(np.bitwise_xor.reduce(fw)) == 0x2accfc81a096f1a3
    assert int(np.bitwise_xor.reduce(cn)) == 0x015ba3c1a0eb8419
    # reference test "space_case_jump" (encoding.cpp:127-151): canon set == canonical(uncanon set)
    lib = oracle.lib()
    assert set(cn.tolist()) == {lib.bo_canonical(int(x), 31) for x in fw}


def test_encoder_semantics(oracle):
    lib = oracle.lib()
    assert [lib.bo_dna4(c) for c in b"ACGTacgtNUu\x00\xff"] == [0, 1, 2, 3, 0, 1, 2, 3, -1, -1, -1, -1, -1]
    k = 5
    assert oracle.encode("ACGTA", k, canon=False).tolist() == [0b0001101100]
    assert oracle.encode("ACGT", k).size == 0
    assert oracle.encode("ACGTNACGTAC", k, canon=False).tolist() == [0b0001101100, 0b0110110001]   # N resets the window
    # revcomp: ACGTA -> TACGT
    assert lib.bo_revcomp(0b0001101100, 5) == 0b1100011011
    assert lib.bo_canonical(0b1100011011, 5) == 0b0001101100
    # k = 32 uses the full word
    s = "ACGT" * 8
    km = oracle.encode(s, 32, canon=False)
    assert km.size == 1 and int(km[0]) == int("00011011" * 8, 2)


def test_spaced_reference_vector(oracle):
    """reference test/encoding.cpp:16-32: mask {1,2,0...}, first spaced k-mer reads A-C--T..."""
    test = b"ACATGCTAGCATGCTGACTGACTGATCGATCGTA"
    gaps = [1, 2] + [0] * 28
    assert oracle.lib().bo_comb_size(np.asarray(gaps, dtype=np.uint16).ctypes.data_as(oracle.u16p), 31) == 34
    km = oracle.encode(test, 31, gaps=gaps, spaced_intended=True)
    assert km.size == 1
    picked = [test[p] for p in np.cumsum([0] + [g + 1 for g in gaps])]
    exp = 0
    for c in picked:
        exp = (exp << 2) | b"ACGT".index(c)
    assert int(km[0]) == exp
    # encoding.cpp:28-31: the decoded k-mer is the test string with positions 1, 3 and 4 dashed out
    assert bytes(picked) == bytes(c for i, c in enumerate(test) if i not in (1, 3, 4))
    # string for_each emits nothing for a spaced seed
    assert oracle.encode(test, 31, gaps=gaps, spaced_intended=False).size == 0
    # reference "qmap" test (encoding.cpp:65-88) count: len - c + 1 windows
    name, seq = oracle.read_fasta(os.path.join(G, "phix.fa"))[0]
    g2 = [1, 2, 18] + [0] * 27
    assert oracle.encode(seq, 31, gaps=g2, spaced_intended=True).size == 5386 - (31 + 21) + 1


def test_parse_spacing(oracle):
    lib = oracle.lib()
    out = np.zeros(64, dtype=np.uint16)
    assert lib.bo_parse_spacing(b"1x15,0x15", 31, out.ctypes.data_as(oracle.u16p)) == 30
    assert out[:30].tolist() == [1] * 15 + [0] * 15
    assert lib.bo_parse_spacing(b"", 31, out.ctypes.data_as(oracle.u16p)) == 30 and not out[:30].any()
    assert lib.bo_parse_spacing(b"3,1,4", 4, out.ctypes.data_as(oracle.u16p)) == 3 and out[:3].tolist() == [3, 1, 4]


# ------------------------------------------------------------------ khash (row 3) vs the reference's own code
def test_wang_golden(oracle):
    g = np.load(os.path.join(G, "khash_ref.npz"))
    lib = oracle.lib()
    assert lib.bo_wang64(1) == 0x5bca7c69b794f8ce
    assert [lib.bo_wang64(int(x)) for x in g["wang_in"]] == g["wang_out"].tolist()


@pytest.mark.parametrize("name", ["t128", "r50k"])
def test_khash_layout_matches_reference(oracle, name):
    """Same insertion sequence => bit-identical flags/keys/vals to the reference's kh_put/kh_resize."""
    g = np.load(os.path.join(G, "khash_ref.npz"))
    t = oracle.Table()
    t.insert_many(g[name + "_ins_keys"], g[name + "_ins_vals"])
    assert list(t.header()) == [int(g[name + "_hdr"][0]), int(g[name + "_hdr"][1]), int(g[name + "_hdr"][2]), int(g[name + "_hdr"][3])]
    f, k, v = t.arrays()
    st = slot_state(f, t.n_buckets)
    k[st != 0] = 0
    v[st != 0] = 0
    assert np.array_equal(f, g[name + "_flags"]) and np.array_equal(k, g[name + "_keys"]) and np.array_equal(v, g[name + "_vals"])
    qv, qf = t.get_batch(g[name + "_q"])
    assert np.array_equal(qf, g[name + "_qf"]) and np.array_equal(qv, g[name + "_qv"])
    slots = [t.get(int(x)) for x in g[name + "_q"][:2000]]
    assert slots == g[name + "_qslot"].tolist()


def test_khash_get_on_reference_arrays_with_deletions(oracle):
    """kh_get over arrays the reference produced after kh_del (deleted flags on the probe path)."""
    g = np.load(os.path.join(G, "khash_ref.npz"))
    h = g["del_hdr"]
    t = oracle.Table.wrap(int(h[0]), int(h[1]), int(h[2]), int(h[3]), g["del_flags"].copy(), g["del_keys_arr"].copy(), g["del_vals"].copy())
    qv, qf = t.get_batch(g["r50k_q"])
    assert np.array_equal(qf, g["del_qf"]) and np.array_equal(qv, g["del_qv"])
    assert (slot_state(g["del_flags"], int(h[0])) == 1).sum() == g["del_keys"].size


def test_khash_live_reference(oracle):
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(77)
    keys = np.unique(rng.integers(0, 1 << 62, size=20000, dtype=np.uint64))
    rng.shuffle(keys)
    vals = rng.integers(0, 1 << 32, size=keys.size, dtype=np.uint32)
    h = R.ref_khc_new()
    R.ref_khc_insert(h, keys.ctypes.data_as(oracle.u64p), vals.ctypes.data_as(oracle.u32p), keys.size)
    t = oracle.Table()
    t.insert_many(keys, vals)
    q = np.concatenate([keys[:5000], rng.integers(0, 1 << 62, size=5000, dtype=np.uint64)])
    rv = np.zeros(q.size, dtype=np.uint32); rf = np.zeros(q.size, dtype=np.uint8)
    R.ref_khc_get_batch(h, q.ctypes.data_as(oracle.u64p), q.size, rv.ctypes.data_as(oracle.u32p), rf.ctypes.data_as(oracle.u8p))
    ov, of = t.get_batch(q)
    assert np.array_equal(ov, rv) and np.array_equal(of, rf)
    R.ref_khc_free(h)


# ------------------------------------------------------------------ counter (row 4)
def test_counter_golden(oracle):
    import ctypes as C
    g = np.load(os.path.join(G, "counter_ref.npz"))
    lib = oracle.lib()

    class Ctr(C.Structure):
        _fields_ = [("keys", oracle.u32p), ("vals", oracle.u16p), ("n", C.c_uint32), ("m", C.c_uint32)]
    lib.bo_counter_add.restype = C.c_uint32
    lib.bo_counter_count.restype = C.c_uint16
    for adds, ek, ev in ((g["adds"], g["keys"], g["vals"]), (np.full(70000, 9, dtype=np.uint32), g["wrap_keys"], g["wrap_vals"])):
        c = Ctr()
        lib.bo_counter_init(C.byref(c))
        for a in adds.tolist():
            lib.bo_counter_add(C.byref(c), C.c_uint32(a))
        assert c.n == ek.size
        assert [c.keys[i] for i in range(c.n)] == ek.tolist()          # insertion order
        assert [c.vals[i] for i in range(c.n)] == ev.tolist()          # u16 wrap: 70000 -> 4464
        assert lib.bo_counter_count(C.byref(c), C.c_uint32(123456)) == 0
        lib.bo_counter_free(C.byref(c))
    assert g["wrap_vals"].tolist() == [70000 % 65536]


# ------------------------------------------------------------------ lca / resolve_tree (rows 5-6): hand-derived
@pytest.fixture(scope="module")
def tax(oracle):
    return oracle.Taxonomy(pairs=synth.TAX_PAIRS)


def test_lca_known_answers(tax):
    assert tax.lca(1001, 1001) == 1001
    assert tax.lca(1001, 0) == 1001 and tax.lca(0, 1001) == 1001
    assert tax.lca(1001, 1002) == 101
    assert tax.lca(1001, 1003) == 11
    assert tax.lca(1001, 1004) == 2
    assert tax.lca(1001, 2001) == 1
    assert tax.lca(101, 1001) == 101 and tax.lca(1001, 101) == 101
    assert tax.lca(1, 2002) == 1
    assert tax.lca(777777, 1001) == 0xFFFFFFFF          # "Missing taxid": (tax_t)-1, util.h:649-650
    assert tax.lca(1001, 777777) == 0xFFFFFFFF
    assert tax.lca(0xFFFFFFFF, 1001) == 0xFFFFFFFF


def test_resolve_known_answers(tax):
    r = tax.resolve
    assert r([], []) == 0
    assert r([1001], [5]) == 1001
    assert r([1001, 1002], [4, 3]) == 1001               # 4 vs 3
    assert r([1001, 1002], [3, 3]) == 101                # tie -> lca
    assert r([1001, 1003], [2, 2]) == 11
    assert r([1001, 2001], [1, 1]) == 1
    assert r([101, 1001], [7, 1]) == 1001                # root-to-leaf sum: 1001 scores 8, 101 scores 7
    assert r([1001, 101], [1, 7]) == 1001                # insertion order does not matter
    assert r([101, 1001, 1002], [5, 1, 1]) == 101        # 1001 and 1002 tie at 6 -> lca = 101
    assert r([1001, 1002, 1003], [2, 2, 2]) == 11        # 3-way tie
    assert r([11, 12], [5, 5]) == 2
    assert r([0], [4]) == 0                              # taxon 0 has an empty chain
    assert r([0, 1001], [9, 1]) == 1001
    assert r([777777], [3]) == 777777                    # not in the map: own count, walk stops (defined behaviour)
    assert r([1001], [0]) == 1001                        # zero score ties with the initial max: lca(0,t)=t
    assert r([1, 2001, 2002], [1, 3, 3]) == 201


def test_resolve_order_independent(oracle, tax):
    rng = np.random.default_rng(5)
    ids = np.array([1, 2, 3, 11, 12, 21, 101, 102, 111, 201] + synth.LEAVES)
    for _ in range(300):
        n = int(rng.integers(1, 8))
        ks = rng.choice(ids, size=n, replace=False)
        cs = rng.integers(1, 6, size=n)
        a = tax.resolve(ks, cs)
        p = rng.permutation(n)
        assert tax.resolve(ks[p], cs[p]) == a


def test_nodes_dmp_loader(oracle, tmp_path):
    p = tmp_path / "nodes.dmp"
    synth.write_nodes_dmp(str(p))
    with open(p, "a") as f:
        f.write("# a comment line\n\n5\t|\t3\t|\tspecies\t|\n")
    t = oracle.Taxonomy(path=str(p))
    par = t.parent
    assert par[1] == 0                                   # util.h:780-781 forces the root
    assert par[5] == 3 and par[1001] == 101 and par[4] == 0xFFFFFFFF
    bad = tmp_path / "bad.dmp"
    bad.write_text("7\n")
    with pytest.raises(ValueError):
        oracle.Taxonomy(path=str(bad))


# ------------------------------------------------------------------ classify_seq + formatting + db IO
def test_classify_seq_and_kraken_line(oracle, small_world):
    w = small_world
    g = w.genomes[1001]
    read = g[100:250].tobytes()
    taxon, missing, ambig, hits = oracle.classify_seq(w.table, w.tax, 31, read)
    assert hits.size + missing == 120 and ambig == 0 and taxon != 0
    line = oracle.kraken_line("r1", taxon, 150, missing, ambig, hits)
    fields = line.decode().rstrip("\n").split("\t")
    assert fields[0] == "C" and fields[1] == "r1" and int(fields[2]) == taxon and fields[3] == "150"
    runs = [f for f in fields[4:] if not f.startswith(("M:", "A:"))]
    assert sum(int(r.split(":")[1]) for r in runs) == hits.size
    assert line.endswith(b"\n") and b"\t\n" not in line
    # unclassified: "U\tname\t0\tlen\tM:n\t0:0\n"
    rnd = synth.rand_seq(np.random.default_rng(1), 60).tobytes()
    t2, m2, a2, h2 = oracle.classify_seq(w.table, w.tax, 31, rnd)
    assert t2 == 0 and m2 == 30 and h2.size == 0
    assert oracle.kraken_line("x", t2, 60, m2, a2, h2) == b"U\tx\t0\t60\tM:30\t0:0\n"
    # paired ambig arithmetic wraps in u32 exactly as classifier.h:235 (A:4294967238 style)
    t3, m3, a3, h3 = oracle.classify_seq(w.table, w.tax, 31, g[0:150].tobytes(), g[300:450].tobytes())
    assert a3 == (150 - 31 + 1 - 120 + 150 - 30 - 240) % (1 << 32)
    # ambiguous bases
    s = bytearray(g[500:650].tobytes()); s[75] = ord("N")
    t4, m4, a4, h4 = oracle.classify_seq(w.table, w.tax, 31, bytes(s))
    assert a4 == 31 and h4.size + m4 == 120 - 31


def test_db_roundtrip_both_spacing_widths(oracle, small_world, tmp_path):
    w = small_world
    for width, name in ((1, "a.db"), (2, "b.db"), (1, "c.db.gz")):
        path = str(tmp_path / name)
        assert oracle.db_write(path, 31, 31, None, w.table, spacing_width=width) == 0
        k, wsz, gaps, t, got_width = oracle.db_read(path)
        assert (k, wsz, got_width) == (31, 31, width) and not gaps.any()
        assert t.header() == w.table.header()
        f0, k0, v0 = w.table.arrays()
        f1, k1, v1 = t.arrays()
        assert np.array_equal(f0, f1) and np.array_equal(k0, k1) and np.array_equal(v0, v1)
        if not name.endswith(".gz"):
            nb = w.table.n_buckets
            assert os.path.getsize(path) == 8 + 30 * width + 32 + 4 * (nb >> 4) + 12 * nb


def test_lca_map_build(oracle, small_world):
    """update_lca_map semantics (feature_min.h:205-228): shared segments store the lca of their owners."""
    w = small_world
    f, k, v = w.table.arrays()
    st = slot_state(f, w.table.n_buckets)
    present_vals = set(np.unique(v[st == 0]).tolist())
    assert set(synth.LEAVES) <= present_vals
    assert present_vals & {101, 11, 2, 1, 201}          # internal nodes from shared segments
    assert all(x in {c for c, _ in synth.TAX_PAIRS} for x in present_vals)


# ------------------------------------------------------------------ windowed minimizers (row 9)
def test_windowed_minimizers(oracle):
    """Counts pinned by the reference ("qmap" test encoding.cpp:65-88: len - w + 1; SURVEY F8: 5337 phiX windows at w=50);
    the selection rule is checked against a brute-force argmin of (score, k-mer) per window."""
    name, seq = oracle.read_fasta(os.path.join(G, "phix.fa"))[0]
    lib = oracle.lib()
    cn = oracle.encode(seq, 31, canon=True)
    for score in (oracle.SCORE_LEX, oracle.SCORE_ENTROPY_PATH):
        for w in (32, 50, 100):
            got = oracle.encode_windowed(seq, 31, w, score)
            assert got.size == len(seq) - w + 1
            ws = w - 31 + 1
            sc = np.array([lib.bo_score(int(x), score) for x in cn], dtype=np.uint64)
            exp = []
            for i in range(cn.size - ws + 1):
                j = min(range(i, i + ws), key=lambda t: (int(sc[t]), int(cn[t])))
                exp.append(int(cn[j]))
            assert got.tolist() == exp
    # SURVEY F8 closed form: score = (u64)(i64)(double(kmer) / (-1 + 1e-4))
    for km in (0, 1, 12345678901234567, (1 << 62) - 1):
        x = float(km) / (-1.0 + 1e-4)
        assert lib.bo_score(km, oracle.SCORE_ENTROPY_PATH) == (int(x) & 0xFFFFFFFFFFFFFFFF)
    # an N k-mer is ENCODE_OVERFLOW -> canonical_representation() -> 0 (encoder.h:624-625): it wins every window it is in
    s = bytearray(seq[:200]); s[100] = ord("N")
    got = oracle.encode_windowed(bytes(s), 31, 50, oracle.SCORE_ENTROPY_PATH)
    assert got.size == 151 and (got[51:101] == 0).all() and got[0] != 0
    # w <= k is the unwindowed canonical stream
    assert np.array_equal(oracle.encode_windowed(seq, 31, 31, 0), cn)


def _py_uncanon_windowed(seq, k, w, score_fn):
    """Second restatement of Encoder::for_each_uncanon_unspaced_windowed (encoder.h:273-306) + QueueMap::next_value
    (qmap.h:79-87), keeping the reference's loop shape: the base is OR-ed in before the ENCODE_OVERFLOW test."""
    from collections import deque
    M64 = (1 << 64) - 1
    lut = {c: i for i, c in enumerate(b"ACGT")}; lut.update({c: i for i, c in enumerate(b"acgt")})
    ws, mask = w - k + 1, M64 >> (64 - 2 * k)
    q, out, pos, l = deque(), [], 0, len(seq)
    restart = True
    while pos < l:
        if restart:
            mn, filled, restart = 0, 0, False
        while filled < k and pos < l:
            mn = (mn * 4) & M64
            nv = lut.get(seq[pos], -1); pos += 1
            mn |= (nv & M64)
            if mn == M64:
                restart = True
                break
            filled += 1
        if restart:
            continue
        if filled == k:
            mn &= mask
            q.append((score_fn(mn), mn))
            if len(q) > ws:
                q.popleft()
            if len(q) == ws:
                el = min(q)[1]
                if el != M64:
                    out.append(el)
            filled -= 1
    if 0 < len(q) < ws:
        out.append(min(q)[1])
    return out


def test_uncanon_windowed_restatement(oracle):
    """`bonsai build -C -w`: forward k-mers, windows over the emitted stream (across N gaps), partial-window flush, and
    the k >= 31 T-run restart.  C restatement == Python transcription == a closed form (T-restart positions treated as
    invalid bases, then windows over the surviving k-mers)."""
    lib = oracle.lib()
    rng = np.random.default_rng(77)
    cases = []
    for _ in range(60):
        L = int(rng.integers(1, 400))
        s = bytearray(rng.choice(list(b"ACGT"), size=L).tobytes())
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, L))] = ord("N")
        if rng.random() < 0.5 and L > 40:                      # T runs around the 32-base threshold
            a = int(rng.integers(0, L - 30)); n = int(rng.integers(28, 100))
            s[a:a + n] = b"T" * min(n, L - a)
        if rng.random() < 0.3:
            s = bytearray(bytes(s).lower())
        cases.append(bytes(s))
    cases += [b"T" * 200, b"A" + b"T" * 31 + b"ACGT" * 20, b"T" * 32, b"T" * 31, b"ACGT" * 10, b"", b"ACGTN" * 30]
    for k, w in ((31, 50), (31, 32), (32, 40), (30, 45), (15, 31), (4, 10)):
        for score in (oracle.SCORE_LEX, oracle.SCORE_ENTROPY_PATH):
            sf = lambda x: lib.bo_score(int(x), score)
            for s in cases:
                got = oracle.encode_windowed(s, k, w, score, canon=False).tolist()
                assert got == _py_uncanon_windowed(s, k, w, sf), (k, w, score, s)
                # closed form
                bad = [c not in b"ACGTacgt" for c in s]
                if k >= 31:
                    run = 0
                    for i, c in enumerate(s):
                        run = run + 1 if c in b"Tt" else 0
                        if run and run % 32 == 0:
                            bad[i] = True
                fw = oracle.encode(bytes(ord("N") if b else c for c, b in zip(s, bad)), k, canon=False).tolist()
                ws = w - k + 1
                exp = [min(fw[i:i + ws], key=lambda x: (sf(x), x)) for i in range(len(fw) - ws + 1)]
                if 0 < len(fw) < ws:
                    exp = [min(fw, key=lambda x: (sf(x), x))]
                assert got == exp, (k, w, score, s)
    # a clean sequence longer than w: one value per window of the forward stream
    name, seq = oracle.read_fasta(os.path.join(G, "phix.fa"))[0]
    assert oracle.encode_windowed(seq, 31, 50, 0, canon=False).size == len(seq) - 50 + 1
    assert np.array_equal(oracle.encode_windowed(seq, 31, 31, 0, canon=False), oracle.encode(seq, 31, canon=False))


def _py_entropy_str(seq, k, w, canon):
    """Transcription of for_each_uncanon_unspaced_windowed_entropy_ (encoder.h:307-346) with an explicit CircusEnt
    (entropy.h:9-60: queue of the last k symbols + counts), terms added in the order A, C, G, T."""
    import math
    from collections import deque
    M64 = (1 << 64) - 1
    lut = {c: i for i, c in enumerate(b"ACGT")}; lut.update({c: i for i, c in enumerate(b"acgt")})
    ws, mask = w - k + 1, M64 >> (64 - 2 * k)

    def to_u64(x):                                   # gcc x86-64 double -> u64
        two63 = 9223372036854775808.0
        if x != x:
            return 1 << 63
        if x < two63:
            return (int(x) & M64) if x >= -two63 else (1 << 63)
        y = x - two63
        return ((int(y) & M64) if y < two63 else (1 << 63)) ^ (1 << 63)

    def rc(x):
        r = 0
        for _ in range(k):
            r = (r << 2) | (3 - (x & 3)); x >>= 2
        return r
    q, out, pos, l = deque(), [], 0, len(seq)
    entq, cnt = deque(), [0, 0, 0, 0]
    mn = filled = 0
    if not (k - 1 < l):
        return out
    while pos < l:
        restart = False
        while filled < k and pos < l:
            nc = lut.get(seq[pos], -1); pos += 1
            if nc < 0:
                restart = True
                break
            mn = ((4 * mn) | nc) & M64
            cnt[nc] += 1                             # CircusEnt::push
            if len(entq) == k:
                cnt[entq.popleft()] -= 1
            entq.append(nc)
            filled += 1
        if restart:
            entq.clear(); cnt = [0, 0, 0, 0]; mn = filled = 0
            continue
        if filled == k:
            mn &= mask
            qi = 1. / k
            val = 0.
            for c in range(4):
                if cnt[c]:
                    val = val + cnt[c] * qi * math.log(cnt[c] * qi)
            q.append((to_u64(float(mn) / (val + .001)), mn))
            if len(q) > ws:
                q.popleft()
            if len(q) == ws and min(q)[1] != M64:
                e = min(q)[1]
                out.append(min(e, rc(e)) if canon else e)
            filled -= 1
    if 0 < len(q) < ws:
        e = min(q)[1]
        out.append(min(e, rc(e)) if canon else e)
    return out


def test_entropy_string_overload_restatement(oracle):
    """Row 9's real-entropy score (string overload): C restatement == Python transcription; entropy known answers."""
    import ctypes as C
    import math
    lib = oracle.lib()
    lib.bo_kmer_entropy.restype = C.c_double; lib.bo_kmer_entropy.argtypes = [C.c_uint64, C.c_uint]
    assert abs(lib.bo_kmer_entropy(0b00011011, 4) - math.log(0.25)) < 1e-15
    assert abs(lib.bo_kmer_entropy(0, 31)) < 1e-14                         # homopolymer: ~0
    rng = np.random.default_rng(31)
    seqs = [b"", b"ACGT" * 10, b"A" * 80, b"T" * 80, b"C" * 40 + b"ACGTTGCA" * 10, b"ACGTN" * 30, b"ACGT" * 7 + b"N" + b"GATTACA" * 12,
            b"acgtacgtac" * 9]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, 0.01, 0.1).tobytes() for L in rng.integers(1, 500, size=30)]
    for k, w in ((31, 50), (31, 32), (32, 45), (15, 40), (4, 9), (21, 31 + 53)):
        for canon in (True, False):
            for s in seqs:
                got = oracle.encode_windowed_entropy_str(s, k, w, canon).tolist()
                assert got == _py_entropy_str(s, k, w, canon), (k, w, canon, s[:60])
    # a low-complexity k-mer scores high (entropy near 0 -> small denominator) and loses to a balanced one
    s = b"A" * 40 + b"ACGTGCTAGCTAGGATCCGATCGATTAGCGCGATATCGG"
    got = oracle.encode_windowed_entropy_str(s, 15, 30, canon=False)
    assert got.size == len(s) - 30 + 1


# ---- RollingHasher (parity unpinned: the character tables are un-vendored, F10) ---------------------
def _py_rolling(seq: bytes, k: int, canon: bool, tf, tr, w=0, score=None):
    """Without a window: min(h, g) / h per position.  With w > k (encoder.h:706-736,771-795): every hash -- on the canonical
    path both strands', forward first -- goes through a queue of w-k+1 entries, the entry with the smallest (score, value)
    comes out once it is full, and a queue that never filled flushes its minimum."""
    pairs = _py_rolling_pairs(seq, k, canon, tf, tr)
    if w <= k:
        return [min(h, g) if canon else h for h, g in pairs]
    from collections import deque
    ws, q, out = w - k + 1, deque(), []
    for h, g in pairs:
        for v in ((h, g) if canon else (h,)):
            q.append((score(v), v))
            if len(q) > ws:
                q.popleft()
            if len(q) == ws and min(q)[1] != (1 << 64) - 1:
                out.append(min(q)[1])
    if 0 < len(q) < ws:
        out.append(min(q)[1])
    return out


def _py_rolling_pairs(seq: bytes, k: int, canon: bool, tf, tr):
    """Character-level Python transcription of encoder.h:692-796 without a window -- independent of oracle/bns_oracle.c's
    structured loops (this one keeps the reference's two loops and its `goto fixup` as a state flag)."""
    M = (1 << 64) - 1
    rotl = lambda x, r: ((x << (r % 64)) | (x >> (64 - r % 64))) & M if r % 64 else x
    rotr = lambda x, r: ((x >> (r % 64)) | (x << (64 - r % 64))) & M if r % 64 else x
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
    rcc = lambda c: 3 - code[c] if c in code else 255
    l = len(seq)
    out = []
    if l < k:
        return out
    myr = k % 64
    i = nf = 0
    h = g = 0
    filling = True
    while True:
        if filling:
            if not (nf < k and i < l):
                if nf < k:
                    return out
                out.append((h, g))
                filling = False
                continue
            c = seq[i]
            if c not in code:
                if canon and i + 2 * k >= l:
                    return out
                i += k; nf = 0; h = g = 0
            else:
                h = rotl(h, 1) ^ int(tf[code[c]])
                if canon:
                    g = rotl(g, 1) ^ int(tr[rcc(seq[i - nf + k - 1])])
                nf += 1
            i += 1
        else:
            if i >= l:
                return out
            c = seq[i]
            if c not in code:                       # goto fixup
                if canon and i + 2 * k >= l:
                    return out
                i += k; nf = 0; h = g = 0
                i += 1
                filling = True
                continue
            h = rotl(h, 1) ^ rotl(int(tf[code[seq[i - k]]]), myr) ^ int(tf[code[c]])
            if canon:
                g ^= rotl(int(tr[rcc(c)]), myr) ^ int(tr[rcc(seq[i - k])])
                g = rotr(g, 1)
            out.append((h, g))
            i += 1


def test_rolling_hasher_restatement(oracle):
    tf, tr = oracle.rolling_tables()
    assert np.unique(tf).size == 256 and not np.array_equal(tf, tr)
    rng = np.random.default_rng(8)
    seqs = [b"", b"ACGT", b"ACGTACGTACGTACGTACGTA", b"A" * 100, b"ACGTN" * 30, b"N" * 50 + b"ACGT" * 20, b"ACGT" * 20 + b"N",
            b"acgtACGT" * 10 + b"N" + b"TTGACCA" * 12]
    seqs += [synth.mutate(rng, synth.rand_seq(rng, int(L)), 0.0, float(r), 0.1).tobytes()
             for L, r in zip(rng.integers(1, 700, size=40), rng.choice([0, 0.01, 0.05], size=40))]
    for k in (1, 4, 21, 31, 63, 64, 65, 100):
        for canon in (False, True):
            for s in seqs:
                got = oracle.rolling_hash(s, k, canon, (tf, tr))
                exp = np.array(_py_rolling(s, k, canon, tf, tr), dtype=np.uint64)
                assert np.array_equal(got, exp), (k, canon, len(s))
    # with a window: minimizers of the hash stream by (lex_score, value)
    lib = oracle.lib()
    sc = lambda v: lib.bo_score(int(v), oracle.SCORE_LEX)
    for k, w in ((4, 10), (21, 30), (31, 50), (31, 32), (63, 70)):
        for canon in (False, True):
            for s in seqs:
                got = oracle.rolling_hash(s, k, canon, (tf, tr), w=w)
                exp = np.array(_py_rolling(s, k, canon, tf, tr, w=w, score=sc), dtype=np.uint64)
                assert np.array_equal(got, exp), (k, w, canon, len(s))
    clean = synth.rand_seq(rng, 300).tobytes()
    assert oracle.rolling_hash(clean, 21, False, (tf, tr), w=40).size == (300 - 20) - 20 + 1        # n - ws + 1 windows
    assert oracle.rolling_hash(clean, 21, True, (tf, tr), w=40).size == 2 * (300 - 20) - 20 + 1     # both strands queued
    assert oracle.rolling_hash(clean[:30], 21, False, (tf, tr), w=40).size == 1                     # partial flush
    # the forward hash of a window does not depend on what came before it: k-mers equal as strings hash alike
    s = b"GATTACAGATTACAGATTACA" * 3
    v = oracle.rolling_hash(s, 7, False, (tf, tr))
    assert v.size == len(s) - 6 and v[0] == v[7] == v[14]
    # reference test "rolling" (test/encoding.cpp:156): one value per k-window of a clean sequence
    assert oracle.rolling_hash(synth.rand_seq(rng, 500).tobytes(), 21, True, (tf, tr)).size == 480
