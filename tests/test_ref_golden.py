"""CPU tier: the oracle restatement and the C++ host code against vectors produced by the REFERENCE's own code.

tests/golden/{resolve,classify,ingest}_ref.npz were written by tests/golden/make_golden_tree.py from oracle/_ref/libbns_ref.so,
i.e. from lca / resolve_tree / build_parent_map / update_lca_map / khash_write_impl / reverse_complement /
canonical_representation / the Kraken + FASTQ formatters / kseq_read + bseq_read compiled out of the reference checkout
(oracle/ref_extract.py).  Nothing here needs the reference at run time; when oracle/_ref happens to be present (the build
container) a live cross-check runs as well.
"""
import gzip
import os

import numpy as np
import pytest

import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLD, "resolve_ref.npz"))


@pytest.fixture(scope="module")
def CL():
    return np.load(os.path.join(GOLD, "classify_ref.npz"))


@pytest.fixture(scope="module")
def IN():
    return np.load(os.path.join(GOLD, "ingest_ref.npz"))


@pytest.fixture(scope="module")
def hostio():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from bonsai_amd.build import build_device_library
    build_device_library()
    subprocess.run(["make", "-s", "-C", os.path.join(root, "bonsai_amd", "csrc", "host")], check=True)
    from bonsai_amd import hostio
    return hostio


def forest_cases(G):
    for fi in range(int(G["n_forests"])):
        g = lambda n: G["f%d_%s" % (fi, n)]          # noqa: E731
        yield fi, g("child"), g("parent"), g("keys"), g("counts"), g("offs"), g("expected"), g("lca_a"), g("lca_b"), g("lca")


# ------------------------------------------------------------------ rows 1-2: the k-mer stream (reference LUT / mask / canonical form)

@pytest.fixture(scope="module")
def ST():
    return np.load(os.path.join(GOLD, "stream_ref.npz"))


def stream_cases(ST):
    strs = [bytes(ST["str_bytes"][int(a):int(b)]) for a, b in zip(ST["str_offs"][:-1], ST["str_offs"][1:])]
    for k in ST["ks"].tolist():
        for canon in (0, 1):
            vals, cnt = ST["s_k%d_c%d" % (k, canon)], ST["n_k%d_c%d" % (k, canon)]
            ends = np.cumsum(cnt)
            for i, s in enumerate(strs):
                yield k, canon, s, vals[int(ends[i] - cnt[i]):int(ends[i])]


def test_kmer_stream_reference_vectors(oracle, ST):
    """The oracle's Encoder::for_each stream == vectors made by the REFERENCE's own symbol table (alphabet.h DNA4 -- with its
    "U:T" alias resolving to -1, as make_lut really does), rhmask, mul and canonical_representation, driven by the restated loop
    of encoder.h:246-271 (tests/golden/make_golden_stream.py): alphabet, bit order (first base in the high bits), mask and
    canonical form are pinned to reference code, for phiX and 30 crafted strings (N runs, lower case, IUPAC, NUL, white space,
    shorter than / equal to k) at k = 1, 5, 16, 21, 31, 32."""
    lib = oracle.lib()
    lut = ST["lut"]
    assert {chr(i): int(v) for i, v in enumerate(lut) if v != -1} == {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}
    assert [lib.bo_dna4(i) for i in range(128)] == lut[:128].tolist()
    assert ST["masks"].tolist() == [(1 << (2 * k)) - 1 for k in range(1, 33)]
    _, phix = oracle.read_fasta(os.path.join(GOLD, "phix.fa"))[0]
    assert np.array_equal(oracle.encode(phix, 31, canon=False), ST["phix_fw31"])
    assert np.array_equal(oracle.encode(phix, 31, canon=True), ST["phix_cn31"])
    n = 0
    for k, canon, s, exp in stream_cases(ST):
        got = oracle.encode(s, k, canon=bool(canon))
        assert np.array_equal(got, exp), (k, canon, s[:40])
        n += exp.size
    assert n > 40000


# ------------------------------------------------------------------ rows 5-6: resolve_tree / lca (reference code)

def test_resolve_tree_reference_vectors(oracle, G):
    """bo_resolve_tree == the reference's resolve_tree (util.h:831-869) on 52 000 counters over 9 random forests:
    2-/3-/many-way ties, nested winners, > 128 distinct taxa, u16-range counts and the u16 wrap (a count of 65536 is 0)."""
    total = 0
    for fi, child, parent, keys, counts, offs, exp, *_ in forest_cases(G):
        tax = oracle.Taxonomy(pairs=list(zip(child.tolist(), parent.tolist())))
        c16 = counts.astype(np.uint16)               # linear::counter<tax_t,u16>: add() wraps at 65536
        got = np.array([tax.resolve(keys[int(a):int(b)], c16[int(a):int(b)]) for a, b in zip(offs[:-1], offs[1:])], dtype=np.uint32)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, "forest %d: %d mismatches, first case %d keys %s counts %s: oracle %d reference %d" % (
            fi, bad.size, bad[0], keys[int(offs[bad[0]]):int(offs[bad[0] + 1])], counts[int(offs[bad[0]]):int(offs[bad[0] + 1])],
            got[bad[0]], exp[bad[0]])
        total += exp.size
    assert total >= 50000


def test_lca_reference_vectors(oracle, G):
    """bo_lca == the reference's lca (util.h:634-663): relatives, different trees (-> 1), identical, 0, missing ids (-> -1)."""
    n = miss = 0
    for fi, child, parent, *_rest in forest_cases(G):
        a, b, exp = _rest[-3], _rest[-2], _rest[-1]
        tax = oracle.Taxonomy(pairs=list(zip(child.tolist(), parent.tolist())))
        got = np.array([tax.lca(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint32)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, "forest %d: lca(%d,%d) oracle %d reference %d" % (fi, a[bad[0]], b[bad[0]], got[bad[0]], exp[bad[0]])
        n += exp.size; miss += int((exp == 0xFFFFFFFF).sum())
    assert n >= 25000 and miss > 100


def test_revcomp_canonical_reference_vectors(oracle, G):
    L = oracle.lib()
    for x, k, rc, cn in zip(G["rc_in"].tolist(), G["rc_k"].tolist(), G["rc_out"].tolist(), G["canon_out"].tolist()):
        assert L.bo_revcomp(x, k) == rc and L.bo_canonical(x, k) == cn


# ------------------------------------------------------------------ db, reads, records, bytes

def oracle_db(oracle, CL):
    tax = oracle.Taxonomy(pairs=list(zip(CL["tax_child"].tolist(), CL["tax_parent"].tolist())))
    t = oracle.Table()
    gb, go = CL["genome_bases"], CL["genome_offs"]
    for i, leaf in enumerate(CL["genome_taxid"].tolist()):
        oracle.lca_map_add(t, tax, int(CL["k"]), gb[int(go[i]):int(go[i + 1])].tobytes(), leaf)
    return tax, t


def test_db_content_reference(oracle, CL):
    """key -> LCA-taxid map of the oracle's sequential build == what the reference's update_lca_map (feature_min.h:205-228,
    over its lca and its khash) produced for the same genomes (golden item 2)."""
    tax, t = oracle_db(oracle, CL)
    f, k, v = t.arrays()
    idx = np.arange(k.size)
    present = ((f[idx >> 4] >> ((idx & 15) << 1)) & 3) == 0
    o = np.argsort(k[present])
    assert np.array_equal(k[present][o], CL["db_keys"]) and np.array_equal(v[present][o], CL["db_vals"])
    assert set(np.unique(CL["db_vals"]).tolist()) > {1, 101, 11}        # shared segments really gave internal nodes
    # the reference-laid-out arrays answer the oracle's kh_get like the oracle's own table
    rt = oracle.Table.wrap(*[int(x) for x in CL["db_hdr"]], CL["db_flags"].copy(), CL["db_keys_arr"].copy(),
                           CL["db_vals_arr"].copy())
    q = np.concatenate([CL["db_keys"], CL["db_keys"] ^ np.uint64(0x5555)])
    v1, f1 = t.get_batch(q); v2, f2 = rt.get_batch(q)
    assert np.array_equal(v1, v2) and np.array_equal(f1, f2)


@pytest.mark.parametrize("paired", [False, True])
def test_classify_reference_vectors(oracle, CL, paired):
    """bo_classify_batch == classify_seq's body (hit lambda, ambig arithmetic, resolve_tree: classifier.h:225-238) run with the
    reference's own kh_get / linear::counter / resolve_tree on 2000 reads and 1000 pairs (golden item 3), and the Kraken line."""
    tax, t = oracle_db(oracle, CL)
    pre = "p_" if paired else "s_"
    bases, offs, exp = CL[pre + "bases"], CL[pre + "offs"], CL[pre + "res"]
    got = oracle.classify_batch(t, tax, int(CL["k"]), bases, offs, paired=paired)
    for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
        assert np.array_equal(got[f], exp[:, j]), f
    if paired:
        assert int((exp[:, 2] > 0xFFFF0000).sum()) >= 1                   # the u32 wrap of `ambig` for a short mate is in the set
    inc = 2 if paired else 1
    lines, lo = CL[pre + "lines"].tobytes(), CL[pre + "lines_offs"]
    hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
    for u in range(exp.shape[0]):
        s1 = bases[int(offs[inc * u]):int(offs[inc * u + 1])].tobytes()
        s2 = bases[int(offs[inc * u + 1]):int(offs[inc * u + 2])].tobytes() if paired else None
        tx, ms, am, h = oracle.classify_seq(t, tax, int(CL["k"]), s1, s2)
        assert np.array_equal(h, hits[int(ho[u]):int(ho[u + 1])])
        assert oracle.kraken_line("r%d" % u, tx, len(s1), ms, am, h) == lines[int(lo[u]):int(lo[u + 1])]


@pytest.mark.parametrize("paired", [False, True])
def test_host_formatters_reference_vectors(hostio, CL, paired):
    """the C++ host formatters == the reference's append_kraken_classification / append_fastq_classification
    (classifier.h:30-129) byte for byte: Kraken lines for every unit, FASTQ records (terse and verbose; FASTA input, i.e.
    qual = NULL -> the sequence is printed twice) for every tenth."""
    pre = "p_" if paired else "s_"
    bases, offs, res = CL[pre + "bases"], CL[pre + "offs"], CL[pre + "res"]
    hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
    lines, lo = CL[pre + "lines"].tobytes(), CL[pre + "lines_offs"]
    fq, fo = CL[pre + "fq"].tobytes(), CL[pre + "fq_offs"]
    inc = 2 if paired else 1
    fi = 0
    for u in range(res.shape[0]):
        s1 = bases[int(offs[inc * u]):int(offs[inc * u + 1])].tobytes()
        s2 = bases[int(offs[inc * u + 1]):int(offs[inc * u + 2])].tobytes() if paired else None
        h = hits[int(ho[u]):int(ho[u + 1])]
        name = "r%d" % u
        tx, ms, am = (int(x) for x in res[u, :3])
        assert hostio.kraken_line(name, len(s1), tx, ms, am, h) == lines[int(lo[u]):int(lo[u + 1])]
        if u % 10 == 0:
            q1 = bytes((33 + (i * 7 + u) % 40) for i in range(len(s1))) if u % 20 == 0 else None
            for verbose in (0, 1):
                m1 = (name.encode(), s1, q1)
                m2 = ((name + "_m").encode(), s2, None) if paired else None
                assert hostio.fastq_record(m1, m2, tx, ms, am, h, verbose) == fq[int(fo[fi]):int(fo[fi + 1])], (u, verbose)
                fi += 1
    assert fi == fo.size - 1


def test_db_table_bytes_reference(oracle, hostio, CL, tmp_path):
    """the table section of a bns.db (n_buckets, n_occupied, size, upper_bound, flags, keys, vals) as the oracle's and the C++
    host's writers emit it == the reference's khash_write_impl (util.h:279-294) bytes (golden item 6); both spacing widths of
    the header in front of it (database.h:46-48 reads 1 byte per entry, :89 writes 2)."""
    ref = CL["db_table_bytes"].tobytes()
    hdr = [int(x) for x in CL["db_hdr"]]                # n_buckets, size, n_occupied, upper_bound (ref_khc_info order)
    k = int(CL["k"])
    for width in (1, 2):
        t = oracle.Table.wrap(hdr[0], hdr[1], hdr[2], hdr[3], CL["db_flags"].copy(), CL["db_keys_arr"].copy(), CL["db_vals_arr"].copy())
        p = str(tmp_path / ("o%d.db" % width))
        assert oracle.db_write(p, k, k, None, t, spacing_width=width) == 0
        raw = open(p, "rb").read()
        head = 8 + (k - 1) * width
        assert raw[:8] == np.array([k, k], dtype="<u4").tobytes() and raw[8:head] == bytes(head - 8)
        assert raw[head:] == ref
        p2 = str(tmp_path / ("h%d.db" % width))
        hostio.write_db(p2, k, k, None, [hdr[0], hdr[2], hdr[1], hdr[3]], CL["db_flags"], CL["db_keys_arr"], CL["db_vals_arr"],
                        spacing_width=width)
        assert open(p2, "rb").read() == raw
        d = hostio.read_db(p2)
        assert np.array_equal(d["keys"], CL["db_keys_arr"]) and np.array_equal(d["vals"], CL["db_vals_arr"])


# ------------------------------------------------------------------ rows 7, f2: build_parent_map, kseq_read / bseq_read

def test_build_parent_map_reference_vectors(oracle, hostio, IN, tmp_path):
    for nm in ("plain", "comments", "no_root_line", "tight"):
        p = str(tmp_path / (nm + ".dmp"))
        open(p, "wb").write(IN["dmp_" + nm].tobytes())
        child, parent = IN["dmp_%s_child" % nm], IN["dmp_%s_parent" % nm]
        for arr in (oracle.Taxonomy(path=p).parent, hostio.read_nodes_dmp(p)):
            keys = np.nonzero(arr != 0xFFFFFFFF)[0]
            assert np.array_equal(keys, child) and np.array_equal(arr[keys], parent), nm
    assert bool(IN["dmp_too_small_throws"])
    p = str(tmp_path / "one.dmp")
    open(p, "wb").write(b"# nothing\n")
    with pytest.raises(hostio.HostIOError):
        hostio.read_nodes_dmp(p)
    with pytest.raises(ValueError):
        oracle.Taxonomy(path=p)


@pytest.mark.parametrize("block_bytes", [0, 97, 1000])
def test_ingest_reference_vectors(hostio, IN, tmp_path, block_bytes):
    """SeqReader / bseq_read of the C++ host == the reference's kseq_read + bseq_read (klib/kseq.h:177-225,
    kseq_declare.h:106-146) record for record: names (with /1 /2 trimming), comments, joined multi-line sequences, qualities,
    where a broken record ends the stream, pairing against a shorter mate file, and the chunk boundaries."""
    for ci in range(int(IN["n_cases"])):
        f1, f2, chunk = str(IN["case%d_file1" % ci]), str(IN["case%d_file2" % ci]), int(IN["case%d_chunk" % ci])
        paths = []
        for f in (f1, f2):
            if f:
                p = str(tmp_path / ("%d_%s%s" % (ci, f, ".gz" if f.endswith("_gz") else "")))
                open(p, "wb").write(IN["text_" + f].tobytes())
                paths.append(p)
        fields = IN["case%d_fields" % ci].tobytes().split(b"\0")[:-1]
        exp = [tuple(fields[4 * i:4 * i + 4]) for i in range(len(fields) // 4)]
        chunk_of = IN["case%d_chunk_of" % ci]
        recs, n_chunks = hostio.read_fastx(paths[0], paths[1] if len(paths) > 1 else None, chunk_size=chunk, block_bytes=block_bytes)
        assert [tuple(r) for r in recs] == exp, (ci, f1, f2)
        assert n_chunks == (int(chunk_of[-1]) + 1 if chunk_of.size else 0), (ci, f1, f2, chunk)
        assert [len(r[2]) for r in recs] == IN["case%d_lseq" % ci].tolist()


# ------------------------------------------------------------------ live cross-check when the reference build is present

def test_live_reference_crosscheck(oracle):
    R = oracle.ref()
    if R is None or not hasattr(R, "ref_resolve_pairs"):
        pytest.skip("oracle/_ref not built (no reference checkout on this box): the committed goldens above are the pin")
    rng = np.random.default_rng(7)
    for trial in range(30):
        n = int(rng.integers(5, 400))
        ids = np.concatenate([[1], rng.choice(np.arange(2, 100000), size=n - 1, replace=False)]).astype(np.uint32)
        par = np.zeros(n, dtype=np.uint32)
        for i in range(1, n):
            par[i] = ids[int(rng.integers(0, i))]
        h = R.ref_khp_from_pairs(ids.ctypes.data_as(oracle.u32p), par.ctypes.data_as(oracle.u32p), n)
        tax = oracle.Taxonomy(pairs=list(zip(ids.tolist(), par.tolist())))
        for _ in range(200):
            nd = int(rng.integers(1, 9))
            ks = rng.choice(ids, size=min(nd, n), replace=False).astype(np.uint32)
            cs = rng.integers(1, 4, size=ks.size).astype(np.uint32)
            want = R.ref_resolve_pairs(h, ks.ctypes.data_as(oracle.u32p), cs.ctypes.data_as(oracle.u32p), ks.size)
            assert tax.resolve(ks, cs.astype(np.uint16)) == want
            a, b = int(rng.choice(ids)), int(rng.choice(ids))
            assert tax.lca(a, b) == R.ref_lca(h, a, b)
        R.ref_khp_free(h)
