"""GPU tier against vectors produced by the REFERENCE's own code (tests/golden/{resolve,classify}_ref.npz, written by
tests/golden/make_golden_tree.py from the reference's lca / resolve_tree / update_lca_map / kh_get / linear::counter /
formatters compiled out of the checkout).  No oracle in this file: HIP through the C ABI (and the CLI binary) versus frozen
reference outputs."""
import os
import subprocess

import numpy as np
import pytest

import bonsai_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "bonsai_amd", "bin", "bonsai")
ABSENT = 0xFFFFFFFF


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLD, "resolve_ref.npz"))


@pytest.fixture(scope="module")
def CL():
    return np.load(os.path.join(GOLD, "classify_ref.npz"))


def flat_parent(child, parent):
    p = np.full(int(max(child.max(), parent.max())) + 1, ABSENT, dtype=np.uint32)
    p[child] = parent
    p[1] = 0                       # build_parent_map forces the root (util.h:780-781); the golden taxonomy's line is "1 | 1"
    return p


def test_resolve_batch_reference_vectors(gpu_ctx, G):
    """bns_resolve_batch (resolve_wave + lca_dev over the Euler-interval taxonomy) == the reference's resolve_tree on 52 000
    counters over 9 random forests (ids up to 2^27, several trees per forest, > 128 distinct taxa, u16 wrap)."""
    total = 0
    for fi in range(int(G["n_forests"])):
        g = lambda n: G["f%d_%s" % (fi, n)]          # noqa: E731
        gpu_ctx.load_taxonomy(flat_parent(g("child"), g("parent")))
        got = gpu_ctx.resolve(g("keys"), g("counts").astype(np.uint16), g("offs"))
        exp = g("expected")
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, "forest %d: %d mismatches, first at case %d: HIP %d reference %d" % (fi, bad.size, bad[0], got[bad[0]], exp[bad[0]])
        total += exp.size
    assert total >= 50000


def test_lca_through_ties_reference_vectors(gpu_ctx, G):
    """lca_dev == the reference's lca (util.h:634-663): a two-entry counter with equal counts ties, so resolve_tree returns
    lca(a, b) -- unless one is the other's ancestor (then the descendant's path sum is larger and wins)."""
    for fi in range(int(G["n_forests"])):
        g = lambda n: G["f%d_%s" % (fi, n)]          # noqa: E731
        child, parent = g("child"), g("parent")
        a, b, exp = g("lca_a"), g("lca_b"), g("lca")
        ok = (a != b) & (a != 0) & (b != 0) & np.isin(a, child) & np.isin(b, child)
        # keep unrelated pairs only (lca is neither a nor b)
        ok &= (exp != a) & (exp != b)
        a, b, exp = a[ok], b[ok], exp[ok]
        assert a.size > 500
        gpu_ctx.load_taxonomy(flat_parent(child, parent))
        keys = np.stack([a, b], axis=1).reshape(-1)
        counts = np.full(keys.size, 3, dtype=np.uint16)
        offs = np.arange(0, keys.size + 1, 2, dtype=np.uint64)
        got = gpu_ctx.resolve(keys, counts, offs)
        assert np.array_equal(got, exp), fi


def test_encode_reference_streams(gpu_ctx):
    """bns_encode_batch == the k-mer streams made with the REFERENCE's own DNA4 table, rhmask and canonical_representation
    (tests/golden/stream_ref.npz, make_golden_stream.py) -- no oracle in between: phiX (5356 31-mers, reference
    test/encoding.cpp:122) and 30 crafted strings at k = 1, 5, 16, 21, 31, 32, forward and canonical."""
    ST = np.load(os.path.join(GOLD, "stream_ref.npz"))
    strs = [bytes(ST["str_bytes"][int(a):int(b)]) for a, b in zip(ST["str_offs"][:-1], ST["str_offs"][1:])]
    phix = b"".join(l.strip() for l in open(os.path.join(GOLD, "phix.fa"), "rb").read().splitlines()[1:])
    bases, offsets = bonsai_amd.concat_reads(strs + [phix])
    total = 0
    for k in ST["ks"].tolist():
        for canon in (0, 1):
            gpu_ctx.set_encoder(k, None, canonicalize=bool(canon))
            got = gpu_ctx.encode(bases, offsets)
            vals, cnt = ST["s_k%d_c%d" % (k, canon)], ST["n_k%d_c%d" % (k, canon)]
            ends = np.cumsum(cnt)
            for i in range(len(strs)):
                assert np.array_equal(got[i], vals[int(ends[i] - cnt[i]):int(ends[i])]), (k, canon, i)
                total += int(cnt[i])
            if k == 31:
                assert np.array_equal(got[-1], ST["phix_cn31"] if canon else ST["phix_fw31"])
    assert total > 40000


def load_golden_db(ctx, CL, layout):
    ctx.set_encoder(int(CL["k"]), None, canonicalize=True)
    ctx.load_table(int(CL["db_hdr"][0]), CL["db_flags"], CL["db_keys_arr"], CL["db_vals_arr"], layout=layout)
    ctx.load_taxonomy(flat_parent(CL["tax_child"], CL["tax_parent"]))


@pytest.mark.parametrize("layout", [bonsai_amd.LAYOUT_MINBUCKET, bonsai_amd.LAYOUT_BUCKET, bonsai_amd.LAYOUT_KHASH])
@pytest.mark.parametrize("paired", [False, True])
def test_classify_reference_vectors(gpu_ctx, CL, layout, paired):
    """bns_classify_batch on the khash arrays the reference's update_lca_map built == classify_seq's body run with the
    reference's kh_get / linear::counter / resolve_tree: taxon, missing, ambig (u32 wrap included), hit count and the ordered
    hit stream, 2000 reads / 1000 pairs (lower case, N, IUPAC, empty, shorter than k)."""
    load_golden_db(gpu_ctx, CL, layout)
    pre = "p_" if paired else "s_"
    exp = CL[pre + "res"]
    got = gpu_ctx.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired, want_hits=True)
    for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
        bad = np.nonzero(got[f] != exp[:, j])[0]
        assert bad.size == 0, "%s: %d mismatches, first unit %d: HIP %d reference %d" % (f, bad.size, bad[0], got[f][bad[0]], exp[bad[0], j])
    hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
    for u in range(exp.shape[0]):
        assert np.array_equal(got["hits"][u], hits[int(ho[u]):int(ho[u + 1])]), u
    assert int((exp[:, 0] != 0).sum()) > exp.shape[0] // 2


@pytest.mark.parametrize("span", [0, 8, 11, 15])
@pytest.mark.parametrize("paired", [False, True])
def test_classify_reference_vectors_every_minimizer_window(CL, span, paired):
    """The clustered table's minimizer window (k - m = 8, 11, 15, or chosen from the db) changes where a key lives, never what
    a lookup returns: the reference-code expectations hold for each."""
    ctx = bonsai_amd.Context(0)
    try:
        ctx.set_minimizer_span(span)
        load_golden_db(ctx, CL, bonsai_amd.LAYOUT_MINBUCKET)
        m = ctx.table_minimizer()["m"]
        k = int(CL["k"])
        assert m == (k - span if span else k - 8)        # a db of every k-mer fills the wide windows' groups: the narrow one is chosen
        pre = "p_" if paired else "s_"
        exp = CL[pre + "res"]
        got = ctx.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired, want_hits=True)
        for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
            assert np.array_equal(got[f], exp[:, j]), (f, span)
        hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
        for u in range(exp.shape[0]):
            assert np.array_equal(got["hits"][u], hits[int(ho[u]):int(ho[u + 1])]), u
        vals, found = ctx.probe(CL["db_keys"])
        assert found.all() and np.array_equal(vals, CL["db_vals"])
    finally:
        ctx.close()


@pytest.mark.parametrize("layout", [bonsai_amd.LAYOUT_MINBUCKET, bonsai_amd.LAYOUT_BUCKET])
def test_streamed_table_load(CL, layout):
    """bns_load_table with the khash arrays streamed from the host in chunks (what a db too big to sit in HBM next to its table
    gets; forced here, with 16 chunks of 2048 slots) builds the same table: the reference-code expectations hold."""
    ctx = bonsai_amd.Context(0)
    try:
        ctx.debug_set(0x1000 | (11 << 16))                # BNS_DBG_STREAM_LOAD, chunks of 2^11 slots
        load_golden_db(ctx, CL, layout)
        assert ctx.table_stats()["n_keys"] == int(CL["db_keys"].size)
        for paired in (False, True):
            pre = "p_" if paired else "s_"
            exp = CL[pre + "res"]
            got = ctx.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired)
            for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
                assert np.array_equal(got[f], exp[:, j]), (f, paired)
        vals, found = ctx.probe(CL["db_keys"])
        assert found.all() and np.array_equal(vals, CL["db_vals"])
        # the faithful layout probes the arrays themselves: it is uploaded whole whatever the switch says
        ctx.load_table(int(CL["db_hdr"][0]), CL["db_flags"], CL["db_keys_arr"], CL["db_vals_arr"], layout=bonsai_amd.LAYOUT_KHASH)
        vals, found = ctx.probe(CL["db_keys"])
        assert found.all() and np.array_equal(vals, CL["db_vals"])
    finally:
        ctx.close()


@pytest.mark.parametrize("layout", [bonsai_amd.LAYOUT_MINBUCKET, bonsai_amd.LAYOUT_BUCKET, bonsai_amd.LAYOUT_KHASH])
@pytest.mark.parametrize("paired", [False, True])
def test_classify_packed_reference_vectors(gpu_ctx, CL, layout, paired):
    """The packed entry points (bns_pack_reads on the host, bns_classify_batch_packed / _packed_device) against the
    reference-code vectors, no oracle and no ASCII path in between: taxon, missing, ambig, hit count and the ordered hit stream of
    2000 reads / 1000 pairs (lower case, N, IUPAC, empty, shorter than k) -- through the host call and, with the flag words made
    dense by hand, through the device-resident call."""
    torch = pytest.importorskip("torch")
    load_golden_db(gpu_ctx, CL, layout)
    pre = "p_" if paired else "s_"
    bases, offs, exp = CL[pre + "bases"], CL[pre + "offs"], CL[pre + "res"]
    words, bw, bm = bonsai_amd.pack_reads(bases, offs, threads=3)
    assert bw.size > 0                                        # (the vectors do hold invalid bases)
    got = gpu_ctx.classify_packed(words, bw, bm, offs, paired=paired, want_hits=True)
    for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
        assert np.array_equal(got[f], exp[:, j]), (f, paired)
    hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
    for u in range(exp.shape[0]):
        assert np.array_equal(got["hits"][u], hits[int(ho[u]):int(ho[u + 1])]), u
    # device-resident: dense flag words
    dev = torch.device("cuda", 0)
    dense = np.zeros(words.size, dtype=np.uint32); dense[bw.astype(np.int64)] = bm
    d_w = torch.from_numpy(words.view(np.int64)).to(dev); d_m = torch.from_numpy(dense.view(np.int32)).to(dev)
    d_o = torch.from_numpy(offs.astype(np.int64)).to(dev)
    n = offs.size - 1
    nu = n // 2 if paired else n
    out = [torch.zeros(nu, dtype=torch.int32, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    gpu_ctx.classify_packed_device(d_w.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), n, int(offs[-1]), 0, paired, out[0].data_ptr(), out[1].data_ptr(),
                                   out[2].data_ptr(), out[3].data_ptr(), None, None)
    torch.cuda.synchronize()
    for j in range(4):
        assert np.array_equal(out[j].cpu().numpy().view(np.uint32), exp[:, j]), j


@pytest.mark.parametrize("bits", [32, 52])
@pytest.mark.parametrize("buckets", [0, 2000, 2049, 4097, 30011])
@pytest.mark.parametrize("span", [0, 8, 15])
def test_table_geometries(CL, bits, buckets, span):
    """The clustered table at bucket counts that are not powers of two (the bucket index is a multiply-high, so a table can take
    exactly the memory there is), from crowded (2000 buckets: 95 % load, chains and overflow table in use) to sparse, with the
    narrow (32-bit) and the wide (52-bit: hash and m-mer carried through the window minimum as one double) minimizer identity
    and three minimizer windows: one key -> value map, so the reference-code vectors hold for every one of them."""
    ctx = bonsai_amd.Context(0)
    try:
        ctx.set_table_buckets(buckets)
        ctx.set_minimizer_identity(bits)
        ctx.set_minimizer_span(span)
        load_golden_db(ctx, CL, bonsai_amd.LAYOUT_MINBUCKET)
        geo = ctx.table_geometry()
        assert geo["identity_bits"] == bits and (buckets == 0 or geo["buckets"] == buckets)
        assert ctx.table_stats()["n_keys"] == int(CL["db_keys"].size)
        for paired in (False, True):
            pre = "p_" if paired else "s_"
            exp = CL[pre + "res"]
            got = ctx.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired)
            for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
                assert np.array_equal(got[f], exp[:, j]), (f, paired)
        vals, found = ctx.probe(CL["db_keys"])
        assert found.all() and np.array_equal(vals, CL["db_vals"])
        absent = CL["db_keys"] ^ np.uint64(0x15555)
        absent = absent[~np.isin(absent, CL["db_keys"])]
        _, found = ctx.probe(absent)
        assert not found.any()
        if buckets == 2000 and span == 15:
            assert "not in their home bucket" in ctx.table_warning()      # a forced wide window on a crowded table says so
    finally:
        ctx.close()


@pytest.mark.parametrize("span", [0, 11, 15])
def test_crowded_table_overflow_paths(CL, span):
    """The golden db squeezed into half as many slots as khash buckets (93 % load): chains fill, a good share of the keys lives in
    the overflow table, home buckets carry the overflow flag, and classify runs the instantiation with the cooperative overflow
    lookup -- same reference-code answers, every key still found, absent keys still missed."""
    ctx = bonsai_amd.Context(0)
    try:
        ctx.set_minimizer_span(span)
        ctx.set_bucket_slots_log2(14)
        load_golden_db(ctx, CL, bonsai_amd.LAYOUT_MINBUCKET)
        st = ctx.table_stats()
        assert st["n_keys"] == int(CL["db_keys"].size) and st["n_overflow_keys"] * 1000 > st["n_keys"]
        for paired in (False, True):
            pre = "p_" if paired else "s_"
            exp = CL[pre + "res"]
            got = ctx.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired, want_hits=True)
            for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
                assert np.array_equal(got[f], exp[:, j]), (f, paired)
            hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
            for u in range(0, exp.shape[0], 7):
                assert np.array_equal(got["hits"][u], hits[int(ho[u]):int(ho[u + 1])]), u
        vals, found = ctx.probe(CL["db_keys"])
        assert found.all() and np.array_equal(vals, CL["db_vals"])
        absent = CL["db_keys"] ^ np.uint64(0x15555)                       # neighbours in key space, (nearly) none of them keys
        absent = absent[~np.isin(absent, CL["db_keys"])]
        _, found = ctx.probe(absent)
        assert not found.any()
    finally:
        ctx.close()


@pytest.mark.parametrize("bits", [32, 52])
@pytest.mark.parametrize("buckets", [2000, 2700, 4097])
@pytest.mark.parametrize("span", [0, 8, 15])
def test_group_fill_and_arrival_order_fill_agree(CL, bits, buckets, span):
    """The two ways a clustered table is filled -- keys in arrival order, and group by group (whole minimizer groups keep their home
    bucket, largest first; the header's tag bits say which groups left) -- and the two forms of the lookup (tag bits consulted or
    not; the debug bits force either) give the reference-code answers on crowded tables, find every key and miss every absent
    one.  The group-aware fill must not leave more keys outside their home bucket than arrival order does."""
    PLAIN, GROUP, OVC_OFF, OVC_ON = 0x10, 0x20, 0x2000, 0x8000
    spilled = {}
    for dbg in (PLAIN | OVC_ON, PLAIN | OVC_OFF, GROUP | OVC_ON, GROUP | OVC_OFF):
        ctx = bonsai_amd.Context(0)
        try:
            ctx.debug_set(dbg)
            ctx.set_table_buckets(buckets)
            ctx.set_minimizer_identity(bits)
            ctx.set_minimizer_span(span)
            load_golden_db(ctx, CL, bonsai_amd.LAYOUT_MINBUCKET)
            geo = ctx.table_geometry()
            assert geo["group_fill"] == (1 if dbg & GROUP else 0)
            spilled[dbg & (PLAIN | GROUP)] = geo["spilled_keys"] + geo["overflow_keys"]
            assert ctx.table_stats()["n_keys"] == int(CL["db_keys"].size)
            for paired in (False, True):
                pre = "p_" if paired else "s_"
                exp = CL[pre + "res"]
                got = ctx.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired, want_hits=True)
                for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
                    assert np.array_equal(got[f], exp[:, j]), (f, paired, hex(dbg))
                hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
                for u in range(0, exp.shape[0], 11):
                    assert np.array_equal(got["hits"][u], hits[int(ho[u]):int(ho[u + 1])]), (u, hex(dbg))
            vals, found = ctx.probe(CL["db_keys"])
            assert found.all() and np.array_equal(vals, CL["db_vals"])
            absent = CL["db_keys"] ^ np.uint64(0x15555)
            absent = absent[~np.isin(absent, CL["db_keys"])]
            _, found = ctx.probe(absent)
            assert not found.any()
        finally:
            ctx.close()
    assert spilled[GROUP] <= spilled[PLAIN] * 1.02 + 8, spilled


def test_table_fill_setter_reproduces_the_root_choice(CL):
    """bns_set_table_fill: 1 = arrival order, 2 = group by group whatever the loader would choose, 0 = its choice back; entry 6 of
    the geometry + 1 is what a second context feeds it to get the root's table."""
    ctx = bonsai_amd.Context(0)
    try:
        for mode, want in ((2, 1), (1, 0), (0, None)):
            ctx.set_table_fill(mode)
            ctx.set_table_buckets(2600)
            load_golden_db(ctx, CL, bonsai_amd.LAYOUT_MINBUCKET)
            geo = ctx.table_geometry()
            if want is not None:
                assert geo["group_fill"] == want
            else:
                assert geo["group_fill"] == 1                         # (a table this crowded: the loader goes group by group)
            vals, found = ctx.probe(CL["db_keys"])
            assert found.all() and np.array_equal(vals, CL["db_vals"])
            exp = CL["s_res"]
            got = ctx.classify(CL["s_bases"], CL["s_offs"])
            for j, f in enumerate(("taxon", "missing", "ambig", "n_hits")):
                assert np.array_equal(got[f], exp[:, j]), (f, mode)
        with pytest.raises(bonsai_amd.BonsaiAmdError):
            ctx.set_table_fill(3)
    finally:
        ctx.close()


def test_minimizer_window_follows_the_db(CL):
    """Nine keys in ten marked deleted (what a db of window minimizers looks like: sparse groups) -> the widest window;
    lookups of the kept keys still hit, the deleted ones miss."""
    ctx = bonsai_amd.Context(0)
    try:
        flags = CL["db_flags"].copy()
        karr, varr = CL["db_keys_arr"], CL["db_vals_arr"]
        nb = int(CL["db_hdr"][0])
        idx = np.arange(nb)
        present = ((flags[idx >> 4] >> ((idx & 15) << 1)) & 3) == 0
        pres_idx = idx[present]
        drop = pres_idx[np.arange(pres_idx.size) % 10 != 0]
        np.bitwise_or.at(flags, drop >> 4, (np.uint32(1) << ((drop & 15) << 1).astype(np.uint32)))      # bit 0 of the pair: deleted
        keep = pres_idx[np.arange(pres_idx.size) % 10 == 0]
        ctx.set_encoder(int(CL["k"]), None, canonicalize=True)
        ctx.load_table(nb, flags, karr, varr, layout=bonsai_amd.LAYOUT_MINBUCKET)
        info = ctx.table_minimizer()
        assert info["m"] == int(CL["k"]) - 15 and ctx.table_stats()["n_keys"] == keep.size
        vals, found = ctx.probe(karr[keep])
        assert found.all() and np.array_equal(vals, varr[keep])
        _, found = ctx.probe(karr[drop])
        assert not found.any()
        with pytest.raises(bonsai_amd.BonsaiAmdError):
            ctx.set_minimizer_span(9)
    finally:
        ctx.close()


@pytest.fixture(scope="module")
def cli_files(CL, tmp_path_factory):
    d = tmp_path_factory.mktemp("refcli")
    k = int(CL["k"])
    db = str(d / "ref.db")
    with open(db, "wb") as f:                      # database.h:33-56 header (k, w, k-1 one-byte spacing entries) + the reference's
        f.write(np.array([k, k], dtype="<u4").tobytes() + bytes(k - 1) + CL["db_table_bytes"].tobytes())   # own table bytes
    nodes = str(d / "nodes.dmp")
    with open(nodes, "w") as f:
        for c, p in zip(CL["tax_child"].tolist(), CL["tax_parent"].tolist()):
            f.write("%d\t|\t%d\t|\tno rank\t|\t\t|\n" % (c, p))

    def qual(u, n):
        return bytes((33 + (i * 7 + u) % 40) for i in range(n))

    out = {"db": db, "nodes": nodes, "qual": qual}
    sb, so = CL["s_bases"], CL["s_offs"]
    fa = str(d / "s.fa")
    with open(fa, "wb") as f:                      # FASTA with a FASTQ record (quality string of make_golden_tree.py) every 20th
        for u in range(so.size - 1):
            s = sb[int(so[u]):int(so[u + 1])].tobytes()
            if u % 20 == 0:
                f.write(b"@r%d\n%s\n+\n%s\n" % (u, s, qual(u, len(s))))
            else:
                f.write(b">r%d\n%s\n" % (u, s))
    out["s"] = fa
    pb, po = CL["p_bases"], CL["p_offs"]
    p1, p2 = str(d / "p1.fa"), str(d / "p2.fa")
    with open(p1, "wb") as f1, open(p2, "wb") as f2:
        for u in range((po.size - 1) // 2):
            s1 = pb[int(po[2 * u]):int(po[2 * u + 1])].tobytes()
            s2 = pb[int(po[2 * u + 1]):int(po[2 * u + 2])].tobytes()
            if u % 20 == 0:
                f1.write(b"@r%d\n%s\n+\n%s\n" % (u, s1, qual(u, len(s1))))
            else:
                f1.write(b">r%d\n%s\n" % (u, s1))
            f2.write(b">r%d_m\n%s\n" % (u, s2))
    out["p1"], out["p2"] = p1, p2
    return out


def run_cli(args):
    assert os.path.exists(BIN), "bonsai CLI not built (run __graft_entry__.build())"
    p = subprocess.run([BIN, "classify"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


@pytest.mark.parametrize("paired", [False, True])
def test_cli_kraken_lines_reference(CL, cli_files, paired):
    """`bonsai classify -a` stdout == the reference's append_kraken_classification lines for the frozen classifications; without
    -a only the classified units' lines (classifier.h:239)."""
    pre = "p_" if paired else "s_"
    lines, lo = CL[pre + "lines"].tobytes(), CL[pre + "lines_offs"]
    inputs = [cli_files["p1"], cli_files["p2"]] if paired else [cli_files["s"]]
    assert run_cli(["-a", cli_files["db"], cli_files["nodes"]] + inputs) == lines
    taxon = CL[pre + "res"][:, 0]
    only = b"".join(lines[int(lo[u]):int(lo[u + 1])] for u in range(taxon.size) if taxon[u])
    assert run_cli(["-c", "20000", cli_files["db"], cli_files["nodes"]] + inputs) == only


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("flags,verbose", [(["-f", "-K"], 0), (["-f"], 1), (["-f", "-k"], 1)])
def test_cli_fastq_mode_reference(CL, cli_files, paired, flags, verbose):
    """FASTQ-comment mode through the CLI on the GPU (classifier.h:72-108; -f alone keeps the default -k, so the comment holds
    the whole Kraken record): every record == the C++ formatter's (itself pinned to the reference's bytes in the CPU tier), and
    the records of every tenth unit == the reference's append_fastq_classification bytes directly."""
    from bonsai_amd import hostio
    pre = "p_" if paired else "s_"
    bases, offs, res = CL[pre + "bases"], CL[pre + "offs"], CL[pre + "res"]
    hits, ho = CL[pre + "hits"], CL[pre + "hoffs"]
    fq, fo = CL[pre + "fq"].tobytes(), CL[pre + "fq_offs"]
    inc = 2 if paired else 1
    exp, frozen = [], []
    for u in range(res.shape[0]):
        s1 = bases[int(offs[inc * u]):int(offs[inc * u + 1])].tobytes()
        s2 = bases[int(offs[inc * u + 1]):int(offs[inc * u + 2])].tobytes() if paired else None
        q1 = cli_files["qual"](u, len(s1)) if u % 20 == 0 else None
        h = hits[int(ho[u]):int(ho[u + 1])]
        rec = hostio.fastq_record((b"r%d" % u, s1, q1), (b"r%d_m" % u, s2, None) if paired else None,
                                  int(res[u, 0]), int(res[u, 1]), int(res[u, 2]), h, verbose)
        exp.append(rec)
        if u % 10 == 0:
            j = 2 * (u // 10) + verbose
            assert rec == fq[int(fo[j]):int(fo[j + 1])], u
            frozen.append(rec)
    inputs = [cli_files["p1"], cli_files["p2"]] if paired else [cli_files["s"]]
    got = run_cli(["-a"] + flags + [cli_files["db"], cli_files["nodes"]] + inputs)
    assert got == b"".join(exp)
    assert len(frozen) == res.shape[0] // 10


def test_cli_no_output_mode(cli_files):
    """-K without -f: neither format selected, nothing is printed (the switch at classifier.h:240-245 has no such case)."""
    assert run_cli(["-a", "-K", cli_files["db"], cli_files["nodes"], cli_files["s"]]) == b""


@pytest.mark.parametrize("paired", [False, True])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_cli_multi_context_sharding(CL, cli_files, paired, devices):
    """`bonsai classify -g 0,0`: one context per listed device (here the same GPU twice / three times -- the db is replicated by
    bns_load_table_multi, every chunk's units are split into contiguous ranges, one per context, and the results are stitched
    back in input order, run strings included).  Output must be byte-identical to the frozen reference lines, with chunks small
    enough that many chunks with ragged splits occur."""
    pre = "p_" if paired else "s_"
    lines = CL[pre + "lines"].tobytes()
    inputs = [cli_files["p1"], cli_files["p2"]] if paired else [cli_files["s"]]
    assert run_cli(["-a", "-g", devices, cli_files["db"], cli_files["nodes"]] + inputs) == lines
    assert run_cli(["-a", "-g", devices, "-c", "7000", cli_files["db"], cli_files["nodes"]] + inputs) == lines
    one = run_cli(["-a", "-f", "-g", "0", cli_files["db"], cli_files["nodes"]] + inputs)
    assert run_cli(["-a", "-f", "-g", devices, "-c", "30000", cli_files["db"], cli_files["nodes"]] + inputs) == one


def test_cli_device_list_errors(cli_files):
    for bad in ("0-", "a", "3-1", "0,,1", "99"):
        p = subprocess.run([BIN, "classify", "-g", bad, cli_files["db"], cli_files["nodes"], cli_files["s"]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode != 0 and b"[E]" in p.stderr, bad


@pytest.mark.parametrize("streamed", [False, True])
def test_load_table_multi_python(gpu_ctx, CL, streamed):
    """bns_load_table_multi through ctypes: three contexts on device 0, all classify like the single-context load -- replicated
    array by array, or (what a db too big for that gets; forced here) every context streaming the host buffers into its own table."""
    import ctypes as C
    a, b, c3 = bonsai_amd.Context(0), bonsai_amd.Context(0), bonsai_amd.Context(0)
    try:
        if streamed:
            a.debug_set(0x1000 | (12 << 16))              # BNS_DBG_STREAM_LOAD on the root: the others follow it
        for c in (a, b, c3):
            c.set_encoder(int(CL["k"]), None, canonicalize=True)
        arr = (C.c_void_p * 3)(a.h, b.h, c3.h)
        f = np.ascontiguousarray(CL["db_flags"]); k = np.ascontiguousarray(CL["db_keys_arr"]); v = np.ascontiguousarray(CL["db_vals_arr"])
        rc = a.L.bns_load_table_multi(arr, 3, int(CL["db_hdr"][0]), f.ctypes.data_as(C.POINTER(C.c_uint32)), k.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      v.ctypes.data_as(C.POINTER(C.c_uint32)), bonsai_amd.LAYOUT_MINBUCKET)
        assert rc == 0, a.L.bns_last_error(a.h)
        par = flat_parent(CL["tax_child"], CL["tax_parent"])
        for c in (a, b, c3):
            c.load_taxonomy(par)
            got = c.classify(CL["s_bases"], CL["s_offs"])
            assert np.array_equal(got["taxon"], CL["s_res"][:, 0]) and np.array_equal(got["missing"], CL["s_res"][:, 1])
        assert a.table_stats()["main_bytes"] == b.table_stats()["main_bytes"] == c3.table_stats()["main_bytes"]
        shape = lambda c: {k: v for k, v in c.table_geometry().items() if k in ("buckets", "m", "identity_bits", "span", "group_fill")}   # noqa: E731
        assert shape(a) == shape(b) == shape(c3)             # one size, one minimizer window, one identity for all
    finally:
        a.close(); b.close(); c3.close()


@pytest.mark.parametrize("layout", [bonsai_amd.LAYOUT_MINBUCKET, bonsai_amd.LAYOUT_KHASH])
def test_load_table_multi_rccl_one_rank(CL, layout):
    """The RCCL leg of bns_load_table_multi, executed for real on the one GPU there is: a ONE-rank communicator
    (ncclCommInitAll(n = 1)) and a grouped ncclBroadcast of each of the three khash arrays through the dlopen()ed librccl
    (debug switch BNS_DBG_FORCE_RCCL; without it a single context is a plain bns_load_table).  Proves the entry points' types
    (taken from <rccl/rccl.h>), ncclUint8 and the in-place root broadcast; the table must then classify the reference vectors."""
    import ctypes as C
    a = bonsai_amd.Context(0)
    try:
        a.set_encoder(int(CL["k"]), None, canonicalize=True)
        a.debug_set(0x800)
        arr = (C.c_void_p * 1)(a.h)
        f = np.ascontiguousarray(CL["db_flags"]); k = np.ascontiguousarray(CL["db_keys_arr"]); v = np.ascontiguousarray(CL["db_vals_arr"])
        rc = a.L.bns_load_table_multi(arr, 1, int(CL["db_hdr"][0]), f.ctypes.data_as(C.POINTER(C.c_uint32)), k.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      v.ctypes.data_as(C.POINTER(C.c_uint32)), layout)
        assert rc == 0, a.L.bns_last_error(a.h)
        maps = open("/proc/self/maps").read()
        assert "librccl" in maps, "the RCCL path did not run: librccl is not mapped"
        a.load_taxonomy(flat_parent(CL["tax_child"], CL["tax_parent"]))
        for paired in (False, True):
            pre = "p_" if paired else "s_"
            got = a.classify(CL[pre + "bases"], CL[pre + "offs"], paired=paired)
            for j, fld in enumerate(("taxon", "missing", "ambig", "n_hits")):
                assert np.array_equal(got[fld], CL[pre + "res"][:, j]), (fld, paired)
        vals, found = a.probe(CL["db_keys"])
        assert found.all() and np.array_equal(vals, CL["db_vals"])
    finally:
        a.close()
