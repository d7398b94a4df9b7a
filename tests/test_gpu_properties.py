"""Size-independent properties of the classify path at (near) benchmark scale -- too big for the oracle, so the checks are
relations that must hold whatever the data: the three table layouts agree, a read and its reverse complement classify
alike (canonical db), every k-mer is a hit, a miss or ambiguous, runs are reproducible, and shards compose."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, L = 31, 150


@pytest.fixture(scope="module")
def big():
    torch = pytest.importorskip("torch")
    sys.path.insert(0, ROOT)
    import bench
    import bonsai_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = bonsai_amd.Context(0)
    NG, G, LG = 256, 1 << 18, 27
    parent, leaves = bench.make_taxonomy(NG)
    ctx.set_encoder(K, None, canonicalize=True)
    ctx.load_taxonomy(parent)
    nb = 1 << LG
    flags = torch.empty(nb >> 4, dtype=torch.int32, device=dev)
    keys = torch.empty(nb, dtype=torch.int64, device=dev)
    vals = torch.empty(nb, dtype=torch.int32, device=dev)
    pool = bench.make_pool(NG, G, dev, seed=7)
    pa = bench.codes_to_ascii(pool)
    goff = torch.arange(NG + 1, device=dev, dtype=torch.int64) * G
    taxid = torch.from_numpy(leaves.astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    ctx.build_table_device(pa.data_ptr(), goff.data_ptr(), NG, NG * G, taxid.data_ptr(), nb, flags.data_ptr(), keys.data_ptr(),
                           vals.data_ptr(), None)
    n = 2_000_000
    reads = bench.gen_reads(pool, n, L, NG, G, dev, seed=99, sub_rate=0.01, n_rate=0.001)
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
    torch.cuda.synchronize()          # the library works on the context's own stream: torch's generator kernels must be done
    yield {"torch": torch, "ctx": ctx, "dev": dev, "tab": (nb, flags, keys, vals), "reads": reads, "offsets": offsets, "n": n,
           "bonsai_amd": bonsai_amd}
    ctx.close()


def run(b, layout, reads=None, offsets=None, n=None, paired=False):
    torch, ctx = b["torch"], b["ctx"]
    reads = b["reads"] if reads is None else reads
    offsets = b["offsets"] if offsets is None else offsets
    n = b["n"] if n is None else n
    nb, flags, keys, vals = b["tab"]
    if b.get("loaded") != layout:
        ctx.load_table_device(nb, flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), layout, None)
        b["loaded"] = layout
    nu = n // 2 if paired else n
    out = [torch.zeros(nu, dtype=torch.int32, device=b["dev"]) for _ in range(4)]
    torch.cuda.synchronize()          # (inputs made by torch ops on its stream; the call below runs on the context's stream)
    ctx.classify_device(reads.data_ptr(), offsets.data_ptr(), n, int(offsets[n].item()), L, paired, out[0].data_ptr(), out[1].data_ptr(),
                        out[2].data_ptr(), out[3].data_ptr(), None, None)
    torch.cuda.synchronize()
    return out                                               # taxon, missing, ambig, n_hits


def test_layouts_agree_and_runs_repeat(big):
    torch, A = big["torch"], big["bonsai_amd"]
    ref = run(big, A.LAYOUT_MINBUCKET)
    again = run(big, A.LAYOUT_MINBUCKET)
    assert all(torch.equal(x, y) for x, y in zip(ref, again))
    for layout in (A.LAYOUT_BUCKET, A.LAYOUT_KHASH):
        got = run(big, layout)
        assert all(torch.equal(x, y) for x, y in zip(ref, got)), layout
    assert (ref[0] != 0).float().mean().item() > 0.99
    assert torch.unique(ref[0]).numel() > 100


def test_batch_size_does_not_matter(big):
    """a prefix of the batch classified on its own gives the prefix of the results: the wavefronts' claim size (63 reads for a full
    batch, an even share of a small one -- down to 4) and the launch width change with the batch, the answers must not"""
    torch, A = big["torch"], big["bonsai_amd"]
    ref = run(big, A.LAYOUT_MINBUCKET)
    for n in (1, 2, 5, 63, 64, 1000, 40_000, 111_848, 300_000, 516_096, 516_097, 1_000_001):
        got = run(big, A.LAYOUT_MINBUCKET, n=n)
        assert all(torch.equal(x[:n], y) for x, y in zip(ref, got)), n
    for n in (2, 126, 111_848, 600_000):                     # pairs: 31 per claim at most
        full = run(big, A.LAYOUT_MINBUCKET, paired=True)
        got = run(big, A.LAYOUT_MINBUCKET, n=n, paired=True)
        assert all(torch.equal(x[:n // 2], y) for x, y in zip(full, got)), n


def test_minimizer_windows_agree(big):
    """the clustered table with its minimizer window fixed at 8, 11 and 15 and chosen by the loader: four placements of the same
    5.7e7 keys, the same answers for 2 M reads"""
    torch, A, ctx = big["torch"], big["bonsai_amd"], big["ctx"]
    ref = run(big, A.LAYOUT_MINBUCKET)
    auto_m = ctx.table_minimizer()["m"]
    seen = {auto_m}
    try:
        for span in (8, 11, 15):
            ctx.set_minimizer_span(span)
            big["loaded"] = None
            got = run(big, A.LAYOUT_MINBUCKET)
            assert ctx.table_minimizer()["m"] == K - span
            seen.add(K - span)
            assert all(torch.equal(x, y) for x, y in zip(ref, got)), span
    finally:
        ctx.set_minimizer_span(0)
        big["loaded"] = None
    assert seen == {K - 8, K - 11, K - 15}                  # (the loader's own choice is one of the three)


def test_every_kmer_is_accounted_for(big):
    A = big["bonsai_amd"]
    taxon, missing, ambig, n_hits = run(big, A.LAYOUT_MINBUCKET)
    total = (missing.long() + ambig.long() + n_hits.long())
    assert bool((total == L - K + 1).all())                  # classifier.h:232: ambig = l - c + 1 - hits - missing
    # a read without N has no ambiguous k-mer; one N costs at most k of them
    has_n = (big["reads"].view(big["n"], L) == ord("N")).any(dim=1)
    assert bool((ambig[~has_n] == 0).all())
    assert bool((ambig[has_n] >= 1).all()) and int(ambig.max().item()) <= L - K + 1


def test_reverse_complement_classifies_alike(big):
    torch, A = big["torch"], big["bonsai_amd"]
    fwd = run(big, A.LAYOUT_MINBUCKET)
    r = big["reads"].view(big["n"], L)
    lut = torch.arange(256, dtype=torch.uint8, device=big["dev"])
    for a, c in zip(b"ACGTacgt", b"TGCAtgca"):
        lut[a] = c
    rc = lut[r.long()].flip(1).contiguous().view(-1)
    rev = run(big, A.LAYOUT_MINBUCKET, reads=rc)
    # canonical k-mers: the same keys in the opposite order -> same counts, and resolve_tree does not depend on hit order
    assert all(torch.equal(x, y) for x, y in zip(fwd, rev))


def test_shards_compose_and_pairs_vote_once(big):
    torch, A = big["torch"], big["bonsai_amd"]
    whole = run(big, A.LAYOUT_MINBUCKET)
    n = big["n"]
    cut = (n // 3) & ~1
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        sub = big["reads"][lo * L:hi * L].contiguous()
        off = torch.arange(hi - lo + 1, device=big["dev"], dtype=torch.int64) * L
        parts.append(run(big, A.LAYOUT_MINBUCKET, reads=sub, offsets=off, n=hi - lo))
    for i in range(4):
        assert torch.equal(whole[i], torch.cat([parts[0][i], parts[1][i]]))
    # mates: hits and misses add up over the pair (classifier.h:233-236); the pair gets ONE taxon
    pt, pm, pa, ph = run(big, A.LAYOUT_MINBUCKET, paired=True)
    assert torch.equal(ph, whole[3][0::2] + whole[3][1::2])
    assert torch.equal(pm, whole[1][0::2] + whole[1][1::2])
    both_same = whole[0][0::2] == whole[0][1::2]
    assert bool((pt[both_same & (whole[0][0::2] != 0)] != 0).all())


def test_nonuniform_genomes_keep_the_clustered_table_sane():
    """Genomes with tandem repeats, homopolymer / micro-satellite tracts and mobile elements copied across unrelated genomes
    (bench.add_repeats) -- what skews minimizer buckets and what iid-uniform genomes lack: oversized minimizer groups must end
    up bounded (chains of 4, then the overflow table), not degrade the table as a whole.  Regression thresholds on the loader's own
    counters (measured: 1.4 % of the keys outside their home bucket, 0.03 % in the overflow table; uniform genomes 0.9 % / 0.01 %), and
    the answers must equal the faithful layout's."""
    torch = pytest.importorskip("torch")
    sys.path.insert(0, ROOT)
    import bench
    import bonsai_amd as A
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = A.Context(0)
    try:
        NG, G, LG = 128, 1 << 19, 27
        parent, leaves = bench.make_taxonomy(NG)
        ctx.set_encoder(K, None, canonicalize=True)
        ctx.load_taxonomy(parent)
        nb = 1 << LG
        flags = torch.empty(nb >> 4, dtype=torch.int32, device=dev)
        keys = torch.empty(nb, dtype=torch.int64, device=dev)
        vals = torch.empty(nb, dtype=torch.int32, device=dev)
        pool = bench.make_pool(NG, G, dev, seed=7)
        info = bench.add_repeats(pool, NG, G, seed=11)
        assert info["mobile_element_copies"] >= NG and info["homopolymer_tracts"] > 0
        pa = bench.codes_to_ascii(pool)
        goff = torch.arange(NG + 1, device=dev, dtype=torch.int64) * G
        taxid = torch.from_numpy(leaves.astype(np.int32)).to(dev)
        torch.cuda.synchronize()
        hdr = ctx.build_table_device(pa.data_ptr(), goff.data_ptr(), NG, NG * G, taxid.data_ptr(), nb, flags.data_ptr(), keys.data_ptr(),
                                     vals.data_ptr(), None)
        n = 1_000_000
        reads = bench.gen_reads(pool, n, L, NG, G, dev, seed=5, sub_rate=0.01, n_rate=0.001)
        offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * L
        torch.cuda.synchronize()
        res = {}
        for layout in (A.LAYOUT_MINBUCKET, A.LAYOUT_KHASH):
            ctx.load_table_device(nb, flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), layout, None)
            out = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(4)]
            torch.cuda.synchronize()
            ctx.classify_device(reads.data_ptr(), offsets.data_ptr(), n, n * L, L, False, out[0].data_ptr(), out[1].data_ptr(),
                                out[2].data_ptr(), out[3].data_ptr(), None, None)
            torch.cuda.synchronize()
            res[layout] = out
            if layout == A.LAYOUT_MINBUCKET:
                geo, st = ctx.table_geometry(), ctx.table_stats()
        assert all(torch.equal(x, y) for x, y in zip(res[A.LAYOUT_MINBUCKET], res[A.LAYOUT_KHASH]))
        n_keys = int(hdr[2])
        assert st["n_keys"] == n_keys
        assert geo["spilled_keys"] < 0.06 * n_keys, geo          # every-k-mer db at the default load: 3.9 % uniform
        assert st["n_overflow_keys"] < 0.004 * n_keys, st
    finally:
        ctx.close()
