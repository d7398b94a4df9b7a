#!/usr/bin/env python
"""Generate golden vectors for the taxonomy half of the hot path from the REFERENCE's own code: oracle/_ref/libbns_ref.so
holds lca (util.h:634-663), resolve_tree (util.h:831-869), build_parent_map (util.h:766-785), update_lca_map
(feature_min.h:205-228), khash_write_impl (util.h:279-294), reverse_complement / canonical_representation
(kmerutil.h:83-90,137-140), the Kraken / FASTQ formatters (classifier.h:30-129) and kseq_read / bseq_read
(klib/kseq.h, kseq_declare.h) compiled from the reference checkout (oracle/ref_extract.py, oracle/ref_harness.cpp).

Run in the build container only:   python tests/golden/make_golden_tree.py
Outputs (data only, committed):
  tests/golden/resolve_ref.npz   random forests; >= 50k counters -> resolve_tree; lca pairs; revcomp / canonical
  tests/golden/classify_ref.npz  SURVEY 8c items (2)(3)(6): a 6-genome db built by update_lca_map (sorted key -> taxid pairs),
                                 2000 single + 1000 paired reads -> taxon / missing / ambig / hits, Kraken lines, FASTQ records,
                                 the khash table section of the bns.db as khash_write_impl writes it
  tests/golden/ingest_ref.npz    crafted FASTA/FASTQ texts -> the records kseq_read / bseq_read return; nodes.dmp texts ->
                                 build_parent_map's (child, parent) pairs
The inputs stored next to the expectations are all a checker needs; no reference code is involved when the tests run.
"""
import ctypes as C
import gzip
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import synth  # noqa: E402


def p32(a):
    return a.ctypes.data_as(O.u32p)


def p64(a):
    return a.ctypes.data_as(O.u64p)


# ------------------------------------------------------------------ forests

def random_forest(rng, n, id_hi, shape):
    """(child, parent) pairs of a forest closed under parent(): node 1 is always present with parent 0 (build_parent_map
    forces that, util.h:780-781); a few other roots have parent 0 too.  shape: 'bushy' | 'chain' | 'mixed'."""
    ids = rng.choice(np.arange(2, id_hi, dtype=np.int64), size=n - 1, replace=False)
    ids = np.concatenate([[1], ids]).astype(np.uint32)
    parent = np.zeros(n, dtype=np.uint32)
    for i in range(1, n):
        if shape == "chain":
            j = i - 1 if rng.random() < 0.9 else int(rng.integers(0, i))
        elif shape == "bushy":
            j = int(rng.integers(0, max(1, i // 3 + 1)))
        else:
            j = int(rng.integers(max(0, i - 6), i)) if rng.random() < 0.6 else int(rng.integers(0, i))
        parent[i] = ids[j]
        if rng.random() < 0.004:
            parent[i] = 0                                      # extra root (a second tree in the forest)
    perm = rng.permutation(n)                                  # file order is not tree order
    return ids[perm].copy(), parent[perm].copy()


def chain_of(par, t):
    out = []
    while t:
        out.append(t)
        t = par[t]
    return out


def make_cases(rng, child, parent, n_cases):
    """Counters as (keys in insertion order, counts).  Biased towards relatives (ancestors / siblings share root paths) and
    small equal counts so that 2-, 3- and many-way ties and nested winners are common."""
    par = dict(zip(child.tolist(), parent.tolist()))
    kids = {}
    for c, p in par.items():
        kids.setdefault(p, []).append(c)
    nodes = child
    keys_all, counts_all, offs = [], [], [0]
    for _ in range(n_cases):
        r = rng.random()
        nd = 0 if r < 0.01 else int(rng.integers(1, 4)) if r < 0.5 else int(rng.integers(2, 13)) if r < 0.97 \
            else int(rng.integers(60, 200))
        nd = min(nd, nodes.size)
        ks = []
        seen = set()
        while len(ks) < nd:
            m = rng.random()
            if ks and m < 0.35:                                # an ancestor of an earlier key
                ch = chain_of(par, ks[int(rng.integers(len(ks)))])
                t = ch[int(rng.integers(len(ch)))]
            elif ks and m < 0.6:                               # a sibling / cousin
                p = par[ks[int(rng.integers(len(ks)))]]
                sib = kids.get(p, [])
                t = sib[int(rng.integers(len(sib)))] if sib else int(nodes[int(rng.integers(nodes.size))])
            else:
                t = int(nodes[int(rng.integers(nodes.size))])
            if t not in seen:
                seen.add(t); ks.append(t)
        m = rng.random()
        if m < 0.45:
            cs = np.full(nd, int(rng.integers(1, 4)))          # all equal: ties everywhere
        elif m < 0.85:
            cs = rng.integers(1, 6, size=nd)
        elif m < 0.97:
            cs = rng.integers(1, 121, size=nd)
        else:
            cs = rng.integers(1, 3, size=nd)
            if nd:
                cs[int(rng.integers(nd))] = int(rng.choice([65535, 65536, 65537, 65536 + 120, 40000]))   # u16 range and wrap
        keys_all.append(np.array(ks, dtype=np.uint32)); counts_all.append(np.asarray(cs, dtype=np.uint32))
        offs.append(offs[-1] + nd)
    return (np.concatenate(keys_all) if keys_all else np.zeros(0, np.uint32),
            np.concatenate(counts_all) if counts_all else np.zeros(0, np.uint32), np.array(offs, dtype=np.uint64))


def gen_resolve(R, rng):
    out = {}
    specs = [(12, 40, "mixed", 2000), (16, 1 << 20, "bushy", 3000), (60, 5000, "chain", 4000), (300, 1 << 22, "mixed", 8000),
             (300, 100000, "bushy", 6000), (1000, 3000000, "mixed", 8000), (3000, 3000000, "bushy", 8000),
             (3000, 1 << 27, "chain", 5000), (20000, 3000000, "mixed", 8000)]
    total = 0
    for fi, (n, id_hi, shape, n_cases) in enumerate(specs):
        child, parent = random_forest(rng, n, id_hi, shape)
        h = R.ref_khp_from_pairs(p32(child), p32(parent), child.size)
        keys, counts, offs = make_cases(rng, child, parent, n_cases)
        exp = np.zeros(n_cases, dtype=np.uint32)
        for i in range(n_cases):
            a, b = int(offs[i]), int(offs[i + 1])
            kk = np.ascontiguousarray(keys[a:b]); cc = np.ascontiguousarray(counts[a:b])
            exp[i] = R.ref_resolve_pairs(h, p32(kk), p32(cc), b - a)
        # lca pairs: relatives, unrelated, identical, 0, ids that are not keys, (tax_t)-1
        m = 3000
        a = child[rng.integers(0, child.size, size=m)].copy()
        b = child[rng.integers(0, child.size, size=m)].copy()
        par = dict(zip(child.tolist(), parent.tolist()))
        for i in range(0, m, 3):                                # ancestor pairs
            ch = chain_of(par, int(a[i])); b[i] = ch[int(rng.integers(len(ch)))]
        b[1::50] = a[1::50]
        a[2::97] = 0
        b[5::101] = 0
        absent = np.setdiff1d(rng.integers(2, id_hi, size=64).astype(np.uint32), child)
        if absent.size:
            a[7::113] = absent[0]
            b[11::127] = absent[-1]
        a[13::211] = 0xFFFFFFFF
        lo = np.zeros(m, dtype=np.uint32)
        R.ref_lca_batch(h, p32(a), p32(b), m, p32(lo))
        R.ref_khp_free(h)
        out.update({"f%d_child" % fi: child, "f%d_parent" % fi: parent, "f%d_keys" % fi: keys, "f%d_counts" % fi: counts,
                    "f%d_offs" % fi: offs, "f%d_expected" % fi: exp, "f%d_lca_a" % fi: a, "f%d_lca_b" % fi: b, "f%d_lca" % fi: lo})
        total += n_cases
    out["n_forests"] = np.array(len(specs))
    # reverse_complement / canonical_representation
    ks = rng.integers(1, 33, size=4000).astype(np.uint32)
    kmers = np.array([int(rng.integers(0, 1 << 62)) * 4 + int(rng.integers(0, 4)) for _ in range(4000)], dtype=np.uint64)
    kmers = np.array([int(x) & ((1 << (2 * int(k))) - 1) for x, k in zip(kmers, ks)], dtype=np.uint64)
    out["rc_k"] = ks; out["rc_in"] = kmers
    out["rc_out"] = np.array([R.ref_revcomp(int(x), int(k)) for x, k in zip(kmers, ks)], dtype=np.uint64)
    out["canon_out"] = np.array([R.ref_canonical(int(x), int(k)) for x, k in zip(kmers, ks)], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "resolve_ref.npz"), **out)
    print("resolve_ref.npz: %d forests, %d counters, %d lca pairs" % (len(specs), total, 3000 * len(specs)))


# ------------------------------------------------------------------ classify through the reference's pieces

def kmers_of(seq, k, gaps=None):
    """k-mer stream of one read; the encoder is the oracle's (pinned by the reference's phiX vector, rows 1-2)."""
    return np.ascontiguousarray(O.encode(seq, k, gaps=gaps, canon=gaps is None, spaced_intended=gaps is not None), dtype=np.uint64)


def gen_classify(R, rng):
    k = 31
    genomes = synth.make_genomes(rng, genome_len=4000, seg=400)
    ch = np.array([c for c, _ in synth.TAX_PAIRS], dtype=np.uint32)
    pa = np.array([p for _, p in synth.TAX_PAIRS], dtype=np.uint32)
    with tempfile.TemporaryDirectory() as td:
        nd = os.path.join(td, "nodes.dmp")
        synth.write_nodes_dmp(nd)
        tax = R.ref_build_parent_map(nd.encode())              # the reference's own loader
        db = R.ref_khc_new()
        for leaf, g in genomes.items():                         # lca_map: one khash set per genome, update_lca_map each
            km = np.unique(kmers_of(g.tobytes(), k))
            R.ref_update_lca_map(db, tax, p64(km), km.size, leaf)
        hdr = np.zeros(4, dtype=np.uint64)
        f = O.u32p(); kk = O.u64p(); vv = O.u32p()
        dbf = os.path.join(td, "table.bin")
        nw = R.ref_khc_write(db, dbf.encode(), 0)               # zeroes absent slots first (util.h:282-284)
        R.ref_khc_info(db, p64(hdr), C.byref(f), C.byref(kk), C.byref(vv))
        nb = int(hdr[0]); fs = 1 if nb < 16 else nb >> 4
        flags = np.ctypeslib.as_array(f, shape=(fs,)).copy()
        keys = np.ctypeslib.as_array(kk, shape=(nb,)).copy()
        vals = np.ctypeslib.as_array(vv, shape=(nb,)).copy()
        table_bytes = np.frombuffer(open(dbf, "rb").read(), dtype=np.uint8).copy()
        assert nw == table_bytes.size
    idx = np.arange(nb)
    present = ((flags[idx >> 4] >> ((idx & 15) << 1)) & 3) == 0
    order = np.argsort(keys[present])
    sk, sv = keys[present][order], vals[present][order]

    reads = synth.simulate_reads(rng, genomes, 1400, sub_rate=0.01, n_rate=0.002, random_frac=0.05, lower_rate=0.01)
    reads += synth.simulate_reads(rng, genomes, 500, var_len=True, sub_rate=0.02, n_rate=0.004)
    reads += [np.frombuffer(s, dtype=np.uint8) for s in
              (b"", b"ACGT", b"A" * 30, b"A" * 31, b"N" * 150, b"ACGTN" * 30, genomes[1001][:31].tobytes(),
               genomes[1001][:64].tobytes() + b"N" + genomes[2001][100:164].tobytes(),
               genomes[1003][:200].tobytes().lower(), genomes[1004][7:300].tobytes().replace(b"A", b"R", 2))]
    reads += synth.simulate_reads(rng, genomes, 2000 - len(reads), length=100)
    pairs = synth.simulate_reads(rng, genomes, 2000, length=150, n_rate=0.003)          # 1000 mate pairs
    pairs[0] = np.frombuffer(b"ACGTAC", dtype=np.uint8)                                   # short mate: ambig wraps (u32)
    pairs[5] = np.zeros(0, dtype=np.uint8)

    def run(unit_reads, paired):
        n_units = len(unit_reads) // (2 if paired else 1)
        res = np.zeros((n_units, 4), dtype=np.uint32)
        hits_all, hoffs, lines, fq = [], [0], [], []
        for u in range(n_units):
            r1 = unit_reads[2 * u] if paired else unit_reads[u]
            r2 = unit_reads[2 * u + 1] if paired else None
            k1 = kmers_of(r1.tobytes(), k)
            k2 = kmers_of(r2.tobytes(), k) if paired else None
            out4 = np.zeros(4, dtype=np.uint32)
            hits = np.zeros(k1.size + (k2.size if paired else 0) + 1, dtype=np.uint32)
            R.ref_classify_kmers(db, tax, k, p64(k1), k1.size, r1.size, p64(k2) if paired else None,
                                 k2.size if paired else 0, r2.size if paired else 0, p32(out4), p32(hits), hits.size)
            res[u] = out4
            h = hits[:out4[3]]
            hits_all.append(h.copy()); hoffs.append(hoffs[-1] + h.size)
            name = ("r%d" % u).encode()
            buf = C.create_string_buffer(64 + 16 * (h.size + 4))
            n = R.ref_kraken_line(name, int(out4[0]), r1.size, int(out4[1]), int(out4[2]), p32(h), h.size, buf, len(buf))
            assert n >= 0
            lines.append(buf.raw[:n])
            if u % 10 == 0:                                     # FASTQ-comment records, terse and verbose
                q1 = bytes((33 + (i * 7 + u) % 40) for i in range(r1.size)) if u % 20 == 0 else None   # None: FASTA input
                for verbose in (0, 1):
                    buf2 = C.create_string_buffer(4096 + 16 * h.size + 4 * (r1.size + (r2.size if paired else 0)))
                    n2 = R.ref_fastq_record(name, r1.tobytes(), q1, r1.size,
                                            (name + b"_m") if paired else None, r2.tobytes() if paired else None,
                                            None, r2.size if paired else 0,
                                            int(out4[0]), int(out4[1]), int(out4[2]), p32(h), h.size, verbose, int(paired),
                                            buf2, len(buf2))
                    assert n2 >= 0
                    fq.append(buf2.raw[:n2])
        return res, (np.concatenate(hits_all) if hits_all else np.zeros(0, np.uint32)), np.array(hoffs, dtype=np.uint64), lines, fq

    s_res, s_hits, s_hoffs, s_lines, s_fq = run(reads, False)
    p_res, p_hits, p_hoffs, p_lines, p_fq = run(pairs, True)
    R.ref_khc_free(db); R.ref_khp_free(tax)
    sb, so = synth.concat(reads)
    pb, po = synth.concat(pairs)

    def blob(lst):
        offs = np.zeros(len(lst) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in lst])
        return np.frombuffer(b"".join(lst), dtype=np.uint8).copy(), offs

    gl, go = blob([genomes[t].tobytes() for t in synth.LEAVES])
    slb, slo = blob(s_lines); plb, plo = blob(p_lines); sfb, sfo = blob(s_fq); pfb, pfo = blob(p_fq)
    np.savez_compressed(
        os.path.join(HERE, "classify_ref.npz"), k=np.array(k), tax_child=ch, tax_parent=pa,
        genome_taxid=np.array(synth.LEAVES, dtype=np.uint32), genome_bases=gl, genome_offs=go,
        db_keys=sk, db_vals=sv, db_hdr=hdr, db_flags=flags, db_keys_arr=keys, db_vals_arr=vals, db_table_bytes=table_bytes,
        s_bases=sb, s_offs=so, s_res=s_res, s_hits=s_hits, s_hoffs=s_hoffs, s_lines=slb, s_lines_offs=slo, s_fq=sfb, s_fq_offs=sfo,
        p_bases=pb, p_offs=po, p_res=p_res, p_hits=p_hits, p_hoffs=p_hoffs, p_lines=plb, p_lines_offs=plo, p_fq=pfb, p_fq_offs=pfo)
    print("classify_ref.npz: %d keys, %d reads (%d classified), %d pairs (%d classified)"
          % (sk.size, len(reads), int((s_res[:, 0] != 0).sum()), len(pairs) // 2, int((p_res[:, 0] != 0).sum())))


# ------------------------------------------------------------------ ingest (kseq_read / bseq_read) and nodes.dmp

FASTX_TEXTS = {
    "fa_multi": b">s1 first comment\nACGTACGT\nACGT\n>s2\nTTTT\n\n>s3\tc with tab\nAC\nGT\n",
    "fa_crlf": b">s1 c1\r\nACGT\r\nAC\r\n>s2\r\nGG\r\n",
    "fq_basic": b"@r1/1 x\nACGTN\n+\nIIIII\n@r2/2\nAC\n+r2\nII\n",
    "fq_multi": b"@q1\nACGT\nAC\n+\nIIII\nII\n@q2 cm\nA\n+\n@\n",
    "fq_at_qual": b"@q1\nACGT\n+\n@III\n@q2\nAAAA\n+\n+@+@\n",
    "fq_trunc": b"@q1\nACGT\n+\nIIII\n@q2\nACGT\n+\nII",
    "fa_noeol": b">last\nACGTACGTAC",
    "fa_junk_head": b"junk before\n>s1\nACGT\n",
    "fa_empty_seq": b">e1\n>e2 c\nAC\n",
    "fa_names": b">a/1\nAC\n>b/x\nAC\n>c/12\nAC\n>/1\nAC\n>d/3 comment/1\nAC\n",
    "empty": b"",
}


def parse_with_reference(R, path1, path2, chunk):
    cap = 1 << 20
    blob = C.create_string_buffer(cap)
    ls = (C.c_int32 * 4096)(); ck = (C.c_int32 * 4096)()
    n = R.ref_bseq_read_all(path1.encode(), path2.encode() if path2 else None, chunk, blob, cap, ls, ck, 4096)
    assert n >= 0, n
    fields = blob.raw.split(b"\0")[:4 * n]
    return n, fields, list(ls[:n]), list(ck[:n])


def gen_ingest(R, rng):
    out = {}
    names = []
    with tempfile.TemporaryDirectory() as td:
        texts = dict(FASTX_TEXTS)
        # a longer FASTQ so that chunking (chunk_size bases, even record counts) shows: 300 reads x 50 bp
        big = b"".join(b"@b%d/%d c%d\n%s\n+\n%s\n" % (i, 1 + (i & 1), i, bytes(synth.rand_seq(rng, 50)), b"I" * 50) for i in range(300))
        texts["fq_big"] = big
        texts["fq_big_mate"] = b"".join(b"@b%d/2\n%s\n+\n%s\n" % (i, bytes(synth.rand_seq(rng, 40)), b"F" * 40) for i in range(290))
        for nm, txt in texts.items():
            p = os.path.join(td, nm)
            open(p, "wb").write(txt)
        cases = [(nm, None, 1 << 20) for nm in texts if nm != "fq_big_mate"]
        cases += [("fq_big", None, 1000), ("fq_big", "fq_big_mate", 1200), ("fa_multi", "fa_crlf", 1 << 20), ("fa_crlf", "fa_multi", 1 << 20)]
        gzp = os.path.join(td, "fq_basic_gz")
        with gzip.open(gzp, "wb") as f:
            f.write(texts["fq_basic"])
        texts["fq_basic_gz"] = open(gzp, "rb").read()
        cases.append(("fq_basic_gz", None, 1 << 20))
        for ci, (a, b, chunk) in enumerate(cases):
            n, fields, ls, ck = parse_with_reference(R, os.path.join(td, a), os.path.join(td, b) if b else None, chunk)
            out["case%d_file1" % ci] = np.array(a); out["case%d_file2" % ci] = np.array(b or ""); out["case%d_chunk" % ci] = np.array(chunk)
            out["case%d_fields" % ci] = np.frombuffer(b"\0".join(fields) + (b"\0" if fields else b""), dtype=np.uint8).copy()
            out["case%d_lseq" % ci] = np.array(ls, dtype=np.int32); out["case%d_chunk_of" % ci] = np.array(ck, dtype=np.int32)
            names.append(a)
        out["n_cases"] = np.array(len(cases))
        for nm, txt in texts.items():
            out["text_" + nm] = np.frombuffer(txt, dtype=np.uint8).copy()
        # nodes.dmp texts -> build_parent_map pairs (well-formed lines only: a line without '|' stores tax_t(-1), which the
        # build rejects at load -- a documented deviation)
        dmps = {
            "plain": b"".join(b"%d\t|\t%d\t|\tno rank\t|\t\t|\n" % (c, p) for c, p in synth.TAX_PAIRS),
            "comments": b"# header\n1\t|\t1\t|\n\n2\t|\t1\t|\n#x\n7\t|\t2\t|\n7\t|\t1\t|\n",      # blank + '#' lines, duplicate child
            "no_root_line": b"5\t|\t1\t|\n9\t|\t5\t|\n",                                             # key 1 is forced in
            "tight": b"10|\t3\n3|\t1\n",                                                             # atoi(p + 2) on short separators
        }
        for nm, txt in dmps.items():
            p = os.path.join(td, nm + ".dmp")
            open(p, "wb").write(txt)
            h = R.ref_build_parent_map(p.encode())
            assert h
            n = R.ref_khp_size(h)
            c = np.zeros(n, dtype=np.uint32); q = np.zeros(n, dtype=np.uint32)
            assert R.ref_khp_pairs(h, p32(c), p32(q), n) == n
            o = np.argsort(c)
            out["dmp_" + nm] = np.frombuffer(txt, dtype=np.uint8).copy(); out["dmp_" + nm + "_child"] = c[o]; out["dmp_" + nm + "_parent"] = q[o]
            R.ref_khp_free(h)
        open(os.path.join(td, "one.dmp"), "wb").write(b"# nothing\n")
        out["dmp_too_small_throws"] = np.array(R.ref_build_parent_map(os.path.join(td, "one.dmp").encode()) is None)
    np.savez_compressed(os.path.join(HERE, "ingest_ref.npz"), **out)
    print("ingest_ref.npz: %d parse cases, %d nodes.dmp texts" % (len(cases), len(dmps)))


def main():
    R = O.ref()
    assert R is not None and hasattr(R, "ref_resolve_pairs"), "build oracle/_ref first (make -C oracle)"
    gen_resolve(R, np.random.default_rng(20260929))
    gen_classify(R, np.random.default_rng(20260930))
    gen_ingest(R, np.random.default_rng(20260931))


if __name__ == "__main__":
    main()
