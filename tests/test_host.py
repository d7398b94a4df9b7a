"""CPU tests of the C++ host side (bonsai_amd/csrc/host via libbns_host.so): bns.db, nodes.dmp, FASTA/FASTQ
batching, output records.  Cross-checked against the oracle's independent restatements."""
import gzip
import os

import numpy as np
import pytest

import synth


@pytest.fixture(scope="module")
def hostio():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from bonsai_amd.build import build_device_library
    build_device_library()
    subprocess.run(["make", "-s", "-C", os.path.join(root, "bonsai_amd", "csrc", "host")], check=True)
    from bonsai_amd import hostio
    return hostio


def test_db_read_write_cross(hostio, oracle, small_world, tmp_path):
    w = small_world
    f0, k0, v0 = w.table.arrays()
    for width, name in ((1, "a.db"), (2, "b.db"), (2, "c.db.gz")):
        p = str(tmp_path / name)
        assert oracle.db_write(p, 31, 50, [1] * 15 + [0] * 15, w.table, spacing_width=width) == 0
        d = hostio.read_db(p)
        assert (d["k"], d["w"], d["spacing_width"]) == (31, 50, width)
        assert d["gaps"].tolist() == [1] * 15 + [0] * 15
        assert (d["n_buckets"], d["size"], d["n_occupied"], d["upper_bound"]) == w.table.header()
        f, k, v = w.table.arrays()            # db_write zeroed the empty slots in place
        assert np.array_equal(d["flags"], f) and np.array_equal(d["keys"], k) and np.array_equal(d["vals"], v)
        # and the other direction: C++ writer -> oracle reader
        p2 = str(tmp_path / ("x_" + name))
        hostio.write_db(p2, 31, 50, d["gaps"], [d["n_buckets"], d["n_occupied"], d["size"], d["upper_bound"]],
                        d["flags"], d["keys"], d["vals"], spacing_width=width)
        k2, w2, g2, t2, got_w = oracle.db_read(p2)
        assert (k2, w2, got_w) == (31, 50, width) and g2.tolist() == d["gaps"].tolist()
        f2, kk2, v2 = t2.arrays()
        assert np.array_equal(f2, f) and np.array_equal(kk2, k) and np.array_equal(v2, v)
        if not name.endswith(".gz"):
            assert open(p, "rb").read() == open(p2, "rb").read()
    with pytest.raises(hostio.HostIOError):
        hostio.read_db(str(tmp_path / "missing.db"))
    bad = tmp_path / "bad.db"
    bad.write_bytes(b"\x1f\x00\x00\x00" * 40)
    with pytest.raises(hostio.HostIOError):
        hostio.read_db(str(bad))


def test_nodes_dmp(hostio, oracle, tmp_path):
    p = tmp_path / "nodes.dmp"
    synth.write_nodes_dmp(str(p))
    with open(p, "a") as f:
        f.write("#comment\n\n7\t|\t3\t|\tspecies\t|\n1001\t|\t102\t|\tlast one wins\t|\n")
    a = hostio.read_nodes_dmp(str(p))
    b = oracle.Taxonomy(path=str(p)).parent
    assert np.array_equal(a, b)
    assert a[1] == 0 and a[7] == 3 and a[1001] == 102
    one = tmp_path / "one.dmp"
    one.write_text("1\t|\t1\t|\n")
    with pytest.raises(hostio.HostIOError):
        hostio.read_nodes_dmp(str(one))            # fewer than 2 entries (util.h:782)


def test_parse_spacing(hostio, oracle):
    for s, k in (("1x15,0x15", 31), ("", 31), ("3,1,4", 4), ("0x30", 31), ("2x3,7,1x2", 7)):
        out = np.zeros(64, dtype=np.uint16)
        n = oracle.lib().bo_parse_spacing(s.encode(), k, out.ctypes.data_as(oracle.u16p))
        assert hostio.parse_spacing(s, k).tolist() == out[:n].tolist()


FASTA = b""">r1/1 first comment\r
ACGTACGTAC
GGGG

TT
>r2 x
>r3
NNNN
@notaheader
"""
FASTQ = b"""@q1/1 c1
ACGTN
+
IIIII
@q2
GG
TT
+q2
IIII
@q3/2
A
+
I
"""


def test_fastx_reader(hostio, tmp_path):
    fa = tmp_path / "a.fa"
    fa.write_bytes(FASTA)
    recs, _ = hostio.read_fastx(str(fa))
    # kseq: name = first token, comment = rest; multi-line sequence joined; '\r' stripped; '@' starts a record
    assert recs[0] == (b"r1", b"first comment", b"ACGTACGTACGGGGTT", b"")
    assert recs[1] == (b"r2", b"x", b"", b"")
    assert recs[2] == (b"r3", b"", b"NNNN", b"")
    assert recs[3][0] == b"notaheader" and recs[3][2] == b""
    fq = tmp_path / "b.fq.gz"
    with gzip.open(fq, "wb") as f:
        f.write(FASTQ)
    recs, _ = hostio.read_fastx(str(fq))
    assert recs == [(b"q1", b"c1", b"ACGTN", b"IIIII"), (b"q2", b"", b"GGTT", b"IIII"), (b"q3", b"", b"A", b"I")]
    # truncated quality ends the stream like kseq_read's -2
    tq = tmp_path / "t.fq"
    tq.write_bytes(b"@a\nACGT\n+\nII\n")
    recs, _ = hostio.read_fastx(str(tq))
    assert recs == []


def test_fastx_pairs_and_chunks(hostio, tmp_path):
    rng = np.random.default_rng(3)
    r1 = tmp_path / "r1.fq"; r2 = tmp_path / "r2.fq"
    seqs = [(synth.rand_seq(rng, 50).tobytes(), synth.rand_seq(rng, 40).tobytes()) for _ in range(101)]
    with open(r1, "wb") as a, open(r2, "wb") as b:
        for i, (s1, s2) in enumerate(seqs):
            a.write(b"@p%d/1\n%s\n+\n%s\n" % (i, s1, b"I" * 50))
            b.write(b"@p%d/2\n%s\n+\n%s\n" % (i, s2, b"J" * 40))
    recs, chunks = hostio.read_fastx(str(r1), str(r2), chunk_size=900)
    assert len(recs) == 202 and chunks == 11              # 10 pairs (900 bases) per bseq_read call
    for i, (s1, s2) in enumerate(seqs):
        assert recs[2 * i] == (b"p%d" % i, b"", s1, b"I" * 50)        # mates interleaved, /1 /2 trimmed
        assert recs[2 * i + 1] == (b"p%d" % i, b"", s2, b"J" * 40)
    # second file shorter: stops at the shorter one (kseq_declare.h:117-120)
    with open(r2, "wb") as b:
        for i, (s1, s2) in enumerate(seqs[:5]):
            b.write(b"@p%d/2\n%s\n+\n%s\n" % (i, s2, b"J" * 40))
    recs, _ = hostio.read_fastx(str(r1), str(r2))
    assert len(recs) == 10


def test_kraken_line_matches_oracle(hostio, oracle):
    rng = np.random.default_rng(8)
    pool = np.array([0, 1, 7, 1001, 4294967295, 123456], dtype=np.uint32)
    for _ in range(300):
        n = int(rng.integers(0, 40))
        hits = pool[rng.integers(0, pool.size, size=n)]
        taxon = int(rng.choice(pool)) if n else 0
        missing, ambig = int(rng.integers(0, 3)) * int(rng.integers(0, 200)), int(rng.choice([0, 0, 5, 4294967238]))
        l_seq = int(rng.integers(0, 300))
        a = hostio.kraken_line("read_%d" % n, l_seq, taxon, missing, ambig, hits)
        b = oracle.kraken_line("read_%d" % n, taxon, l_seq, missing, ambig, hits)
        assert a == b
    assert hostio.kraken_line("x", 60, 0, 30, 0, []) == b"U\tx\t0\t60\tM:30\t0:0\n"
    assert hostio.kraken_line("x", 150, 9, 0, 0, [9, 9, 0, 4294967295, 9]) == b"C\tx\t9\t150\t9:2\tU:1\tA:1\t9:1\n"


def test_fastq_record_known_answers(hostio):
    """classifier.h:72-108 reproduced as written."""
    m1 = (b"q1", b"ACGT", b"IIII")
    assert hostio.fastq_record(m1, None, 9, 2, 0, [9, 9], verbose=False) == b"q1 C\t9\t4\tM:2\nACGT\n+\nIIII\n"
    assert hostio.fastq_record(m1, None, 9, 0, 3, [9, 9], verbose=True) == b"q1 C\t9\t4\tA:3\t9:2\nACGT\n+\nIIII\n"
    assert hostio.fastq_record(m1, None, 0, 1, 0, [], verbose=True) == b"q1 U\t0\t4\tM:1\t0:0\nACGT\n+\nIIII\n"
    # FASTA record (no quality): the sequence stands in for the quality line; mate 2 repeats the comment
    m1 = (b"a", b"AC", None); m2 = (b"b", b"GT", None)
    assert hostio.fastq_record(m1, m2, 5, 0, 0, [5], verbose=False) == b"a C\t5\t2\nAC\n+\nAC\nb C\t5\t2\n\nGT\n+\nGT\n"


def test_genome_name_and_taxid(hostio, oracle, tmp_path):
    for h in ("NC_000913.3 Escherichia coli str. K-12", "gi|556503834|ref|NC_000913.3| Escherichia coli", "gi|1|gb|X.1|",
              "plain", "a|b|c|", "weird\ttab sep"):
        assert hostio.genome_name(h) == oracle.genome_name(h), h
    assert hostio.genome_name("gi|556503834|ref|NC_000913.3| Escherichia coli") == "NC_000913.3"
    m = tmp_path / "nameidmap.txt"
    m.write_text("#comment\nNC_1\t562\nNC_2\t1280\nNC_1\t511145\n")       # later line wins (util.h:709-719)
    g1 = tmp_path / "g1.fna"; g1.write_text(">NC_1 some genome\nACGT\n")
    g2 = tmp_path / "g2.fna.gz"
    with gzip.open(g2, "wt") as f:
        f.write(">gi|9|ref|NC_2| x\nACGT\n")
    g3 = tmp_path / "g3.fna"; g3.write_text(">unknown\nACGT\n")
    assert hostio.get_taxid(str(g1), str(m)) == 511145
    assert hostio.get_taxid(str(g2), str(m)) == 1280
    assert hostio.get_taxid(str(g3), str(m)) == 1                                # unknown name -> root (util.h:924)


# ---- differential test of the block reader against the character-level kseq_read restatement (tests/kseq_py.py, pinned to the
# reference's own reader by tests/test_ingest_oracle.py) --------------------------------------------------------------------------
import kseq_py
import ingest_fuzz


def _kseq_records(data: bytes):
    return kseq_py.read_all(data, trim=True)


def _fuzz_doc(rng, n_rec, fastq_p, messy):
    parts = []
    for i in range(n_rec):
        L = int(rng.integers(0, 90)) if messy else int(rng.integers(20, 90))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), size=L))
        name = b"r%d" % i + (b"/1" if rng.random() < 0.3 else b"")
        comment = rng.choice([b"", b" c", b"\tx y", b" a\r"]) if messy else b" c"
        eol = b"\r\n" if (messy and rng.random() < 0.2) else b"\n"
        if rng.random() < fastq_p:
            lines = [seq] if not messy or rng.random() < 0.7 else [seq[:L // 2], seq[L // 2:]]
            qual = bytes(rng.choice(np.frombuffer(b"@>+I#5", dtype=np.uint8), size=L))
            if messy and L > 4 and rng.random() < 0.15:          # wrapped quality; now and then one character short or long
                cut = int(rng.integers(1, L - 1))
                tail = qual[cut:] if rng.random() < 0.8 else (qual[cut:-1] if rng.random() < 0.5 else qual[cut:] + b"I")
                qual = qual[:cut] + eol + tail
            parts.append(b"@" + name + comment + eol + eol.join(lines) + eol + b"+" + (name if rng.random() < 0.2 else b"") + eol + qual + eol)
        else:
            w = int(rng.integers(10, 60))
            lines = [seq[j:j + w] for j in range(0, L, w)] or [b""]
            parts.append(b">" + name + comment + eol + eol.join(lines) + eol)
        if messy and rng.random() < 0.1:
            parts.append(b"\n")
    doc = b"".join(parts)
    if messy and rng.random() < 0.5:
        doc = doc[:len(doc) - int(rng.integers(0, 30))]      # truncated tail
    return doc


@pytest.mark.parametrize("seed", range(18))
def test_fastx_reader_matches_kseq_fuzz(hostio, tmp_path, seed):
    rng = np.random.default_rng(1000 + seed)
    messy = seed % 2 == 1
    doc = _fuzz_doc(rng, 400, fastq_p=(0.0, 1.0, 0.5)[seed % 3], messy=messy)
    p = tmp_path / "f.fx"
    p.write_bytes(doc)
    want = _kseq_records(doc)
    # default blocks; tiny blocks (every record crosses a refill, some need the concatenating path); medium blocks
    for kw in ({}, {"block_bytes": 97}, {"block_bytes": 4096}, {"block_bytes": 70000}):
        got, _ = hostio.read_fastx(str(p), **kw)
        assert got == want, kw


@pytest.mark.parametrize("seed", range(6))
def test_fastx_reader_matches_kseq_on_crlf_and_wrapped_quality(hostio, tmp_path, seed):
    """the documents the device parser's fuzz uses (tests/ingest_fuzz.py): CRLF line ends on some or all lines, wrapped quality, empty
    sequences, stray text -- among them the two places kseq is easy to get wrong: the quality loop reads at least ONE line, also behind
    an empty sequence (klib/kseq.h:217), and a one-byte last line without a newline keeps its byte, a '\\r' too (:135 is not reached)"""
    rng = np.random.default_rng(4200 + seed)
    for it in range(60):
        wild = float(rng.choice([0, 0.3])); crlf = float(rng.choice([0.3, 1.0])); wrapq = float(rng.choice([0, 0.5]))
        doc = ingest_fuzz.make_doc(rng, int(rng.integers(1, 60)), wild=wild, final_newline=bool(rng.integers(0, 2)), crlf=crlf, wrapq=wrapq,
                                   fastq_comments=False)
        p = tmp_path / "t.fq"
        p.write_bytes(doc)
        want = _kseq_records(doc)
        for kw in ({}, {"block_bytes": 61}):
            got, _ = hostio.read_fastx(str(p), **kw)
            assert got == want, (it, kw, doc[:200])


def test_fastx_reader_kseq_corner_known_answers(hostio, tmp_path):
    p = tmp_path / "k.fq"
    for doc in (b">x\r\nNGCY\r\n\r", b">x\nACGT\n\r", b"@r\r\n+\r\n\r\n@s\nAC\n+\nII\n", b"@a\n+\n@b\nACGT\n+\nIIII\n@c\nAC\n+\nII\n",
                b"@a\n\n+\n\n@b\nAC\n+\nII\n", b">x\n\r", b">x\n\r\n", b"@q\nAC\r\n+\r\nI\r\nI\r\n", b"@q\nA\n+\n\r"):
        p.write_bytes(doc)
        assert hostio.read_fastx(str(p))[0] == _kseq_records(doc), doc


def _big_doc(rng, n_rec, kind):
    """a few MB of records: strict four-line FASTQ with quality lines that like to start with '@' / '>' / '+', or wrapped FASTA"""
    parts = []
    for i in range(n_rec):
        L = int(rng.integers(30, 200))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L))
        if kind == "fastq":
            qual = bytes(rng.choice(np.frombuffer(b"@>+I#5", dtype=np.uint8), size=L))
            parts.append(b"@r%d c%d\n" % (i, i) + seq + b"\n+\n" + qual + b"\n")
        else:
            w = int(rng.integers(40, 80))
            parts.append(b">g%d d\n" % i + b"\n".join(seq[j:j + w] for j in range(0, L, w)) + b"\n")
    return b"".join(parts)


@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_parallel_stretches_give_the_sequential_records(hostio, tmp_path, kind):
    """one plain file parsed in stretches on 2-3 threads (ChunkSource): the cuts are record starts, the records and their order
    are those of the one-thread reader, whatever the stretch length and chunk size"""
    rng = np.random.default_rng(77)
    doc = _big_doc(rng, 20000, kind)
    p = tmp_path / "big.fx"
    p.write_bytes(doc)
    want, _ = hostio.read_fastx(str(p))
    starts = set()
    pos = 0
    for line in doc.split(b"\n"):
        if line[:1] == (b"@" if kind == "fastq" else b">") and (pos == 0 or doc[pos - 1:pos] == b"\n"):
            starts.add(pos)
        pos += len(line) + 1
    for seg in (1 << 18, 300_000, 1 << 20):
        cuts = hostio.find_cut_points(str(p), seg)
        assert cuts.size >= len(doc) // seg - 2 and all(int(c) in starts for c in cuts), seg
        if kind == "fastq":                                   # ('@' also begins quality lines: a cut must be a HEADER)
            assert all(doc[int(c):].split(b"\n", 1)[0].startswith(b"@r") for c in cuts)
        for threads, chunk in ((2, 1 << 16), (3, 5000), (2, 1 << 22)):
            got, n_st, fell = hostio.read_fastx_par(str(p), chunk_size=chunk, parser_threads=threads, segment_bytes=seg)
            assert n_st == cuts.size + 1 and not fell
            assert got == want, (seg, threads, chunk)
    # nothing is cut: a small file, a gzip file, one parser thread
    assert hostio.find_cut_points(str(p), len(doc)).size == 0
    gz = tmp_path / "big.fx.gz"
    with gzip.open(gz, "wb") as f:
        f.write(doc)
    assert hostio.find_cut_points(str(gz), 1 << 16).size == 0
    got, n_st, fell = hostio.read_fastx_par(str(gz), parser_threads=2, segment_bytes=1 << 16)
    assert got == want and n_st == 1
    got, n_st, fell = hostio.read_fastx_par(str(p), parser_threads=1, segment_bytes=1 << 18)
    assert got == want and n_st == 1


def test_parallel_stretch_that_ends_inside_a_record_falls_back(hostio, tmp_path):
    """cut offsets forced into the middle of records (find_cut_points would never give them): the stretch before such a cut does
    not end between two records, the rest of the file is parsed again by one sequential reader, the records are the same"""
    rng = np.random.default_rng(78)
    for kind in ("fastq", "fasta"):
        doc = _big_doc(rng, 6000, kind)
        p = tmp_path / ("f." + kind)
        p.write_bytes(doc)
        want, _ = hostio.read_fastx(str(p))
        good = hostio.find_cut_points(str(p), 1 << 17)
        assert good.size >= 3
        # forced cuts inside a header; FASTQ also inside the sequence line, the '+' line and the quality line (in FASTA any line that
        # starts with '>' ends the record before it, so a cut there cannot be wrong and nothing else is ever chosen)
        h = int(good[1])
        l1 = doc.index(b"\n", h) + 1
        bads = [h + 5]
        if kind == "fastq":
            l2 = doc.index(b"\n", l1) + 1
            l3 = doc.index(b"\n", l2) + 1
            bads += [l1 + 10, l2, l2 + 1, l3, l3 + 7]
        for bad in bads:
            cuts = np.array([good[0], bad, good[2]], dtype=np.uint64)
            got, n_st, fell = hostio.read_fastx_par(str(p), chunk_size=1 << 15, parser_threads=2, cuts=cuts)
            assert n_st == 4 and fell, (kind, bad)
            assert got == want, (kind, bad)


def test_paired_files_parsed_side_by_side(hostio, tmp_path):
    """two files, one parser thread each (ChunkSource): mates interleaved exactly as the one-thread bseq_read interleaves them, for
    FASTQ / wrapped FASTA / gzip, several chunk sizes, and files of unequal length (the shorter one ends the input)"""
    rng = np.random.default_rng(79)
    d1 = _big_doc(rng, 5000, "fastq"); d2 = _big_doc(rng, 5000, "fastq")
    f1 = _big_doc(rng, 3000, "fasta"); f2 = _big_doc(rng, 2500, "fasta")
    cases = []
    for nm, a, b in (("q", d1, d2), ("f", f1, f2), ("g", f2, f1), ("mix", d1, f1)):
        pa = tmp_path / (nm + "_1.fx"); pb = tmp_path / (nm + "_2.fx")
        pa.write_bytes(a); pb.write_bytes(b)
        cases.append((str(pa), str(pb)))
    ga = tmp_path / "z_1.fq.gz"; gb = tmp_path / "z_2.fq.gz"
    with gzip.open(ga, "wb") as f:
        f.write(d1)
    with gzip.open(gb, "wb") as f:
        f.write(d2)
    cases.append((str(ga), str(gb)))
    for pa, pb in cases:
        for chunk in (3000, 1 << 16, 1 << 24):
            want, _ = hostio.read_fastx(pa, pb, chunk_size=chunk)
            got, _, _ = hostio.read_fastx_par(pa, chunk_size=chunk, parser_threads=2, path2=pb)
            assert got == want, (pa, chunk)
            assert len(got) % 2 == 0 and len(got) >= 5000


def test_paired_side_by_side_truncated_record_falls_back(hostio, tmp_path):
    """a record cut short in the middle of one file of a pair: the one-thread reader drops it and pairs what follows with the wrong
    mates (the reference's behaviour, kseq_declare.h:112-145).  The two-thread reader stops parsing side by side there and pairs the
    rest on one thread the same way: same records, same (shifted) mates -- in file 1 (the record alone is dropped) and in file 2
    (the file-1 record read for it goes too)."""
    rng = np.random.default_rng(80)
    d1 = _big_doc(rng, 3000, "fastq"); d2 = _big_doc(rng, 3000, "fastq")

    def cut_record(d, i):
        cut = d.index(b"\n@r%d " % i) + 1
        rec_end = d.index(b"\n@r%d " % (i + 1)) + 1
        return d[:cut] + d[cut:rec_end - 20] + b"\n" + d[rec_end:]      # the quality line loses 19 characters
    checked = 0
    for which, idx in ((0, 1500), (1, 1501), (0, 2990), (1, 7)):
        a, b = (cut_record(d1, idx), d2) if which == 0 else (d1, cut_record(d2, idx))
        pa = tmp_path / ("t%d_%d_1.fq" % (which, idx)); pb = tmp_path / ("t%d_%d_2.fq" % (which, idx))
        pa.write_bytes(a); pb.write_bytes(b)
        for chunk in (4000, 30000):
            want, _ = hostio.read_fastx(str(pa), str(pb), chunk_size=chunk)
            if len(want) < 5000:
                continue            # (the truncated record was the first of a one-thread chunk: the reference ends the input there)
            got, _, _ = hostio.read_fastx_par(str(pa), chunk_size=chunk, parser_threads=2, path2=str(pb))
            assert got == want, (which, idx, chunk)
            assert len(got) % 2 == 0 and len(got) in (5996, 5998)
            checked += 1
    assert checked >= 6


def test_pack_container_format(tmp_path):
    """`bonsai pack` (host only): the container's chunks hold exactly bns_pack_reads' image of the reads -- lengths, words, sparse
    invalid-base list -- and the names, whatever the chunk size; a pair of files is interleaved"""
    import struct
    import subprocess
    import bonsai_amd
    BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bonsai_amd", "bin", "bonsai")
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = [bytes(rng.choice(alpha, size=int(n), p=[.24, .24, .24, .24, .04])) for n in rng.integers(1, 400, size=3000)]
    fq = tmp_path / "a.fq"
    fq.write_bytes(b"".join(b"@r%d c%d\n%s\n+\n%s\n" % (i, i, s, b"I" * len(s)) for i, s in enumerate(seqs)))
    fq2 = tmp_path / "b.fq"
    seqs2 = seqs[::-1]
    fq2.write_bytes(b"".join(b"@m%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(seqs2)))

    def parse(path):
        d = open(path, "rb").read()
        assert d[:8] == b"BNSPACK\x01"
        version, flags = struct.unpack_from("<II", d, 8)
        assert version == 1
        at, chunks = 32, []
        while at < len(d):
            magic, n, total, n_words, n_bad, names_bytes, payload = struct.unpack_from("<IIQQQQQ", d, at)
            assert magic == 0x4B4E4843
            p = at + 64
            lens = np.frombuffer(d, dtype="<u4", count=n, offset=p)
            w0 = (n * 4 + 7) & ~7
            words = np.frombuffer(d, dtype="<u8", count=n_words, offset=p + w0)
            bw = np.frombuffer(d, dtype="<u8", count=n_bad, offset=p + w0 + n_words * 8)
            bm = np.frombuffer(d, dtype="<u4", count=n_bad, offset=p + w0 + n_words * 8 + n_bad * 8)
            n0 = (w0 + n_words * 8 + n_bad * 12 + 7) & ~7
            names = d[p + n0:p + n0 + names_bytes].split(b"\0")[:-1]
            assert int(lens.sum()) == total and len(names) == n
            chunks.append((lens, words, bw, bm, names))
            at = p + payload
        assert at == len(d)
        return flags, chunks

    for chunk in (5000, 1 << 27):
        out = tmp_path / ("se_%d.bnsp" % chunk)
        assert subprocess.run([BIN, "pack", "-o", str(out), "-c", str(chunk), str(fq)], stderr=subprocess.PIPE).returncode == 0
        flags, chunks = parse(out)
        assert flags == 2 and (len(chunks) > 10 if chunk == 5000 else len(chunks) == 1)
        i = 0
        for lens, words, bw, bm, names in chunks:
            part = seqs[i:i + len(lens)]
            assert [len(s) for s in part] == lens.tolist() and names == [b"r%d" % (i + j) for j in range(len(lens))]
            ew, ebw, ebm = bonsai_amd.pack_reads(*bonsai_amd.concat_reads(part))
            assert np.array_equal(words, ew) and np.array_equal(bw, ebw) and np.array_equal(bm, ebm)
            i += len(lens)
        assert i == len(seqs)
    out = tmp_path / "pe.bnsp"
    assert subprocess.run([BIN, "pack", "-o", str(out), "-c", "7000", str(fq), str(fq2)], stderr=subprocess.PIPE).returncode == 0
    flags, chunks = parse(out)
    assert flags == 3
    got_names = [n for c in chunks for n in c[4]]
    assert got_names == [x for i in range(len(seqs)) for x in (b"r%d" % i, b"m%d" % i)]
    assert all(len(c[0]) % 2 == 0 for c in chunks)
    # no names: -n
    out = tmp_path / "nn.bnsp"
    assert subprocess.run([BIN, "pack", "-n", "-o", str(out), str(fq)], stderr=subprocess.PIPE).returncode == 0
    d = open(out, "rb").read()
    assert struct.unpack_from("<II", d, 8)[1] == 0 and struct.unpack_from("<IIQQQQQ", d, 32)[5] == 0


def test_bgzf_input_is_inflated_side_by_side(hostio, tmp_path):
    """a BGZF file (gzip members with their compressed size in a 'BC' subfield) is split at member boundaries and inflated on
    several threads: the records are those of the plain text, for members of every size (records crossing members, members and
    text blocks of all alignments), with libdeflate and with zlib; damage is an error, not a short input"""
    import subprocess
    import sys
    rng = np.random.default_rng(21)
    doc = _big_doc(rng, 40000, "fastq")
    plain = tmp_path / "d.fq"
    plain.write_bytes(doc)
    want, _ = hostio.read_fastx(str(plain))
    import gzip as _gz
    for tag, kw in (("std", {}), ("tiny", {"member_sizes": [1, 7, 300, 65280, 12345]}), ("lvl1", {"level": 1, "block": 40000})):
        p = tmp_path / ("d_%s.fq.gz" % tag)
        synth.write_bgzf(str(p), doc, **kw)
        assert _gz.open(p, "rb").read() == doc                       # (it IS a valid multi-member gzip file)
        for chunk, blk in ((1 << 20, 0), (5000, 70000)):
            got, _ = hostio.read_fastx(str(p), chunk_size=chunk, block_bytes=blk)
            assert got == want, (tag, chunk)
    # the zlib decoder (no libdeflate): another process, the choice is made once
    code = ("import sys; sys.path.insert(0, %r); from bonsai_amd import hostio; r, _ = hostio.read_fastx(%r); "
            "print(len(r), sum(len(x[2]) for x in r))" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "d_tiny.fq.gz")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BNS_NO_LIBDEFLATE="1", BNS_GZ_THREADS="3"), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == [str(len(want)), str(sum(len(x[2]) for x in want))]
    # a pair: one BGZF, one plain
    d2 = _big_doc(rng, 40000, "fastq")
    p2 = tmp_path / "e.fq"; p2.write_bytes(d2)
    want2, _ = hostio.read_fastx(str(plain), str(p2), chunk_size=1 << 16)
    got2, _ = hostio.read_fastx(str(tmp_path / "d_std.fq.gz"), str(p2), chunk_size=1 << 16)
    assert got2 == want2
    # damage: a flipped byte inside a member's payload, a truncated file
    raw = bytearray((tmp_path / "d_std.fq.gz").read_bytes())
    raw[len(raw) // 2] ^= 0x55
    bad = tmp_path / "bad.fq.gz"; bad.write_bytes(bytes(raw))
    with pytest.raises(hostio.HostIOError, match="BGZF"):
        hostio.read_fastx(str(bad))
    cut = tmp_path / "cut.fq.gz"; cut.write_bytes((tmp_path / "d_std.fq.gz").read_bytes()[:-5000])
    with pytest.raises(hostio.HostIOError, match="BGZF"):
        hostio.read_fastx(str(cut))


def _gz_member(data, level=6, name=None):
    """one gzip member around `data` (zlib's deflate; optional FNAME field)"""
    import struct
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    flg = 8 if name else 0
    hdr = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03" + ((name + b"\0") if name else b"")
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def test_one_gzip_stream_is_inflated_on_many_threads(hostio, tmp_path, monkeypatch):
    """a plain gzip file (one DEFLATE stream: no sync points) is cut into chunks of compressed bytes; every chunk finds a block
    header for itself, is decoded without its 32 KiB window (marker symbols) and resolved when the chunk in front is
    (csrc/host/pgzip.hpp).  The records are those of the text whatever the chunk size (down to chunks that lie inside one
    block), the compression level (stored, fixed and dynamic blocks), the number of members, padding between them and header
    fields; damage and truncation are errors."""
    rng = np.random.default_rng(31)
    doc = _big_doc(rng, 30000, "fastq")
    plain = tmp_path / "d.fq"
    plain.write_bytes(doc)
    want, _ = hostio.read_fastx(str(plain))
    third = len(doc) // 3
    files = {
        "l6": _gz_member(doc, 6),
        "l1_name": _gz_member(doc, 1, name=b"reads.fq"),
        "l9": _gz_member(doc, 9),
        "stored": _gz_member(doc, 0),
        # members: a large one, a tiny one (fixed-Huffman block), an empty one, the rest
        "members": _gz_member(doc[:third], 6) + _gz_member(doc[third:third + 40], 6) + _gz_member(b"", 6) +
                   _gz_member(doc[third + 40:], 4, name=b"x"),
    }
    monkeypatch.setenv("BNS_GZ_THREADS", "3")
    for tag, blob in files.items():
        p = tmp_path / ("d_%s.fq.gz" % tag)
        p.write_bytes(blob)
        assert gzip.open(p, "rb").read() == doc
        for chunk_bytes in (4096, 50001, 1 << 20):
            monkeypatch.setenv("BNS_PGZ_CHUNK", str(chunk_bytes))
            got, _ = hostio.read_fastx(str(p), chunk_size=1 << 18, block_bytes=0 if chunk_bytes != 50001 else 70000)
            assert got == want, (tag, chunk_bytes)
    monkeypatch.setenv("BNS_PGZ_CHUNK", "30000")
    # a pair of gzip files
    d2 = _big_doc(rng, 30000, "fastq")
    p2 = tmp_path / "e.fq"; p2.write_bytes(d2)
    g2 = tmp_path / "e.fq.gz"; g2.write_bytes(_gz_member(d2, 6))
    want2, _ = hostio.read_fastx(str(plain), str(p2), chunk_size=1 << 16)
    got2, _ = hostio.read_fastx(str(tmp_path / "d_l6.fq.gz"), str(g2), chunk_size=1 << 16)
    assert got2 == want2
    # damage: flipped bytes anywhere in the deflate data, a wrong CRC, a wrong length, a truncated file, a missing trailer
    good = files["l6"]
    # (text that no longer parses ends the input at the malformed record, as kseq does -- before the stream's end, where a
    # gzip reader learns that the checksum is wrong: then the records are fewer; text that still parses is read to the end
    # and the checksum is an error)
    for at in (len(good) // 7, len(good) // 2, len(good) - 5000):
        raw = bytearray(good); raw[at] ^= 0x41
        bad = tmp_path / "bad.fq.gz"; bad.write_bytes(bytes(raw))
        try:
            got, _ = hostio.read_fastx(str(bad))
            assert len(got) < len(want)
        except hostio.HostIOError as e:
            assert "gzip" in str(e)
    for off in (8, 4):
        raw = bytearray(good); raw[len(raw) - off] ^= 1
        bad = tmp_path / "badsum.fq.gz"; bad.write_bytes(bytes(raw))
        with pytest.raises(hostio.HostIOError, match="checksum"):
            hostio.read_fastx(str(bad))
    for cut in (5000, 8, 3):
        bad = tmp_path / "cut.fq.gz"; bad.write_bytes(good[:-cut])
        with pytest.raises(hostio.HostIOError, match="gzip"):
            hostio.read_fastx(str(bad))
    # trailing garbage behind the last member is ignored, as gzip -d (with a warning) and gzread do
    tail = tmp_path / "tail.fq.gz"; tail.write_bytes(good + b"\0\0\0garbage that is not a gzip header")
    got, _ = hostio.read_fastx(str(tail))
    assert got == want
    # zero padding between members ends the input, as it does for gzread (the reference's reader): same records from both readers
    padded = tmp_path / "pad.fq.gz"
    padded.write_bytes(_gz_member(doc[:third], 6) + b"\0" * 37 + _gz_member(doc[third:], 6))
    got_pad, _ = hostio.read_fastx(str(padded))
    # the zlib reader gives the same records (BNS_NO_PGZ)
    monkeypatch.setenv("BNS_NO_PGZ", "1")
    got, _ = hostio.read_fastx(str(tmp_path / "d_members.fq.gz"))
    assert got == want
    got_pad_z, _ = hostio.read_fastx(str(padded))
    assert got_pad == got_pad_z and 0 < len(got_pad) < len(want)


def test_parallel_gzip_long_matches_cross_the_buffer_growth(hostio, tmp_path, monkeypatch):
    """highly compressible text (N-run FASTA, one FASTQ record repeated): matches of 258 symbols back to back carry the decoder
    across the point where its 4 Mi-symbol buffer grows (round-4 advisor finding: the careful path behind the fast loop wrote a
    full-length match with 22 symbols of room left).  The records are those of the plain text, through chunk sizes that put the
    growth point inside and between chunks."""
    fasta = b">n1 long run\n" + (b"N" * 80 + b"\n") * 120000 + b">n2\nACGT\n" + (b"ACGTTGCA" * 10 + b"\n") * 60000
    rec = b"@r/1\n" + b"ACGTACGTTTGACCA" * 10 + b"\n+\n" + b"I" * 150 + b"\n"
    fastq = rec * 40000
    monkeypatch.setenv("BNS_GZ_THREADS", "3")
    for tag, doc in (("fa", fasta), ("fq", fastq)):
        plain = tmp_path / ("rep." + tag); plain.write_bytes(doc)
        want, _ = hostio.read_fastx(str(plain))
        for level in (1, 6, 9):
            p = tmp_path / ("rep_%s_%d.gz" % (tag, level))
            p.write_bytes(_gz_member(doc, level))
            for chunk_bytes in (4096, 20000, 1 << 20):
                monkeypatch.setenv("BNS_PGZ_CHUNK", str(chunk_bytes))
                got, _ = hostio.read_fastx(str(p))
                assert got == want, (tag, level, chunk_bytes)


def test_parallel_gzip_survives_random_damage(hostio, tmp_path, monkeypatch):
    """random byte damage to a gzip file: every outcome is an error, the intact records (a flip the format does not look at:
    header mtime, OS) or an input that ends early at a record that no longer parses (kseq's behaviour; the checksum is only
    known at the end of the stream) -- never a hang, a crash, or a full-length input with different text"""
    rng = np.random.default_rng(5)
    doc = _big_doc(rng, 8000, "fastq")
    plain = tmp_path / "d.fq"; plain.write_bytes(doc)
    want, _ = hostio.read_fastx(str(plain))
    good = _gz_member(doc[: len(doc) // 2], 6) + _gz_member(doc[len(doc) // 2:], 2)
    monkeypatch.setenv("BNS_GZ_THREADS", "2")
    monkeypatch.setenv("BNS_PGZ_CHUNK", "20000")
    n_err = 0
    for it in range(40):
        raw = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            raw[int(rng.integers(0, len(raw)))] ^= int(rng.integers(1, 256))
        if it % 5 == 4:
            del raw[int(rng.integers(len(raw) // 2, len(raw))):]
        bad = tmp_path / "fz.fq.gz"; bad.write_bytes(bytes(raw))
        try:
            got, _ = hostio.read_fastx(str(bad))
        except hostio.HostIOError:
            n_err += 1
            continue
        assert got == want or len(got) < len(want), it
    assert n_err >= 10


@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_bgzf_parsed_in_stretches_of_its_text_blocks(hostio, tmp_path, kind, monkeypatch):
    """a BGZF file on several parser threads (ChunkSource: a distributor cuts the inflated text blocks into stretches at record
    starts it finds in the text, parsers work on readers over in-memory blocks): the records and their order are those of the plain
    text on one thread -- quality lines that start with '@', wrapped FASTA, members of every size, stretches of one and of several
    blocks, two and three parsers; a cut forced into the middle of a record is caught and the rest parsed on one reader"""
    rng = np.random.default_rng(91)
    doc = _big_doc(rng, 150000, kind)                               # ~18-30 MB: several 4 MiB text blocks
    plain = tmp_path / ("d." + kind)
    plain.write_bytes(doc)
    want, _ = hostio.read_fastx(str(plain))
    for tag, kw in (("std", {}), ("tiny", {"member_sizes": [1, 7, 300, 65280, 12345]})):
        p = tmp_path / ("d_%s.gz" % tag)
        synth.write_bgzf(str(p), doc, **kw)
        for threads, seg, chunk in ((2, 1, 1 << 20), (3, 1, 70000), (2, 9 << 20, 1 << 22)):
            got, n_stretch, fell = hostio.read_fastx_par(str(p), chunk_size=chunk, parser_threads=threads, segment_bytes=seg)
            assert got == want, (tag, threads, seg)
            assert not fell
            if seg == 1:
                assert n_stretch >= 3                                # (it did run in stretches)
    # a bad cut at the second stretch: caught by the end-of-stretch check, everything from there on goes through one reader
    # (FASTQ: a record cut short has no quality line; a FASTA cut is a line that starts with '>', which ends a record whatever
    # came before -- there the cut itself is the guarantee, as for a plain file)
    if kind == "fastq":
        for at in ("1", "3"):
            monkeypatch.setenv("BNS_BGZF_FORCE_BAD_CUT", at)
            got, n_stretch, fell = hostio.read_fastx_par(str(tmp_path / "d_std.gz"), chunk_size=1 << 20, parser_threads=2, segment_bytes=1)
            assert got == want and fell
        monkeypatch.delenv("BNS_BGZF_FORCE_BAD_CUT")
    # one parser (the switch) and damage
    monkeypatch.setenv("BNS_BGZF_ONE_PARSER", "1")
    got, n_stretch, fell = hostio.read_fastx_par(str(tmp_path / "d_std.gz"), chunk_size=1 << 20, parser_threads=2, segment_bytes=1)
    assert got == want and n_stretch == 1
    monkeypatch.delenv("BNS_BGZF_ONE_PARSER")
    raw = bytearray((tmp_path / "d_std.gz").read_bytes())
    raw[len(raw) // 2] ^= 0x55
    bad = tmp_path / "bad.gz"
    bad.write_bytes(bytes(raw))
    with pytest.raises(hostio.HostIOError, match="BGZF"):
        hostio.read_fastx_par(str(bad), parser_threads=2, segment_bytes=1)
    # an empty member list / a tiny file
    small = tmp_path / "small.gz"
    synth.write_bgzf(str(small), doc[:doc.index(b"\n", 5000) + 1] if kind == "fasta" else b"".join(doc.split(b"\n")[i] + b"\n" for i in range(40)))
    w2, _ = hostio.read_fastx(str(small))
    g2, _, _ = hostio.read_fastx_par(str(small), parser_threads=2, segment_bytes=1)
    assert g2 == w2 and len(w2) > 0
