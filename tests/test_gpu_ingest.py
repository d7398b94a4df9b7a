"""FASTA / FASTQ text parsed on the device (bns_classify_text, csrc/bns_ingest.hip) against kseq_read / bseq_read: the
reference-made records of tests/golden/ingest_ref.npz and the character-level restatement tests/kseq_py.py (pinned to the
reference's reader in the CPU tier, tests/test_ingest_oracle.py) on random text -- names, sequence lengths, every base (through
the packed words), where a call stops and why -- then the classify results of the text path against bns_classify_batch on the
same records: single files, pairs, pieces of a file with a limit, text that is already in HBM."""
import gzip
import os

import numpy as np
import pytest

import bonsai_amd
from bonsai_amd import _lib
import ingest_fuzz
import kseq_py
import synth

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3}


def unpack(words, nmask, seq_len):
    """packed image (bns_pack_reads layout, dense flags) -> list of byte strings over ACGT with 'N' for flagged bases"""
    out = []
    off = 0
    for r, L in enumerate(int(x) for x in seq_len):
        wb = (off >> 5) + r
        s = bytearray(L)
        for i in range(L):
            w = int(words[wb + (i >> 5)]); m = int(nmask[wb + (i >> 5)])
            s[i] = ord("N") if (m >> (31 - (i & 31))) & 1 else b"ACGT"[(w >> (62 - 2 * (i & 31))) & 3]
        out.append(bytes(s))
        off += L
    return out


def norm(seq):
    return bytes(b"ACGT"[CODE[c]] if c in CODE else ord("N") for c in seq)


@pytest.fixture(scope="module")
def ctx():
    c = bonsai_amd.Context(0)
    c.set_encoder(31, None, canonicalize=True)
    yield c
    c.close()


def check_call(ctx, doc, final=True, trim=True, limit=None, want_words=True):
    """one call on `doc`: what it took is what kseq takes, in order, and kseq restarted at consumed[] reads the rest"""
    recs, rc, _ = kseq_py.read_until_error(doc, trim=trim)
    res = ctx.classify_text(doc, final=final, trim_readno=trim, parse_only=True, want_words=want_words, limit=limit)   # (the words come back for one-slice calls only)
    n, cons = res["n_records"], res["consumed"][0]
    assert res["status"] in (_lib.TEXT_OK, _lib.TEXT_IRREGULAR, _lib.TEXT_NO_RECORD)
    took = [r for r in recs if r[4] < cons]
    assert n == len(took), (n, len(took), res["status"], res["why"], cons)
    assert res["names"] == [r[0] for r in took]
    assert res["seq_len"].tolist() == [len(r[2]) for r in took]
    assert res["rec_pos"].tolist() == [r[4] for r in took]
    if n and want_words:
        assert unpack(res["words"], res["nmask"], res["seq_len"]) == [norm(r[2]) for r in took]
    # the rest of the input read from consumed on gives kseq's remaining records: consumed is a point between records
    rest, rc2, _ = kseq_py.read_until_error(doc[cons:], trim=trim)
    assert [r[:4] for r in rest] == [r[:4] for r in recs[n:]] and rc2 == rc
    if limit is not None and res["status"] == _lib.TEXT_OK and cons < len(doc):
        assert cons >= min(limit, len(doc)) or not final
    return res, recs


def test_parse_reference_vectors(ctx):
    """the 17+ crafted texts of ingest_ref.npz (records made by the reference's kseq_read / bseq_read): everything in the regular
    form -- round 6: CRLF text and wrapped quality included -- is taken whole; the others (broken records, text between records) are
    handed back at a record boundary"""
    IN = np.load(os.path.join(GOLD, "ingest_ref.npz"))
    seen_ok = seen_irregular = 0
    for ci in range(int(IN["n_cases"])):
        f1, f2 = str(IN["case%d_file1" % ci]), str(IN["case%d_file2" % ci])
        if f2:
            continue
        raw = IN["text_" + f1].tobytes()
        doc = gzip.decompress(raw) if f1.endswith("_gz") else raw
        res, recs = check_call(ctx, doc)
        fields = IN["case%d_fields" % ci].tobytes().split(b"\0")[:-1]
        exp = [tuple(fields[4 * i:4 * i + 4]) for i in range(len(fields) // 4)]
        if res["status"] == _lib.TEXT_OK:
            seen_ok += 1
            assert res["consumed"][0] == len(doc)
            assert res["names"] == [e[0] for e in exp] and res["seq_len"].tolist() == [len(e[2]) for e in exp], f1
        else:
            seen_irregular += 1
            assert res["why"] != 0
    assert seen_ok >= 9 and seen_irregular >= 1, (seen_ok, seen_irregular)


@pytest.mark.parametrize("seed", range(6))
def test_parse_fuzz_regular(ctx, seed):
    rng = np.random.default_rng(100 + seed)
    for it in range(25):
        kinds = (("fastq",), ("fasta",), ("fastq", "fasta"))[it % 3]
        # (round 6: every third text has CRLF records, every other one quality wrapped over several lines -- both regular now)
        doc = ingest_fuzz.make_doc(rng, int(rng.integers(1, 200)), wild=0.0, kinds=kinds, final_newline=bool(it % 4),
                                   crlf=(0.0, 0.4, 1.0)[it % 3] if it % 5 else 0.0, wrapq=0.5 if it % 2 else 0.0)
        if it % 11 == 0:
            doc += b"\n\n" + (b"@" if it % 2 else b">")       # a bare header byte at the very end: no record (klib/kseq.h:189)
        if it % 7 == 0:
            doc = b"\n\n" + doc
        res, recs = check_call(ctx, doc, trim=bool(it % 2))
        # (CRLF text: a blank line in FRONT of a record's sequence leaves its '\r' in the sequence -- klib/kseq.h:135 strips it only from
        # a string of more than one byte -- and a FASTQ record is then one byte longer than its quality: kseq's own error -2, or, when the
        # quality is wrapped, quality lines left over that kseq skips as text between records)
        if kseq_py.reads_cleanly(doc):
            assert res["status"] == _lib.TEXT_OK and res["consumed"][0] == len(doc) and res["n_records"] == len(recs), (seed, it, res["why"])
        else:
            assert res["status"] == _lib.TEXT_IRREGULAR and res["why"] & (4 | 8), (seed, it, res["why"])


@pytest.mark.parametrize("seed", range(4))
def test_parse_fuzz_wild(ctx, seed):
    """text kseq reads but the kernels do not take: every call stops at a point between records and says why"""
    rng = np.random.default_rng(200 + seed)
    n_irr = 0
    for it in range(40):
        doc = ingest_fuzz.make_doc(rng, int(rng.integers(1, 60)), wild=0.5, final_newline=bool(it % 3))
        res, recs = check_call(ctx, doc)
        if res["status"] == _lib.TEXT_IRREGULAR:
            n_irr += 1
            assert res["why"] != 0 and res["n_records"] == 0
    assert n_irr >= 10


def test_parse_irregular_reasons(ctx):
    ok = b"@a\nACGT\n+\nIIII\n"
    cases = {
        b"junk\n" + ok: 2,                                            # LEADING
        ok + b"stray\n" + ok: 4,                                      # AFTER_QUAL
        b"@a\nACGT\n+\nIII\n" + ok: 8,                                # QUAL_LEN (kseq reads the next header as quality)
        b"@a\nACGT\n+\nIIIII\n": 8,
        b"@a\nACGTAC\n+\nIII\nIIII\n": 8,                             # wrapped quality that overshoots
        b"@a\nACGT\n+": 8,                                            # '+' line cut off by the end of the input: kseq's -2
        b"@a\nACGT\n" + b"+\n" * 40 + ok: 4,                          # (round 5: PLUS_RUN) four one-byte quality lines, then '+' lines between records
        b">g\n" + b"ACGTACGTACGTACGTACGT\n" * 5000 + b">h\nAC\n": 32,                 # LONG_RECORD
    }
    for doc, bit in cases.items():
        res, _ = check_call(ctx, doc)
        assert res["status"] == _lib.TEXT_IRREGULAR and res["why"] & bit, (doc[:30], res["why"], bit)
    # ... and the near misses that ARE regular
    for doc in (ok + b"\n\n" + ok, b"@a\nAC\nGT\n+\nIIII\n", b"@a\n\n+\n\n" + ok, b"@a\nACGT\n+\n@III\n" + ok, b"@a\nACGT\n+\n+III\n" + ok,
                b"@a\nACGT\n+\nIIII", b">x\n>y\nAC\n>z", b"@a\nACGT\n+\nIIII\n@", b"", b"\n\n",
                # round 6: CRLF (ks_getuntil2 strips one '\r' per appended line, klib/kseq.h:135) ...
                b"@a\r\nACGT\r\n+\r\nIIII\r\n" + ok, b">x c\r\nAC\r\nGT\r\n\r\n>y\r\nA\r\n", b"@a\r\n\r\nACGT\r\n+\r\n\rIIII\r\n",
                b"@a\nACGT\r\r\n+\nIIIII\n", b"@a\n\r\n+\n\r\n" + ok, b">x\r\n\r\n\r\nAC\r\n", b">x\r\nNGCY\r\n\r", b">x\r\nNGCY\r\nA\r", b"@a\r\nAC\r\n+\r\nI\r\nI\r",
                # ... and quality over several lines, whatever they start with (klib/kseq.h:217)
                b"@a\nACGTAC\n+\nIII\nIII\n" + ok, b"@a\nACGTAC\n+\n@II\n@II\n" + ok, b"@a\nACGTAC\n+\n@I\n+I\n>I\n" + ok,
                b"@a\nACGTACG\n+\n@b\nAC\n+\nII\n" + ok, b"@a\nACGTAC\n+\n@x\nAAA\n+\n@r2\nACG\n+\nIII\n" + ok, b"@a\nACGTAC\n+\nIII\n\nIII\n" + ok, b"@a\nAC\nGT\n+\n@\n+\n@\n+\n@r\nAC\n+\nII\n"):
        res, recs = check_call(ctx, doc)
        assert res["status"] == _lib.TEXT_OK and res["consumed"][0] == len(doc) and res["n_records"] == len(recs), doc


@pytest.mark.parametrize("window", [97, 1000, 30000])
def test_parse_in_windows(ctx, window):
    """the host loop: a window of the file per call, not final until the end of the file; the next window starts at consumed[].
    Records and order are those of one parse of the whole text, whatever the window (a window smaller than a record grows)."""
    rng = np.random.default_rng(7)
    doc = ingest_fuzz.make_doc(rng, 600, wild=0.0, final_newline=False)
    recs, rc, _ = kseq_py.read_until_error(doc)
    pos, names, lens, w = 0, [], [], window
    n_calls = 0
    while pos < len(doc):
        end = min(len(doc), pos + w)
        res = ctx.classify_text(doc[pos:end], final=end == len(doc), trim_readno=True, parse_only=True)
        n_calls += 1
        assert res["status"] in (_lib.TEXT_OK, _lib.TEXT_NO_RECORD), res["why"]
        names += res["names"]; lens += res["seq_len"].tolist()
        if res["consumed"][0] == 0 and end < len(doc):
            w *= 2                                              # less than one record in the window
            continue
        pos += res["consumed"][0]
        w = window
        if end == len(doc) and res["status"] == _lib.TEXT_OK:
            assert pos == len(doc)
    assert names == [r[0] for r in recs] and lens == [len(r[2]) for r in recs]
    assert n_calls > 3


def test_limit_cuts_a_file_into_stretches(ctx):
    """pieces of one file for several devices: a stretch is the records that START in [begin, nominal end); the call is handed
    text beyond the nominal end (so that the record that straddles it is whole) and told the limit.  Stretches concatenate to the
    whole file whatever the nominal ends are."""
    rng = np.random.default_rng(8)
    doc = ingest_fuzz.make_doc(rng, 900, wild=0.0, final_newline=True)
    recs, _, _ = kseq_py.read_until_error(doc)
    for n_cuts in (1, 3, 8):
        nominal = sorted(int(x) for x in rng.integers(1, len(doc), size=n_cuts)) + [len(doc)]
        names, begin = [], 0
        for end in nominal:
            if begin >= len(doc):
                break
            hi = min(len(doc), end + 4096)                      # slack: at least one whole record behind the nominal end
            res = ctx.classify_text(doc[begin:hi], final=hi == len(doc), trim_readno=True, parse_only=True, limit=max(0, end - begin))
            assert res["status"] == _lib.TEXT_OK
            names += res["names"]
            begin += res["consumed"][0]
            assert begin >= min(end, len(doc)) or res["n_records"] == 0
        assert begin == len(doc) and names == [r[0] for r in recs], n_cuts


@pytest.fixture(scope="module")
def world(oracle):
    return synth.make_world(oracle, seed=21, k=31, genome_len=5000)


def fastq_of(reads, names=None, wrap=0, fasta=False):
    out = []
    for i, r in enumerate(reads):
        s = r.tobytes() if hasattr(r, "tobytes") else bytes(r)
        nm = names[i] if names else b"read%d/1" % i
        body = b"\n".join(s[j:j + wrap] for j in range(0, len(s), wrap)) if wrap else s
        out.append((b">" + nm + b" c\n" + body + b"\n") if fasta else (b"@" + nm + b"\n" + body + b"\n+\n" + b"I" * len(s) + b"\n"))
    return b"".join(out)


@pytest.mark.parametrize("form", ["fastq", "fasta_wrapped", "fastq_wrapped"])
def test_classify_text_equals_classify_batch(world, form):
    """the whole chain on text: taxon / missing / ambig / n_hits and the hit runs are those of bns_classify_batch(_runs) on the
    same records; one piece, (BNS_DBG_SLICE_8K) many pieces of 8 KiB whose records all go into ONE classify launch, and
    (BNS_DBG_BATCH_TINY as well) many pieces and a classify launch per >= 64 records"""
    w = world
    c = bonsai_amd.Context(0)
    c.set_encoder(31, None, canonicalize=True)
    c.load_table(w.n_buckets, w.flags, w.keys, w.vals)
    c.load_taxonomy(w.parent)
    rng = np.random.default_rng(3)
    reads = synth.simulate_reads(rng, w.genomes, 3000)
    reads[5] = reads[5][:10]                                   # shorter than k
    reads[6] = reads[6][:0]
    doc = fastq_of(reads, wrap={"fastq": 0, "fasta_wrapped": 61, "fastq_wrapped": 50}[form], fasta=form == "fasta_wrapped")
    bases, offsets = synth.concat(reads)
    exp = c.classify_runs(bases, offsets)
    for dbg in (0, 0x4000, 0x4040):
        c.debug_set(dbg)
        got = c.classify_text(doc, final=True, trim_readno=True, want_runs=True)
        assert got["status"] == _lib.TEXT_OK and got["n_records"] == len(reads) and got["consumed"][0] == len(doc)
        assert dbg == 0 or got["n_slices"] > 10
        # many slices per classify launch: the records of every 8 KiB piece are appended to one packed image
        assert got["n_launches"] == 1 if dbg != 0x4040 else got["n_launches"] > 10
        for k in ("taxon", "missing", "ambig", "n_hits"):
            assert np.array_equal(got[k], exp[k]), (form, dbg, k)
        for u in range(len(reads)):
            assert np.array_equal(got["runs"][u][0], exp["runs"][u][0]) and np.array_equal(got["runs"][u][1], exp["runs"][u][1]), u
        assert got["names"] == [b"read%d" % i for i in range(len(reads))]
        # the caller's own run arrays: the same runs; arrays too small for all of them end the call early at a slice boundary
        # (BNS_TEXT_CAP, nothing half-delivered) and the next call goes on from consumed[]
        got = c.classify_text(doc, final=True, trim_readno=True, want_runs=True, runs_cap=len(doc))
        assert got["status"] == _lib.TEXT_OK and got["n_records"] == len(reads)
        for u in range(len(reads)):
            assert np.array_equal(got["runs"][u][0], exp["runs"][u][0]) and np.array_equal(got["runs"][u][1], exp["runs"][u][1]), u
        if dbg:
            pos, n_done, cap, calls = 0, 0, 400, 0
            while pos < len(doc):
                part = c.classify_text(doc[pos:], final=True, trim_readno=True, want_runs=True, runs_cap=cap)
                calls += 1
                assert part["status"] in (_lib.TEXT_OK, _lib.TEXT_CAP)
                for i in range(part["n_records"]):
                    assert np.array_equal(part["runs"][i][0], exp["runs"][n_done + i][0]) and np.array_equal(part["runs"][i][1], exp["runs"][n_done + i][1])
                assert np.array_equal(part["taxon"], exp["taxon"][n_done:n_done + part["n_records"]])
                n_done += part["n_records"]; pos += part["consumed"][0]
                if part["n_records"] == 0:
                    cap *= 2
                assert calls < 200
            assert n_done == len(reads) and calls > 3
    c.debug_set(0)
    assert int((exp["taxon"] != 0).sum()) > 2000
    # the text already in HBM (what a device-side inflate leaves)
    ptr = c.dev_alloc(len(doc) + 256)
    c.dev_upload(ptr, np.frombuffer(doc, dtype=np.uint8))
    got = c.classify_text([], final=True, trim_readno=True, device_ptrs=[(ptr, len(doc))])
    for k in ("taxon", "missing", "ambig", "n_hits"):
        assert np.array_equal(got[k], exp[k]), k
    c.dev_free(ptr)
    c.close()


def test_classify_text_in_two_halves(world):
    """BNS_TEXT_DEFER + bns_text_finish: the first half says what the records are (count, where the call stopped), the second fills
    the arrays -- everything equal to the call in one piece; one slice, many 8 KiB slices, many batches; run arrays that are too small
    are reported by the second half; the context takes no other call in between"""
    w = world
    c = bonsai_amd.Context(0)
    c.set_encoder(31, None, canonicalize=True)
    c.load_table(w.n_buckets, w.flags, w.keys, w.vals)
    c.load_taxonomy(w.parent)
    rng = np.random.default_rng(5)
    reads = synth.simulate_reads(rng, w.genomes, 2500)
    doc = fastq_of(reads)
    cut = doc[:len(doc) - 37]                                  # ends inside the last record
    for dbg in (0, 0x4000, 0x4040):
        c.debug_set(dbg)
        for text, final in ((doc, True), (cut, False)):
            one = c.classify_text(text, final=final, trim_readno=True, want_runs=True)
            def busy():
                with pytest.raises(Exception):
                    c.classify_text(text, final=final, parse_only=True)
            two = c.classify_text(text, final=final, trim_readno=True, want_runs=True, defer=True, between=busy)
            fh = two["first_half"]
            assert fh["n_records"] == one["n_records"] == two["n_records"] and fh["consumed"] == one["consumed"] == two["consumed"]
            assert fh["status"] == one["status"] == two["status"] == _lib.TEXT_OK and fh["total_bases"] == one["total_bases"]
            for k in ("taxon", "missing", "ambig", "n_hits", "seq_len", "rec_pos"):
                assert np.array_equal(one[k], two[k]), (dbg, k)
            assert one["names"] == two["names"]
            for u in range(one["n_records"]):
                assert np.array_equal(one["runs"][u][0], two["runs"][u][0]) and np.array_equal(one["runs"][u][1], two["runs"][u][1])
            assert two["n_records"] == (len(reads) if final else len(reads) - 1)
        # the caller's run arrays too small: the second half says so, and fewer records than the first half promised are his
        two = c.classify_text(doc, final=True, trim_readno=True, want_runs=True, runs_cap=50, defer=True)
        assert two["status"] == _lib.TEXT_CAP and two["n_records"] < len(reads)
        assert dbg == 0x4040 or two["first_half"]["n_records"] == len(reads)       # (many batches: all but the last are classified in the first half, which then sees the overflow itself)
        # the device text form, and a pair
        two = c.classify_text([doc, doc], final=True, trim_readno=True, defer=True)
        one = c.classify_text([doc, doc], final=True, trim_readno=True)
        assert two["first_half"]["n_records"] == 2 * len(reads) and np.array_equal(one["taxon"], two["taxon"])
    c.debug_set(0)
    with pytest.raises(Exception):
        c._chk(c.L.bns_text_finish(c.h, __import__("ctypes").byref(_lib.TextInfo())), "bns_text_finish")      # nothing waits
    c.close()


def test_classify_text_pair_of_files(world):
    """two files, mates by record index (kseq_declare.h:116-131): one vote per pair as bns_classify_batch(paired=1); files whose
    records differ in size consume different numbers of bytes; the shorter file ends the pairing"""
    w = world
    c = bonsai_amd.Context(0)
    c.set_encoder(31, None, canonicalize=True)
    c.load_table(w.n_buckets, w.flags, w.keys, w.vals)
    c.load_taxonomy(w.parent)
    rng = np.random.default_rng(4)
    r1 = synth.simulate_reads(rng, w.genomes, 2000)
    r2 = [r[:int(rng.integers(40, len(r) + 1))] for r in synth.simulate_reads(rng, w.genomes, 2000)]
    inter = [x for p in zip(r1, r2) for x in p]
    bases, offsets = synth.concat(inter)
    exp = c.classify_runs(bases, offsets, paired=True)
    d1 = fastq_of(r1, names=[b"p%d/1" % i for i in range(len(r1))])
    d2 = fastq_of(r2, names=[b"p%d/2" % i for i in range(len(r2))], wrap=70, fasta=True)
    for dbg in (0, 0x4000, 0x4040):
        c.debug_set(dbg)
        got = c.classify_text([d1, d2], final=True, trim_readno=True, want_runs=True)
        assert got["n_launches"] == (1 if dbg != 0x4040 else got["n_launches"]) and (dbg != 0x4040 or got["n_launches"] > 10)
        assert got["status"] == _lib.TEXT_OK and got["n_records"] == 2 * len(r1) and got["consumed"] == [len(d1), len(d2)]
        for k in ("taxon", "missing", "ambig", "n_hits"):
            assert np.array_equal(got[k], exp[k]), (dbg, k)
        for u in range(len(r1)):
            assert np.array_equal(got["runs"][u][0], exp["runs"][u][0]) and np.array_equal(got["runs"][u][1], exp["runs"][u][1])
        assert got["names"] == [b"p%d" % (i // 2) for i in range(2 * len(r1))]
        assert got["seq_len"].tolist() == [len(x) for x in inter]
    c.debug_set(0)
    # the second file is shorter: pairs up to its last record; the first file's surplus is not consumed
    cut = d2[:d2.index(b">p1500/2")]
    got = c.classify_text([d1, cut], final=True, trim_readno=True)
    assert got["n_records"] == 3000 and got["consumed"][1] == len(cut) and got["consumed"][0] == d1.index(b"@p1500/1")
    assert np.array_equal(got["taxon"], exp["taxon"][:1500])
    c.close()


def test_classify_text_argument_errors(ctx):
    with pytest.raises(bonsai_amd.BonsaiAmdError):
        ctx.classify_text(b"@a\nAC\n+\nII\n")                      # no table loaded
    res = ctx.classify_text(b"@a\nAC\n+\nII\n@b\nAC\n+\nII\n", parse_only=True, cap_records=1)
    assert res["status"] == _lib.TEXT_CAP and res["n_records"] == 0


@pytest.mark.parametrize("via_host", [False, True])
def test_dev_copy_between_contexts(via_host):
    """bns_dev_copy_peer: what one context's block left unfinished goes in front of the next context's text -- device to device, and
    (BNS_PEER_VIA_HOST: a process of its own, the switch is read once) through page-locked host memory as between two devices"""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, bonsai_amd
        a, b = bonsai_amd.Context(0), bonsai_amd.Context(0)
        rng = np.random.default_rng(1)
        for n in (1, 317, 65536, 3 << 20):
            src = rng.integers(0, 256, size=n + 11, dtype=np.uint8)
            pa, pb = a.dev_alloc(n + 64), b.dev_alloc(n + 64)
            a.dev_upload(pa, src)
            b.dev_upload(pb, np.zeros(n + 64, dtype=np.uint8))
            b.dev_copy_from(pb + 5, a, pa + 3, n)              # (unaligned on both sides)
            out = np.zeros(n + 64, dtype=np.uint8)
            b.dev_download(pb, out)
            assert np.array_equal(out[5:5 + n], src[3:3 + n]) and not out[:5].any() and not out[5 + n:].any(), n
            a.dev_free(pa); b.dev_free(pb)
        b.dev_copy_from(0, a, 0, 0)                            # nothing to copy: fine
        print("copied")
    """)
    env = dict(os.environ)
    env.pop("BNS_PEER_VIA_HOST", None)
    if via_host:
        env["BNS_PEER_VIA_HOST"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, timeout=300)
    assert p.returncode == 0 and b"copied" in p.stdout, p.stderr.decode()[-2000:]
