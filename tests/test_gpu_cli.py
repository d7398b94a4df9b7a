"""End-to-end: the `bonsai classify` drop-in binary (C++ host + C ABI + HIP) against the oracle's lines."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bonsai_amd", "bin", "bonsai")


@pytest.fixture(scope="module")
def files(oracle, small_world, tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    w = small_world
    db = str(d / "bns.db")
    oracle.db_write(db, 31, 31, None, w.table, spacing_width=1)
    nodes = str(d / "nodes.dmp")
    synth.write_nodes_dmp(nodes)
    rng = np.random.default_rng(77)
    reads = synth.simulate_reads(rng, w.genomes, 600, var_len=True, n_rate=0.003)
    r1 = str(d / "r1.fq"); r2 = str(d / "r2.fq.gz")
    with open(r1, "wb") as f:
        for i, r in enumerate(reads[:300]):
            f.write(b"@read%d/1 some comment\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size))
    with gzip.open(r2, "wb") as f:
        for i, r in enumerate(reads[300:]):
            f.write(b"@read%d/2\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size))
    fa = str(d / "multi.fa")
    with open(fa, "wb") as f:
        for i, r in enumerate(reads[:50]):
            s = r.tobytes()
            f.write(b">fa%d\n" % i + b"\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + b"\n")
    return {"db": db, "nodes": nodes, "r1": r1, "r2": r2, "fa": fa, "reads": reads, "w": w}


def run(args):
    assert os.path.exists(BIN), "bonsai CLI not built (run __graft_entry__.build())"
    p = subprocess.run([BIN, "classify"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    assert b"Successfully completed classify!" in p.stderr
    return p.stdout


def expected_lines(oracle, w, names, reads1, reads2=None, emit_all=False):
    out = []
    for i, r in enumerate(reads1):
        s2 = reads2[i].tobytes() if reads2 is not None else None
        t, m, a, hits = oracle.classify_seq(w.table, w.tax, 31, r.tobytes(), s2)
        if t or emit_all:
            out.append(oracle.kraken_line(names[i], t, r.size, m, a, hits))
    return b"".join(out)


@pytest.mark.parametrize("layout", ["minbucket", "bucket", "khash"])
def test_cli_single_end(oracle, files, layout):
    w, reads = files["w"], files["reads"]
    got = run(["-a", "-L", layout, files["db"], files["nodes"], files["r1"]])
    exp = expected_lines(oracle, w, ["read%d" % i for i in range(300)], reads[:300], emit_all=True)
    assert got == exp
    got2 = run(["-c", "5000", files["db"], files["nodes"], files["r1"]])          # many small batches, classified only
    assert got2 == expected_lines(oracle, w, ["read%d" % i for i in range(300)], reads[:300])


def test_cli_parallel_parse_same_output(oracle, files, tmp_path):
    """-P 2 / 3 with short stretches: the file is parsed side by side in pieces, the output is byte for byte that of -P 1"""
    w, reads = files["w"], files["reads"]
    big = str(tmp_path / "many.fq")
    with open(big, "wb") as f:
        for rep in range(12):
            for i, r in enumerate(reads[:300]):
                f.write(b"@m%d_%d/1 c\n%s\n+\n%s\n" % (rep, i, r.tobytes(), (b"@>+I" * r.size)[:r.size]))
    one = run(["-a", "-P", "1", files["db"], files["nodes"], big])
    exp = expected_lines(oracle, w, ["m0_%d" % i for i in range(300)], reads[:300], emit_all=True)
    assert one[:len(exp)] == exp
    for spec, chunk in (("2:65536", "20000"), ("3:100000", "5000"), ("2:200000", str(1 << 24))):
        # (BNS_TEXT_GPU=0: this test is about the HOST parser's stretches; the device's text parser is tests/test_gpu_cli_text.py)
        p = subprocess.run([BIN, "classify", "-a", "-P", spec, "-c", chunk, files["db"], files["nodes"], big], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, BNS_CLI_TIMING="1", BNS_TEXT_GPU="0"))
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout == one, spec
        assert b"stretches)" in p.stderr, p.stderr.decode()          # (it really was split)
    # a pair of plain files: one parser thread per file under -P 2, mates interleaved as -P 1 interleaves them
    m2 = str(tmp_path / "many_2.fq")
    with open(m2, "wb") as f:
        for rep in range(12):
            for i, r in enumerate(reads[300:600]):
                f.write(b"@m%d_%d/2\n%s\n+\n%s\n" % (rep, i, r.tobytes(), b"I" * r.size))
    pair_one = run(["-a", "-P", "1", files["db"], files["nodes"], big, m2])
    for chunk in ("20000", str(1 << 24)):
        assert run(["-a", "-P", "2", "-c", chunk, files["db"], files["nodes"], big, m2]) == pair_one, chunk
    assert pair_one.count(b"\n") == 3600
    # the fasta file, split as well
    fa_one = run(["-a", "-P", "1", files["db"], files["nodes"], files["fa"]])
    assert run(["-a", "-P", "2:2000", files["db"], files["nodes"], files["fa"]]) == fa_one


def test_cli_paired_gz_and_fasta(oracle, files):
    w, reads = files["w"], files["reads"]
    got = run(["-a", files["db"], files["nodes"], files["r1"], files["r2"]])
    exp = expected_lines(oracle, w, ["read%d" % i for i in range(300)], reads[:300], reads[300:], emit_all=True)
    assert got == exp
    got = run(["-a", files["db"], files["nodes"], files["fa"]])
    assert got == expected_lines(oracle, w, ["fa%d" % i for i in range(50)], reads[:50], emit_all=True)


def test_cli_output_file_and_errors(files, tmp_path):
    out = tmp_path / "o.txt"
    assert run(["-o", str(out), files["db"], files["nodes"], files["r1"]]) == b""
    assert out.stat().st_size > 0
    p = subprocess.run([BIN, "classify", str(tmp_path / "nope.db"), files["nodes"], files["r1"]], stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"Could not" in p.stderr


def test_encoder_cpp_api(gpu_ctx, oracle):
    """bns::Encoder::for_each is exercised through the CLI build; here the same C-ABI call it makes."""
    seq = synth.rand_seq(np.random.default_rng(2), 500)
    b, o = synth.concat([seq])
    gpu_ctx.set_encoder(31, None, canonicalize=True, spaced_intended=False)
    assert np.array_equal(gpu_ctx.encode(b, o)[0], oracle.encode(seq.tobytes(), 31))


@pytest.mark.parametrize("name", ["HiSeq", "MiSeq"])
def test_cli_real_reads(oracle, name, tmp_path):
    """Real Illumina reads (first 300 records of the reference's kraken_benchmarks/*_accuracy.fa, multi-line FASTA,
    92..251 bp): db built from every other read under a per-species taxonomy, all reads classified by the CLI."""
    fa = os.path.join(ROOT, "tests", "golden", "%s_accuracy_300.fa" % name)
    recs = oracle.read_fasta(fa)
    species = sorted({n.rsplit("_%s" % name, 1)[0] for n, _ in recs})
    # taxonomy: root 1 -> genus-like groups of 3 species (10+g) -> species (100+i)
    pairs = [(1, 1)] + [(10 + g, 1) for g in range((len(species) + 2) // 3)] + [(100 + i, 10 + i // 3) for i in range(len(species))]
    tax = oracle.Taxonomy(pairs=pairs)
    table = oracle.Table()
    for i, (n, sq) in enumerate(recs):
        if i % 2 == 0:
            oracle.lca_map_add(table, tax, 31, sq, 100 + species.index(n.rsplit("_%s" % name, 1)[0]))
    db = str(tmp_path / "real.db.gz")
    oracle.db_write(db, 31, 31, None, table, spacing_width=2)         # the width the reference's gz writer emits
    nodes = str(tmp_path / "nodes.dmp")
    synth.write_nodes_dmp(nodes, pairs)
    got = run(["-a", db, nodes, fa])
    exp = []
    n_class = 0
    for n, sq in recs:
        t, m, a, hits = oracle.classify_seq(table, tax, 31, sq)
        n_class += t != 0
        exp.append(oracle.kraken_line(n, t, len(sq), m, a, hits))
    assert got == b"".join(exp)
    assert n_class >= len(recs) // 2 - 10                              # (nearly) every db read classifies itself


@pytest.mark.parametrize("flags,w,score", [([], 31, 0), (["-w", "50", "-e"], 50, 1), (["-w", "40", "-z"], 40, 0),
                                           (["-w", "50", "-t"], 50, 1),                         # -t / -f = the entropy build (bin/bonsai.cpp:228)
                                           (["-S", "1x15,0x15", "-w", "50", "-e"], 50, 1),      # BASELINE configs[2]: spaced, w = 50
                                           (["-C", "-w", "50", "-e"], 50, 1), (["-C"], 31, 0)])  # forward k-mers only
def test_cli_build_then_classify(oracle, small_world, tmp_path, flags, w, score):
    """`bonsai build` (lca_map on the GPU) -> bns.db -> `bonsai classify`, both against the oracle."""
    from bonsai_amd import hostio
    wld = small_world
    nodes = str(tmp_path / "nodes.dmp")
    synth.write_nodes_dmp(nodes)
    names = str(tmp_path / "nameidmap.txt")
    paths = []
    with open(names, "w") as nf:
        for i, (leaf, g) in enumerate(wld.genomes.items()):
            acc = "NC_%06d.1" % i
            nf.write("%s\t%d\n" % (acc, leaf))
            s = g.tobytes()
            half = len(s) // 2
            body = (">%s contig one\n" % acc).encode() + b"\n".join(s[:half][j:j + 70] for j in range(0, half, 70)) + \
                   (b"\n>%s_2 contig two\n" % acc.encode()) + s[half:] + b"\n"
            p = str(tmp_path / ("g%d.fna%s" % (i, ".gz" if i % 2 else "")))
            (gzip.open(p, "wb") if i % 2 else open(p, "wb")).write(body)
            paths.append(p)
    out = str(tmp_path / "built.db")
    pr = subprocess.run([BIN, "build", "-k", "31", "-T", nodes, "-M", names] + flags + [out, "unused"] + paths,
                        stderr=subprocess.PIPE, timeout=300)
    assert pr.returncode == 0, pr.stderr.decode()
    if "-z" in flags:
        out += ".gz"
    d = hostio.read_db(out)
    gaps = [1] * 15 + [0] * 15 if "-S" in flags else None
    comb = 31 + (sum(gaps) if gaps else 0)
    assert (d["k"], d["w"]) == (31, max(w, comb)) and d["gaps"].tolist() == (gaps or [0] * 30)
    assert d["upper_bound"] == int(d["n_buckets"] * 0.77 + 0.5) and d["size"] <= d["upper_bound"]
    assert d["n_buckets"] < 4 or int((d["n_buckets"] // 2) * 0.77 + 0.5) <= d["size"]      # as compact as khash grows it
    # expected map: two contigs per genome, each its own sequence (k-mers do not span the contig break)
    canon = "-C" not in flags
    exp_t = oracle.Table()
    for leaf, g in wld.genomes.items():
        s = g.tobytes()
        for part in (s[:len(s) // 2], s[len(s) // 2:]):
            if w > comb:
                oracle.lca_map_add_windowed(exp_t, wld.tax, 31, w, score, part, leaf, gaps=gaps, canon=canon)
            else:
                oracle.lca_map_add(exp_t, wld.tax, 31, part, leaf, gaps=gaps, canon=canon)
    ef, ek, ev = exp_t.arrays()
    i = np.arange(exp_t.n_buckets)
    m = ((ef[i >> 4] >> ((i & 15) << 1)) & 3) == 0
    exp = dict(zip(ek[m].tolist(), ev[m].tolist()))
    i = np.arange(d["n_buckets"])
    m = ((d["flags"][i >> 4] >> ((i & 15) << 1)) & 3) == 0
    got = dict(zip(d["keys"][m].tolist(), d["vals"][m].tolist()))
    assert got == exp
    # classify with the built file
    rng = np.random.default_rng(5)
    reads = synth.simulate_reads(rng, wld.genomes, 200)
    fq = str(tmp_path / "r.fq")
    with open(fq, "wb") as f:
        for j, r in enumerate(reads):
            f.write(b"@q%d\n%s\n+\n%s\n" % (j, r.tobytes(), b"I" * r.size))
    got_out = run(["-a"] + ([] if canon else ["-C"]) + [out, nodes, fq])
    lines = []
    for j, r in enumerate(reads):
        t, mm, a, hits = oracle.classify_seq(exp_t, wld.tax, 31, r.tobytes(), gaps=gaps, canon=canon, spaced_intended=True)
        lines.append(oracle.kraken_line("q%d" % j, t, r.size, mm, a, hits))
    assert got_out == b"".join(lines)


def test_cli_build_errors(tmp_path):
    p = subprocess.run([BIN, "build", "-k", "31", "out.db", "x", "nofile.fna"], stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"seq2taxpath required" in p.stderr
    # -t / -f: the reference's `LEX == mode || score_scheme::ENTROPY` is always true, so these flags run the entropy lca build
    p = subprocess.run([BIN, "build", "-t", "-k", "31", "out.db", "x", "nofile.fna"], stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"entropy-minimized lca map" in p.stderr and b"seq2taxpath required" in p.stderr


def test_cli_pack_container_same_output(oracle, files, tmp_path):
    """`bonsai pack` + `bonsai classify <container>`: Kraken lines byte for byte those of the FASTQ itself -- single-end, a pair of
    files (one of them gzip), small chunks (several per container) -- and the -b taxon file"""
    for ins, tag in (([files["r1"]], "se"), ([files["r1"], files["r2"]], "pe"), ([files["fa"]], "fa")):
        pk = str(tmp_path / ("reads_%s.bnsp" % tag))
        p = subprocess.run([BIN, "pack", "-o", pk, "-c", "20000"] + ins, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode == 0, p.stderr.decode()
        want = run(["-a", files["db"], files["nodes"]] + ins)
        tb = str(tmp_path / ("tax_%s.bin" % tag))
        got = run(["-a", "-b", tb, files["db"], files["nodes"], pk])
        assert got == want and want.count(b"\n") == (300 if tag != "fa" else 50)
        tax = np.fromfile(tb, dtype="<u4")
        assert tax.tolist() == [int(l.split(b"\t")[2]) for l in want.splitlines()]
        # no Kraken lines (-K): just the tally, same as for the FASTQ
        p1 = subprocess.run([BIN, "classify", "-K", files["db"], files["nodes"], pk], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        p2 = subprocess.run([BIN, "classify", "-K", files["db"], files["nodes"]] + ins, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        tally = lambda e: [l for l in e.decode().splitlines() if l.startswith("Classified")]
        assert p1.returncode == 0 and tally(p1.stderr) == tally(p2.stderr) and tally(p1.stderr)
    # FASTQ-style output needs the bases: refused with a message, not garbage
    p = subprocess.run([BIN, "classify", "-f", files["db"], files["nodes"], pk], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"container" in p.stderr


def test_cli_bgzf_on_cpu_gpu_and_in_stretches(oracle, files, tmp_path):
    """BGZF input through the CLI: CPU inflaters, the device inflating beside them and alone (BNS_BGZF_GPU=1), one parser and
    stretches of the inflated text on two -- Kraken lines byte for byte those of the plain file, single-end and as one file of a pair"""
    reads = files["reads"]
    doc = bytearray()
    for rep in range(40):
        for i, r in enumerate(reads[:300]):
            doc += b"@m%d_%d/1 c\n%s\n+\n%s\n" % (rep, i, r.tobytes(), (b"@>+I" * r.size)[:r.size])
    plain = str(tmp_path / "many.fq")
    open(plain, "wb").write(bytes(doc))
    bg = str(tmp_path / "many.fq.gz")
    synth.write_bgzf(bg, bytes(doc), member_sizes=[65280, 30000, 1000, 65280, 7])
    one = run(["-a", "-P", "1", files["db"], files["nodes"], plain])
    assert one.count(b"\n") == 12000
    for env in ({}, {"BNS_BGZF_ONE_PARSER": "1"}, {"BNS_BGZF_GPU": "1", "BNS_BGZF_GPU_BATCH": "1"}, {"BNS_BGZF_GPU": "1", "BNS_GZ_THREADS": "0", "BNS_BGZF_GPU_BATCH": "1"},
                {"BNS_BGZF_GPU": "1", "BNS_GZ_THREADS": "0"}):
        for extra in ([], ["-P", "2:1"], ["-c", "20000"]):
            # (BNS_TEXT_GPU=0: the HOST reader's BGZF paths; the members' text parsed on the device is tests/test_gpu_cli_text.py)
            p = subprocess.run([BIN, "classify", "-a"] + extra + [files["db"], files["nodes"], bg], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                               env=dict(os.environ, BNS_CLI_TIMING="1", BNS_TEXT_GPU="0", **env))
            assert p.returncode == 0, p.stderr.decode()
            assert p.stdout == one, (env, extra)
            if env.get("BNS_GZ_THREADS") == "0":
                assert b"BGZF on the GPU" in p.stderr and b" 0 batches" not in p.stderr
    # as the first file of a pair (the second one plain): mates interleaved as for two plain files
    m2 = str(tmp_path / "many_2.fq")
    with open(m2, "wb") as f:
        for rep in range(40):
            for i, r in enumerate(reads[300:600]):
                f.write(b"@m%d_%d/2\n%s\n+\n%s\n" % (rep, i, r.tobytes(), b"I" * r.size))
    pair_one = run(["-a", "-P", "1", files["db"], files["nodes"], plain, m2])
    p = subprocess.run([BIN, "classify", "-a", files["db"], files["nodes"], bg, m2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, BNS_BGZF_GPU="1"))
    assert p.returncode == 0 and p.stdout == pair_one
