"""`bonsai classify` on a plain FASTA / FASTQ file with the text parsed on the device (process_text_gpu -> bns_classify_text):
stdout byte for byte that of the host-parser path (BNS_TEXT_GPU=0) and of the oracle's lines -- one block and many, one context and
several on the device (guessed block starts, checked and classified again when wrong), CRLF text and wrapped quality (round 6: on the
device), text the kernels hand back (stray text, quality of the wrong length: the host parser takes over at a record boundary), FASTA, -K, -b."""
import os
import subprocess

import numpy as np
import pytest

import ingest_fuzz
import synth
from test_gpu_cli import BIN, expected_lines, files  # noqa: F401  (the module's fixture)

pytestmark = pytest.mark.gpu


def cli(args, **env):
    e = dict(os.environ, BNS_CLI_TIMING="1")
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([BIN, "classify"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout, p.stderr.decode()


def test_text_path_is_the_default_and_matches_the_oracle(oracle, files):
    w, reads = files["w"], files["reads"]
    exp = expected_lines(oracle, w, ["read%d" % i for i in range(300)], reads[:300], emit_all=True)
    out, err = cli(["-a", files["db"], files["nodes"], files["r1"]])
    assert "text on the device" in err and "host parser takes the rest" not in err
    assert out == exp
    host, err = cli(["-a", files["db"], files["nodes"], files["r1"]], BNS_TEXT_GPU=0)
    assert "text on the device" not in err and host == exp
    # classified only; wrapped FASTA
    out, _ = cli([files["db"], files["nodes"], files["r1"]])
    assert out == expected_lines(oracle, w, ["read%d" % i for i in range(300)], reads[:300])
    out, err = cli(["-a", files["db"], files["nodes"], files["fa"]])
    assert "text on the device" in err
    assert out == expected_lines(oracle, w, ["fa%d" % i for i in range(50)], reads[:50], emit_all=True)


@pytest.fixture(scope="module")
def big_file(files, tmp_path_factory):
    d = tmp_path_factory.mktemp("clitext")
    reads = files["reads"]
    big = str(d / "many.fq")
    with open(big, "wb") as f:
        for rep in range(12):
            for i, r in enumerate(reads[:300]):
                f.write(b"@m%d_%d/1 c\n%s\n+\n%s\n" % (rep, i, r.tobytes(), (b"@>+I" * r.size)[:r.size]))
    return big


@pytest.mark.parametrize("block", [700, 3000, 50000, 1 << 20])
@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
def test_blocks_and_contexts(files, big_file, block, devices):
    """many blocks; several contexts take blocks side by side, guessing where their first record starts (quality lines of this
    file start with '@', '>' and '+'): output is that of the host parser whatever the block size and the number of contexts"""
    host, _ = cli(["-a", files["db"], files["nodes"], big_file], BNS_TEXT_GPU=0)
    out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], big_file], BNS_TEXT_BLOCK_BYTES=block)
    assert "text on the device" in err and "host parser takes the rest" not in err, err
    assert out == host and out.count(b"\n") == 3600
    if devices != "0" and block < 50000:
        assert " 0 guessed starts" not in err                    # (it did run blocks side by side)


def test_handover_to_the_host_parser(files, tmp_path):
    """text the kernels do not take ends the device path at a record boundary; the host parser reads on from there"""
    w, reads = files["w"], files["reads"]
    good = b"".join(b"@g%d\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[:200]))
    tails = {
        "crlf": b"".join(b"@c%d\r\n%s\r\n+\r\n%s\r\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[200:260])),
        "wrapped_quality": b"".join(b"@q%d\n%s\n+\n%s\n%s\n" % (i, r.tobytes(), b"@" * (r.size // 2), b"+" * (r.size - r.size // 2)) for i, r in enumerate(reads[200:260])),
        "stray": b"stray text\n" + b"".join(b"@s%d\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[200:260])),
    }
    tails["long_quality"] = b"".join(b"@l%d\n%s\n+\n%sI\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[200:260]))
    for tag, tail in tails.items():
        p = str(tmp_path / (tag + ".fq"))
        open(p, "wb").write(good + tail + good)
        host, _ = cli(["-a", files["db"], files["nodes"], p], BNS_TEXT_GPU=0)
        for block in (5000, 1 << 22):
            out, err = cli(["-a", files["db"], files["nodes"], p], BNS_TEXT_BLOCK_BYTES=block)
            # round 6: CRLF line ends and quality over several lines are read on the device (klib/kseq.h:135, :217)
            assert ("host parser takes the rest" in err) == (tag in ("stray", "long_quality")), (tag, err)
            assert out == host and out.count(b"\n") >= (200 if tag == "long_quality" else 460), (tag, block)    # (a quality string that is too long: kseq's error -2 ends the reading)
    # a CRLF file, and a file that is irregular from its first byte
    for tag, text, handed in (("all_crlf", tails["crlf"], False), ("all_stray", tails["stray"], True)):
        p = str(tmp_path / (tag + ".fq"))
        open(p, "wb").write(text)
        host, _ = cli(["-a", files["db"], files["nodes"], p], BNS_TEXT_GPU=0)
        out, err = cli(["-a", files["db"], files["nodes"], p])
        assert out == host and out.count(b"\n") == 60, tag
        assert ("text on the device" in err and "host parser takes the rest" not in err) == (not handed), tag      # (a file that does not start like FASTA / FASTQ never takes the device path)


def test_fuzzed_files(files, tmp_path):
    """random regular and wild text: device path + handover == host parser, byte for byte"""
    rng = np.random.default_rng(12)
    for it in range(12):
        doc = ingest_fuzz.make_doc(rng, int(rng.integers(20, 400)), wild=(0.0 if it % 2 else 0.3), final_newline=bool(it % 3))
        p = str(tmp_path / ("fz%d.txt" % it))
        open(p, "wb").write(doc)
        host, _ = cli(["-a", files["db"], files["nodes"], p], BNS_TEXT_GPU=0)
        for block, dev in ((1 << 22, "0"), (4000, "0"), (2500, "0,0")):
            out, err = cli(["-a", "-g", dev, files["db"], files["nodes"], p], BNS_TEXT_BLOCK_BYTES=block)
            assert out == host, (it, block, dev)


def test_no_lines_and_taxon_file(files, big_file, tmp_path):
    """-K (no Kraken lines: the taxon alone comes back) with -b: the raw taxon per read, and the tally on stderr"""
    b1, b2 = str(tmp_path / "t1.bin"), str(tmp_path / "t2.bin")
    out, err = cli(["-K", "-b", b1, files["db"], files["nodes"], big_file], BNS_TEXT_BLOCK_BYTES=40000)
    host, herr = cli(["-K", "-b", b2, files["db"], files["nodes"], big_file], BNS_TEXT_GPU=0)
    assert out == host == b""
    t1, t2 = np.fromfile(b1, dtype=np.uint32), np.fromfile(b2, dtype=np.uint32)
    assert t1.size == 3600 and np.array_equal(t1, t2)
    tally = [l for l in err.splitlines() if "lassified" in l and "timing" not in l]
    assert tally == [l for l in herr.splitlines() if "lassified" in l and "timing" not in l]


def test_bgzf_text_stays_on_the_device(files, big_file, tmp_path):
    """a BGZF file: members inflated on the device, their text parsed and classified where it lies -- output byte for byte that of
    the plain file through the host parser; batches of a few members (records straddle every batch boundary), members of every
    size, wrapped FASTA, -K with -b, and text the kernels hand back (the host parser reads the file and leaves out what was printed)"""
    doc = open(big_file, "rb").read()
    host, _ = cli(["-a", files["db"], files["nodes"], big_file], BNS_TEXT_GPU=0)
    for tag, kw in (("std", {}), ("odd", {"member_sizes": [65280, 30000, 1000, 65280, 7]})):
        bg = str(tmp_path / ("many_%s.fq.gz" % tag))
        synth.write_bgzf(bg, doc, **kw)
        for members in (16384, 3, 1):
            out, err = cli(["-a", files["db"], files["nodes"], bg], BNS_BGZF_BATCH_MEMBERS=members)
            assert "BGZF text on the device" in err and "host parser takes the rest" not in err, err
            assert out == host, (tag, members)
        # round 6: the batches over several contexts in turn (batch b inflated and classified on device b % G, cut in file order); what a
        # batch leaves unfinished goes to the next context's buffer -- device to device, and (BNS_PEER_VIA_HOST) through host memory as
        # between two devices
        for devices, members, env in (("0,0", 3, {}), ("0,0,0", 1, {}), ("0,0", 2, {"BNS_PEER_VIA_HOST": 1}), ("0,0,0", 16384, {"BNS_PEER_VIA_HOST": 1})):
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], bg], BNS_BGZF_BATCH_MEMBERS=members, **env)
            assert "BGZF text on the device" in err and "on %d device(s)" % len(devices.split(",")) in err and "host parser takes the rest" not in err, err
            assert out == host, (tag, devices, members)
    b1, b2 = str(tmp_path / "t1.bin"), str(tmp_path / "t2.bin")
    out, err = cli(["-K", "-b", b1, files["db"], files["nodes"], bg], BNS_BGZF_BATCH_MEMBERS=5)
    cli(["-K", "-b", b2, files["db"], files["nodes"], big_file], BNS_TEXT_GPU=0)
    assert out == b"" and np.array_equal(np.fromfile(b1, dtype=np.uint32), np.fromfile(b2, dtype=np.uint32))
    # wrapped FASTA
    fa_host, _ = cli(["-a", files["db"], files["nodes"], files["fa"]], BNS_TEXT_GPU=0)
    fbg = str(tmp_path / "multi.fa.gz")
    synth.write_bgzf(fbg, open(files["fa"], "rb").read(), member_sizes=[500, 70, 3000])
    out, err = cli(["-a", files["db"], files["nodes"], fbg], BNS_BGZF_BATCH_MEMBERS=2)
    assert "BGZF text on the device" in err and out == fa_host
    # text the kernels do not take, in the middle of the file and from its first byte
    reads = files["reads"]
    good = b"".join(b"@g%d\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[:200]))
    crlf = b"".join(b"@c%d\r\n%s\r\n+\r\n%s\r\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[200:260]))
    stray = b"stray text\n" + b"".join(b"@s%d\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[200:260]))
    for tag, text, handed in (("mid", good + stray + good, True), ("all", stray, True), ("mid_crlf", good + crlf + good, False), ("all_crlf", crlf, False)):
        plain = str(tmp_path / (tag + ".fq")); open(plain, "wb").write(text)
        bgz = str(tmp_path / (tag + ".fq.gz")); synth.write_bgzf(bgz, text, member_sizes=[4000, 900])
        want, _ = cli(["-a", files["db"], files["nodes"], plain], BNS_TEXT_GPU=0)
        for members, devices in ((16384, "0"), (4, "0"), (4, "0,0,0")):
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], bgz], BNS_BGZF_BATCH_MEMBERS=members)
            assert ("host parser takes the rest" in err) == handed and out == want, (tag, members, devices)      # (round 6: CRLF text stays on the device)


def test_gzip_text_stays_on_the_device(files, big_file, tmp_path):
    """round 6: ONE plain gzip stream (what `gzip` writes: no BGZF members) entered at block headers found on the device
    (bns_inflate_stream_device), inflated, parsed and classified where the text lies -- output byte for byte that of the plain file through
    the host parser: every level, calls of a few dozen KB (a call ends at a block boundary, the next goes on behind it), little room for
    text, several members, wrapped FASTA, -K -b; text the kernels hand back and streams the device decoder refuses go to the host reader"""
    import gzip
    doc = open(big_file, "rb").read()
    host, _ = cli(["-a", files["db"], files["nodes"], big_file], BNS_TEXT_GPU=0)
    for level in (1, 6, 9):
        gz = str(tmp_path / ("many_l%d.fq.gz" % level))
        open(gz, "wb").write(gzip.compress(doc, compresslevel=level))
        # (chunks of a few KB: a chunk's room for symbols is BNS_GZ_RATIO_CAP x the chunk, and this text -- quality lines of four repeating
        # characters -- inflates 8:1 in blocks of 100 KB and more)
        for env in ({}, {"BNS_GZ_CHUNK_KB": 4}, {"BNS_GZ_PIECE_BYTES": 70000, "BNS_GZ_CHUNK_KB": 4}, {"BNS_GZ_CHUNK_KB": 8, "BNS_GZ_TEXT_BYTES": 500000, "BNS_BGZF_HEAD_BYTES": 4096},
                    {"BNS_GZ_PIECE_BYTES": 65536, "BNS_GZ_CHUNK_KB": 16, "BNS_GZ_TEXT_BYTES": 600000}):
            if env:
                env = dict(env, BNS_GZ_RATIO_CAP=400)
            out, err = cli(["-a", files["db"], files["nodes"], gz], **env)
            assert "gzip text on the device" in err and "host parser takes the rest" not in err, (level, env, err)
            assert out == host, (level, env)
            if env.get("BNS_GZ_CHUNK_KB") == 4 and "BNS_GZ_PIECE_BYTES" not in env:
                assert " 1 calls" in err and " 0 calls cut short" in err, err           # (one call; dozens of chunks that chain)
        out, err = cli(["-a", "-g", "0,0", files["db"], files["nodes"], gz], BNS_GZ_CHUNK_KB=4, BNS_GZ_RATIO_CAP=400)      # (more contexts than the stream can use: the first one's)
        assert "gzip text on the device" in err and out == host
    # the host reader's answer for the same file
    out, err = cli(["-a", files["db"], files["nodes"], gz], BNS_GZ_GPU=0)
    assert "gzip text on the device" not in err and out == host
    # several members, one of them empty; a member per record
    third = len(doc) // 3
    cut1, cut2 = doc.index(b"\n@m", third) + 1, doc.index(b"\n@m", 2 * third) + 1
    multi = str(tmp_path / "multi.fq.gz")
    open(multi, "wb").write(gzip.compress(doc[:cut1], 6) + gzip.compress(b"") + gzip.compress(doc[cut1:cut2], 1) + gzip.compress(doc[cut2:], 9))
    out, err = cli(["-a", files["db"], files["nodes"], multi], BNS_GZ_CHUNK_KB=4, BNS_GZ_RATIO_CAP=400)
    assert "gzip text on the device" in err and "3 member(s)" not in err and "4 member(s)" in err and out == host, err
    # many small members (a member per ~40 records): a call per member on the device, independent work for the host reader's threads -- its file
    # from the eighth member on, what the device printed left out
    recs = doc.rstrip(b"\n").split(b"\n@m")
    recs = [recs[0]] + [b"@m" + r for r in recs[1:]]
    many = str(tmp_path / "many_members.fq.gz")
    open(many, "wb").write(b"".join(gzip.compress(b"\n".join(recs[i:i + 40]) + b"\n", 6) for i in range(0, len(recs), 40)))
    out, err = cli(["-a", files["db"], files["nodes"], many])
    assert "gave up: many small members" in err and "host parser takes the rest" in err and out == host, err
    b1, b2 = str(tmp_path / "g1.bin"), str(tmp_path / "g2.bin")
    out, err = cli(["-K", "-b", b1, files["db"], files["nodes"], multi], BNS_GZ_CHUNK_KB=8, BNS_GZ_RATIO_CAP=400)
    cli(["-K", "-b", b2, files["db"], files["nodes"], big_file], BNS_TEXT_GPU=0)
    assert out == b"" and np.array_equal(np.fromfile(b1, dtype=np.uint32), np.fromfile(b2, dtype=np.uint32))
    # wrapped FASTA (a few KB: the member's only block is its last one)
    fa_host, _ = cli(["-a", files["db"], files["nodes"], files["fa"]], BNS_TEXT_GPU=0)
    fgz = str(tmp_path / "one.fa.gz")
    open(fgz, "wb").write(gzip.compress(open(files["fa"], "rb").read()))
    out, err = cli(["-a", files["db"], files["nodes"], fgz])
    assert "gzip text on the device" in err and out == fa_host
    # text the kernels do not take: the host reader reads the file and leaves out what was printed
    reads = files["reads"]
    good = b"".join(b"@g%d\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[:200]))
    stray = b"stray text\n" + b"".join(b"@s%d\n%s\n+\n%s\n" % (i, r.tobytes(), b"I" * r.size) for i, r in enumerate(reads[200:260]))
    for tag, text in (("mid", good + stray + good), ("all", stray)):
        plain = str(tmp_path / (tag + ".fq")); open(plain, "wb").write(text)
        want, _ = cli(["-a", files["db"], files["nodes"], plain], BNS_TEXT_GPU=0)
        g = str(tmp_path / (tag + ".fq.gz")); open(g, "wb").write(gzip.compress(text))
        out, err = cli(["-a", files["db"], files["nodes"], g], BNS_GZ_CHUNK_KB=4, BNS_GZ_TEXT_BYTES=70000, BNS_GZ_RATIO_CAP=400)
        assert "host parser takes the rest" in err and out == want, tag
    # a stream whose blocks inflate beyond a chunk's room (the same record 30 000 times: 300:1): the device gives up, the host reader's output
    rep = str(tmp_path / "rep.fq"); open(rep, "wb").write(good[:good.index(b"@g1\n")] * 30000)
    want, _ = cli(["-K", "-b", b2, files["db"], files["nodes"], rep], BNS_TEXT_GPU=0)
    g = rep + ".gz"; open(g, "wb").write(gzip.compress(open(rep, "rb").read()))
    out, err = cli(["-K", "-b", b1, files["db"], files["nodes"], g], BNS_GZ_CHUNK_KB=4, BNS_GZ_ROOM_RETRY=0)
    assert "gave up" in err and "host parser takes the rest" in err, err
    assert np.array_equal(np.fromfile(b1, dtype=np.uint32), np.fromfile(b2, dtype=np.uint32)) and np.fromfile(b1, dtype=np.uint32).size == 30000
    # ... unless it may ask again with more room (and fewer bytes): eight times, and eight times again
    os.remove(b1)
    out, err = cli(["-K", "-b", b1, files["db"], files["nodes"], g])
    assert "gave up" not in err and "host parser takes the rest" not in err and " 0 asked again" not in err, err
    assert np.array_equal(np.fromfile(b1, dtype=np.uint32), np.fromfile(b2, dtype=np.uint32))
    # a damaged stream fails on either path
    bad = bytearray(open(gz, "rb").read()); bad[len(bad) // 2] ^= 0x55
    badp = str(tmp_path / "bad.fq.gz"); open(badp, "wb").write(bytes(bad))
    for env in ({}, {"BNS_GZ_GPU": "0"}):
        e = dict(os.environ); e.update(env)
        p = subprocess.run([BIN, "classify", "-a", files["db"], files["nodes"], badp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
        assert p.returncode != 0, env


def test_pair_of_plain_files_as_text(files, tmp_path):
    """two plain files, mates by record index, both parsed on the device: output byte for byte that of the host parser -- blocks of
    every size (file 2's blocks scaled to its size), mates of different lengths and forms (FASTQ against wrapped FASTA), a second
    file that is shorter (the pairing ends there, with the reference's warning), text the kernels hand back in either file"""
    reads = files["reads"]
    n = 300
    f1 = str(tmp_path / "p_1.fq"); f2 = str(tmp_path / "p_2.fq"); f2fa = str(tmp_path / "p_2.fa")
    with open(f1, "wb") as a, open(f2, "wb") as b, open(f2fa, "wb") as c:
        for rep in range(6):
            for i in range(n):
                r1, r2 = reads[i], reads[300 + i]
                a.write(b"@m%d_%d/1 c\n%s\n+\n%s\n" % (rep, i, r1.tobytes(), (b"@>+I" * r1.size)[:r1.size]))
                b.write(b"@m%d_%d/2\n%s\n+\n%s\n" % (rep, i, r2.tobytes(), b"I" * r2.size))
                s2 = r2.tobytes()[:max(1, r2.size // 2)]
                c.write(b">m%d_%d/2 x\n" % (rep, i) + b"\n".join(s2[j:j + 40] for j in range(0, len(s2), 40)) + b"\n")
    for second in (f2, f2fa):
        host, _ = cli(["-a", files["db"], files["nodes"], f1, second], BNS_TEXT_GPU=0)
        assert host.count(b"\n") == 6 * n
        for block, devices in ((1 << 22, "0"), (30000, "0"), (4000, "0"), (30000, "0,0"), (4000, "0,0,0"), (2500, "0,0")):
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], f1, second], BNS_TEXT_BLOCK_BYTES=block)
            assert "pair of files, text on the device" in err and "on %d device(s)" % len(devices.split(",")) in err and "host parser takes the rest" not in err, err
            assert out == host, (second, block, devices)
    out, err = cli(["-K", files["db"], files["nodes"], f1, f2])
    _, herr = cli(["-K", files["db"], files["nodes"], f1, f2], BNS_TEXT_GPU=0)
    assert out == b"" and [l for l in err.splitlines() if l.startswith("Classified")] == [l for l in herr.splitlines() if l.startswith("Classified")]
    # the second file is shorter: pairs up to its end
    short = str(tmp_path / "short_2.fq")
    data = open(f2, "rb").read()
    open(short, "wb").write(data[:data.index(b"@m4_17/2")])
    host, herr = cli(["-a", files["db"], files["nodes"], f1, short], BNS_TEXT_GPU=0)
    for block in (1 << 22, 20000):
        out, err = cli(["-a", files["db"], files["nodes"], f1, short], BNS_TEXT_BLOCK_BYTES=block)
        assert out == host and out.count(b"\n") == 4 * n + 17, block
        assert "2nd file has fewer sequences" in err
    # CRLF text in the middle of the second file: read on the device (round 6); stray text there: the device path stops, the host parser
    # reads both files and leaves out what was printed
    lines = data.split(b"\n")
    mid = (len(lines) // 8) * 4
    for tag, text, handed in (("crlf", b"\n".join(lines[:mid]) + b"\n" + b"\r\n".join(lines[mid:mid + 40]) + b"\r\n" + b"\n".join(lines[mid + 40:]), False),
                              ("stray", b"\n".join(lines[:mid]) + b"\nstray text\n" + b"\n".join(lines[mid:]), True)):
        odd = str(tmp_path / (tag + "_2.fq"))
        open(odd, "wb").write(text)
        host, _ = cli(["-a", files["db"], files["nodes"], f1, odd], BNS_TEXT_GPU=0)
        for block, devices in ((1 << 22, "0"), (20000, "0"), (20000, "0,0,0")):
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], f1, odd], BNS_TEXT_BLOCK_BYTES=block)
            assert ("host parser takes the rest" in err) == handed and out == host, (tag, block, devices)


def test_pair_of_bgzf_files_on_the_device(files, tmp_path):
    """two BGZF files, mates by record index, both inflated into device memory and paired there (process_bgzf_gpu_pair): output byte for
    byte that of the plain files through the host parser -- batches of a few members and windows of a few records (each side takes its
    next batch when its window runs low: the two files' batches never end at the same record), members of different sizes in the two
    files, a second file that is shorter (the reference's warning), text the kernels hand back in the second file"""
    reads = files["reads"]
    n = 300
    f1 = str(tmp_path / "q_1.fq"); f2 = str(tmp_path / "q_2.fq")
    with open(f1, "wb") as a, open(f2, "wb") as b:
        for rep in range(8):
            for i in range(n):
                r1, r2 = reads[i], reads[300 + i]
                a.write(b"@m%d_%d/1 comment %d\n%s\n+\n%s\n" % (rep, i, i * rep, r1.tobytes(), (b"@>+I" * r1.size)[:r1.size]))
                b.write(b"@m%d_%d/2\n%s\n+\n%s\n" % (rep, i, r2.tobytes()[:max(1, r2.size - i % 50)], b"I" * max(1, r2.size - i % 50)))
    host, _ = cli(["-a", files["db"], files["nodes"], f1, f2], BNS_TEXT_GPU=0)
    assert host.count(b"\n") == 8 * n
    g1 = str(tmp_path / "q_1.fq.gz"); g2 = str(tmp_path / "q_2.fq.gz")
    synth.write_bgzf(g1, open(f1, "rb").read(), member_sizes=[65280, 30000, 1000, 7])
    synth.write_bgzf(g2, open(f2, "rb").read(), member_sizes=[5000, 65280, 300])
    for members, head in ((16384, None), (3, 20000), (1, 9000), (2, 300000)):
        env = {"BNS_BGZF_BATCH_MEMBERS": members}
        if head:
            env["BNS_BGZF_HEAD_BYTES"] = head
        out, err = cli(["-a", files["db"], files["nodes"], g1, g2], **env)
        assert "pair of BGZF files, text on the device" in err and "host parser takes the rest" not in err, err
        assert out == host, (members, head)
    # round 6: several contexts -- batch b of either file inflated on device b % G, call b = what call b - 1 left of either file + batch b
    # of either, cut in order.  The two files' batches hold different numbers of records (members of other sizes): what one file runs
    # ahead waits in front of its next batch, which is fine while it fits the room there (HEAD: 64 MiB unless set) ...
    for devices, members, env in (("0,0", 3, {}), ("0,0,0", 1, {}), ("0,0", 16384, {}), ("0,0,0", 2, {"BNS_PEER_VIA_HOST": 1})):
        out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], g1, g2], BNS_BGZF_BATCH_MEMBERS=members, **env)
        assert "pair of BGZF files, text on the device" in err and "on %d devices" % len(devices.split(",")) in err and "host parser takes the rest" not in err, err
        assert out == host, (devices, members)
    # ... and handed back to the host parser when it does not
    out, err = cli(["-a", "-g", "0,0", files["db"], files["nodes"], g1, g2], BNS_BGZF_BATCH_MEMBERS=1, BNS_BGZF_HEAD_BYTES=9000)
    assert "host parser takes the rest" in err and out == host
    out, err = cli(["-K", files["db"], files["nodes"], g1, g2], BNS_BGZF_BATCH_MEMBERS=4, BNS_BGZF_HEAD_BYTES=50000)
    _, herr = cli(["-K", files["db"], files["nodes"], f1, f2], BNS_TEXT_GPU=0)
    assert out == b"" and [l for l in err.splitlines() if l.startswith("Classified")] == [l for l in herr.splitlines() if l.startswith("Classified")]
    # the second file is shorter: pairs up to its end
    data = open(f2, "rb").read()
    short = data[:data.index(b"@m5_17/2")]
    ps = str(tmp_path / "qs_2.fq"); open(ps, "wb").write(short)
    gs = str(tmp_path / "qs_2.fq.gz"); synth.write_bgzf(gs, short, member_sizes=[20000, 700])
    host_s, _ = cli(["-a", files["db"], files["nodes"], f1, ps], BNS_TEXT_GPU=0)
    for members, head, devices in ((16384, None, "0"), (2, 30000, "0"), (2, None, "0,0"), (5, None, "0,0,0")):
        env = {"BNS_BGZF_BATCH_MEMBERS": members}
        if head:
            env["BNS_BGZF_HEAD_BYTES"] = head
        out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], g1, gs], **env)
        assert out == host_s and out.count(b"\n") == 5 * n + 17, (members, head, devices)
        assert "2nd file has fewer sequences" in err or "host parser takes the rest" in err, (members, head, devices)
    # CRLF text in the middle of the second file: read on the device (round 6); stray text there: the device path stops, the host parser
    # reads both files and leaves out what was printed
    lines = data.split(b"\n")
    mid = (len(lines) // 8) * 4
    for tag, text, handed in (("crlf", b"\n".join(lines[:mid]) + b"\n" + b"\r\n".join(lines[mid:mid + 40]) + b"\r\n" + b"\n".join(lines[mid + 40:]), False),
                              ("stray", b"\n".join(lines[:mid]) + b"\nstray text\n" + b"\n".join(lines[mid:]), True)):
        pc = str(tmp_path / ("q%s_2.fq" % tag)); open(pc, "wb").write(text)
        gc = str(tmp_path / ("q%s_2.fq.gz" % tag)); synth.write_bgzf(gc, text, member_sizes=[20000, 700])
        host_c, _ = cli(["-a", files["db"], files["nodes"], f1, pc], BNS_TEXT_GPU=0)
        for members, head, devices in ((16384, None, "0"), (2, 30000, "0"), (2, None, "0,0")):
            env = {"BNS_BGZF_BATCH_MEMBERS": members}
            if head:
                env["BNS_BGZF_HEAD_BYTES"] = head
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], g1, gc], **env)
            assert ("host parser takes the rest" in err) == handed and out == host_c, (tag, members, head, devices)


def test_pair_of_gzip_files_on_the_device(files, tmp_path):
    """two plain gzip files (R1.fq.gz + R2.fq.gz), mates by record index: both streams inflated on the device (a GzDeviceSource each, their
    calls side by side) and paired there (process_device_text_pair) -- output byte for byte that of the plain files through the host parser;
    calls of a few dozen KB and windows of a few records, a second file that is shorter, stray text in the second file, a stream the
    device refuses"""
    import gzip
    reads = files["reads"]
    n = 300
    f1 = str(tmp_path / "q_1.fq"); f2 = str(tmp_path / "q_2.fq")
    with open(f1, "wb") as a, open(f2, "wb") as b:
        for rep in range(8):
            for i in range(n):
                r1, r2 = reads[i], reads[300 + i]
                a.write(b"@m%d_%d/1 comment %d\n%s\n+\n%s\n" % (rep, i, i * rep, r1.tobytes(), (b"@>+I" * r1.size)[:r1.size]))
                b.write(b"@m%d_%d/2\n%s\n+\n%s\n" % (rep, i, r2.tobytes()[:max(1, r2.size - i % 50)], b"I" * max(1, r2.size - i % 50)))
    host, _ = cli(["-a", files["db"], files["nodes"], f1, f2], BNS_TEXT_GPU=0)
    assert host.count(b"\n") == 8 * n
    g1 = str(tmp_path / "q_1.fq.gz"); g2 = str(tmp_path / "q_2.fq.gz")
    open(g1, "wb").write(gzip.compress(open(f1, "rb").read(), 6)); open(g2, "wb").write(gzip.compress(open(f2, "rb").read(), 9))
    for env in ({}, {"BNS_GZ_CHUNK_KB": 4, "BNS_GZ_RATIO_CAP": 400}, {"BNS_GZ_CHUNK_KB": 4, "BNS_GZ_RATIO_CAP": 400, "BNS_GZ_TEXT_BYTES": 400000, "BNS_BGZF_HEAD_BYTES": 300000},
                {"BNS_GZ_PIECE_BYTES": 70000, "BNS_GZ_CHUNK_KB": 8, "BNS_GZ_RATIO_CAP": 400, "BNS_GZ_TEXT_BYTES": 600000}):
        out, err = cli(["-a", files["db"], files["nodes"], g1, g2], **env)
        assert "pair of gzip files, text on the device" in err and "host parser takes the rest" not in err, (env, err)
        assert out == host, env
    out, err = cli(["-K", files["db"], files["nodes"], g1, g2], BNS_GZ_CHUNK_KB=4, BNS_GZ_RATIO_CAP=400)
    _, herr = cli(["-K", files["db"], files["nodes"], f1, f2], BNS_TEXT_GPU=0)
    assert out == b"" and [l for l in err.splitlines() if l.startswith("Classified")] == [l for l in herr.splitlines() if l.startswith("Classified")]
    # the host readers' answer for the same files
    out, err = cli(["-a", files["db"], files["nodes"], g1, g2], BNS_GZ_GPU=0)
    assert "text on the device" not in err and out == host
    # the second file is shorter: pairs up to its end (the reference's warning)
    data = open(f2, "rb").read()
    short = data[:data.index(b"@m5_17/2")]
    ps = str(tmp_path / "qs_2.fq"); open(ps, "wb").write(short)
    gs = str(tmp_path / "qs_2.fq.gz"); open(gs, "wb").write(gzip.compress(short))
    host_s, _ = cli(["-a", files["db"], files["nodes"], f1, ps], BNS_TEXT_GPU=0)
    out, err = cli(["-a", files["db"], files["nodes"], g1, gs], BNS_GZ_CHUNK_KB=4, BNS_GZ_RATIO_CAP=400)
    assert out == host_s and out.count(b"\n") == 5 * n + 17 and "2nd file has fewer sequences" in err
    # stray text in the middle of the second file: the device path stops, the host parser reads both files and leaves out what was printed
    lines = data.split(b"\n")
    mid = (len(lines) // 8) * 4
    text = b"\n".join(lines[:mid]) + b"\nstray text\n" + b"\n".join(lines[mid:])
    pc = str(tmp_path / "qstray_2.fq"); open(pc, "wb").write(text)
    gc = str(tmp_path / "qstray_2.fq.gz"); open(gc, "wb").write(gzip.compress(text))
    host_c, _ = cli(["-a", files["db"], files["nodes"], f1, pc], BNS_TEXT_GPU=0)
    for env in ({}, {"BNS_GZ_CHUNK_KB": 4, "BNS_GZ_RATIO_CAP": 400, "BNS_GZ_TEXT_BYTES": 300000}):
        out, err = cli(["-a", files["db"], files["nodes"], g1, gc], **env)
        assert "host parser takes the rest" in err and out == host_c, env
    # a second file whose blocks inflate beyond a chunk's room: the device gives up on it, the host readers take both files
    out, err = cli(["-a", files["db"], files["nodes"], g1, g2], BNS_GZ_CHUNK_KB=4, BNS_GZ_RATIO_CAP=2, BNS_GZ_ROOM_RETRY=0)
    assert "gave up" in err and "host parser takes the rest" in err and out == host
    out, err = cli(["-a", files["db"], files["nodes"], g1, g2], BNS_GZ_CHUNK_KB=4, BNS_GZ_RATIO_CAP=2)      # (asks again with more room: 16, 128 symbols per byte)
    assert "gave up" not in err and "host parser takes the rest" not in err and out == host


def test_fuzzed_bgzf_files_and_pairs(files, tmp_path):
    """random regular and wild text as BGZF -- one file, and two files as mates -- through the device paths (members of random sizes,
    batches of a few members, windows of a few records): stdout byte for byte that of the plain files through the host parser,
    whether the kernels take all of it or hand part of it back"""
    rng = np.random.default_rng(21)
    for it in range(10):
        docs = [ingest_fuzz.make_doc(rng, int(rng.integers(30, 500)), wild=(0.0 if it % 2 else 0.2), final_newline=bool(it % 3)) for _ in range(2)]
        plain, bgz = [], []
        for k, doc in enumerate(docs):
            p = str(tmp_path / ("fb%d_%d.txt" % (it, k))); open(p, "wb").write(doc)
            g = p + ".gz"
            synth.write_bgzf(g, doc, member_sizes=[int(x) for x in rng.integers(1, 5000, 5)])
            plain.append(p); bgz.append(g)
        host1, _ = cli(["-a", files["db"], files["nodes"], plain[0]], BNS_TEXT_GPU=0)
        host2, herr = cli(["-a", files["db"], files["nodes"], plain[0], plain[1]], BNS_TEXT_GPU=0)
        for members, head, devices in ((16384, None, "0"), (2, 6000, "0"), (1, 4096, "0"), (2, 6000, "0,0"), (1, None, "0,0,0")):
            env = {"BNS_BGZF_BATCH_MEMBERS": members}
            if head:
                env["BNS_BGZF_HEAD_BYTES"] = head
            if it % 3 == 0 and devices != "0":
                env["BNS_PEER_VIA_HOST"] = 1
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], bgz[0]], **env)
            assert "BGZF text on the device" in err and out == host1, (it, members, head, devices)
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], bgz[0], bgz[1]], **env)
            assert "pair of BGZF files, text on the device" in err and out == host2, (it, members, head, devices)
        # ... and the plain pair over several contexts
        for block, devices in ((3000, "0,0"), (1500, "0,0,0")):
            out, err = cli(["-a", "-g", devices, files["db"], files["nodes"], plain[0], plain[1]], BNS_TEXT_BLOCK_BYTES=block)
            assert out == host2, (it, block, devices)
