"""BGZF members inflated on the device (bns_inflate_members, csrc/bns_inflate.hpp): the decoder's logic against zlib on the host
(CPU tier: the same source compiled with g++), the kernel against zlib on the GPU."""
import ctypes as C
import os
import random
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return co.compress(data) + co.flush()


def fastq_text(rng, n):
    recs = []
    for i in range(n):
        s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150))
        q = bytes(rng.integers(35, 75, 150).astype(np.uint8))
        recs.append(b"@r%08d\n" % i + s + b"\n+\n" + q + b"\n")
    return b"".join(recs)


def corpus():
    rng = np.random.default_rng(1)
    return {"empty": b"", "one": b"a", "zeros": bytes(65280), "fastq": fastq_text(rng, 200),
            "rand": bytes(rng.integers(0, 256, 65280).astype(np.uint8)),
            "text": (b"the quick brown fox jumps over the lazy dog " * 1500)[:65280], "ramp": bytes(range(256)) * 200}


def all_streams():
    """(name, text, raw DEFLATE stream): every block type, code shapes from one symbol to 286, matches of every length class"""
    out = []
    for name, d in corpus().items():
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                for ml in (1, 8, 9):
                    out.append(("%s/l%d/s%d/m%d" % (name, level, strat, ml), d, deflate(d, level, strat, ml)))
    return out


@pytest.fixture(scope="module")
def host_decoder(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("inf") / "libinf_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "helpers", "inflate_harness.cpp"),
                    "-o", so], check=True)
    lib = C.CDLL(so)
    lib.inf_host.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]

    def run(comp, n):
        out = C.create_string_buffer(max(n, 1) + 8)
        on, crc = C.c_uint32(), C.c_uint32()
        st = lib.inf_host(comp, len(comp), out, n, C.byref(on), C.byref(crc))
        return st, out.raw[:on.value], crc.value
    return run


def test_host_build_matches_zlib(host_decoder):
    for name, d, c in all_streams():
        st, o, crc = host_decoder(c, len(d))
        assert st == 0 and o == d and crc == (zlib.crc32(d) & 0xFFFFFFFF), name


def test_host_build_reports_damage(host_decoder):
    d = corpus()["fastq"]
    c = deflate(d)
    assert host_decoder(c, len(d) - 5)[0] == 6            # more output than ISIZE says
    assert host_decoder(c, len(d) + 5)[0] == 8            # less
    assert host_decoder(c[:len(c) // 2], len(d))[0] == 7  # the stream runs past its payload
    random.seed(3)
    n_err = 0
    for it in range(600):
        b = bytearray(c)
        for _ in range(random.randint(1, 4)):
            b[random.randrange(len(b))] ^= 1 << random.randrange(8)
        st, o, crc = host_decoder(bytes(b), len(d))
        if st:
            n_err += 1
        elif o != d:
            assert crc != (zlib.crc32(d) & 0xFFFFFFFF)      # what the status does not catch, the checksum does
    assert n_err > 300


def gpu_inflate(lib, h, members, out_lens, pinned=False):
    """members: list of raw DEFLATE payloads; out_lens: expected sizes.  -> (texts, crc, status)"""
    n = len(members)
    in_len = np.array([len(m) for m in members], dtype=np.uint32)
    in_off = np.zeros(n, dtype=np.uint64)
    in_off[1:] = np.cumsum(in_len[:-1].astype(np.uint64) + 3)          # (payloads need no alignment: odd gaps)
    comp = np.zeros(int(in_off[-1]) + int(in_len[-1]) + 1, dtype=np.uint8)
    for m, o in zip(members, in_off):
        comp[int(o):int(o) + len(m)] = np.frombuffer(m, dtype=np.uint8)
    out_len = np.array(out_lens, dtype=np.uint32)
    out_off = np.zeros(n, dtype=np.uint64)
    out_off[1:] = np.cumsum(out_len[:-1].astype(np.uint64))
    text = np.zeros(int(out_off[-1]) + int(out_len[-1]) + 1, dtype=np.uint8)
    crc = np.zeros(n, dtype=np.uint32)
    status = np.full(n, 99, dtype=np.uint32)
    rc = lib.bns_inflate_members(h, comp.ctypes.data, comp.size, in_off.ctypes.data_as(C.POINTER(C.c_uint64)), in_len.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 out_off.ctypes.data_as(C.POINTER(C.c_uint64)), out_len.ctypes.data_as(C.POINTER(C.c_uint32)), n, text.ctypes.data, text.size,
                                 crc.ctypes.data_as(C.POINTER(C.c_uint32)), status.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == 0, lib.bns_inflater_error(h)
    texts = [text[int(o):int(o) + int(l)].tobytes() for o, l in zip(out_off, out_len)]
    return texts, crc, status


@pytest.fixture(scope="module")
def inflater():
    import bonsai_amd
    lib = bonsai_amd.load()
    h = C.c_void_p()
    assert lib.bns_inflater_create(0, C.byref(h)) == 0
    yield lib, h
    lib.bns_inflater_destroy(h)


FORMS = [("lane", "0"), ("lane", "1"), ("wave", "1")]


@pytest.mark.gpu
@pytest.mark.parametrize("form,lut", FORMS)
def test_gpu_matches_zlib(inflater, monkeypatch, form, lut):
    """every form of the kernel: a member per lane with the literal/length code's direct table (small batches) and without (large
    ones), and a member per wavefront (bns_inflate_wave.hpp)"""
    monkeypatch.setenv("BNS_INFLATE_LUT", lut)
    monkeypatch.setenv("BNS_INFLATE_FORM", form)
    lib, h = inflater
    streams = all_streams()
    texts, crc, status = gpu_inflate(lib, h, [c for _, _, c in streams], [len(d) for _, d, _ in streams])
    for (name, d, _), t, cr, st in zip(streams, texts, crc, status):
        assert st == 0 and t == d and int(cr) == (zlib.crc32(d) & 0xFFFFFFFF), name


@pytest.mark.gpu
@pytest.mark.parametrize("form,lut", FORMS)
def test_gpu_many_members_and_damage(inflater, monkeypatch, form, lut):
    monkeypatch.setenv("BNS_INFLATE_LUT", lut)
    monkeypatch.setenv("BNS_INFLATE_FORM", form)
    _many_members_and_damage(inflater)


def _many_members_and_damage(inflater):
    """a BGZF-like batch (5000 members of FASTQ text, the last one short) with damaged members scattered in: the good ones are right,
    the bad ones are flagged (status or checksum) and hurt nobody else"""
    lib, h = inflater
    rng = np.random.default_rng(7)
    base = fastq_text(rng, 4000)
    blocks = [base[i:i + 65280] for i in range(0, len(base), 65280)]
    members, texts_want = [], []
    for i in range(5000):
        b = blocks[i % len(blocks)]
        b = b[(i * 37) % 200:]                               # every member its own text
        members.append(deflate(b, 6 if i % 3 else 1))
        texts_want.append(b)
    random.seed(11)
    damaged = set(random.sample(range(5000), 200))
    sent = []
    for i, m in enumerate(members):
        if i in damaged:
            bb = bytearray(m)
            for _ in range(3):
                bb[random.randrange(len(bb))] ^= 1 << random.randrange(8)
            m = bytes(bb)
        sent.append(m)
    texts, crc, status = gpu_inflate(lib, h, sent, [len(t) for t in texts_want])
    n_flagged = 0
    for i in range(5000):
        want_crc = zlib.crc32(texts_want[i]) & 0xFFFFFFFF
        if i in damaged:
            if status[i] or int(crc[i]) != want_crc:
                n_flagged += 1
            else:
                assert texts[i] == texts_want[i]                # (a flipped bit the stream never used)
        else:
            assert status[i] == 0 and int(crc[i]) == want_crc and texts[i] == texts_want[i], i
    assert n_flagged > 150
    assert lib.bns_inflater_last_kernel_ms(h) > 0


def illumina_like_text(rng, n):
    """FASTQ as sequencers write it: few distinct quality values in long runs (matches of distance 1 and 2, literal codes of 2-3 bits:
    several symbols per look-up), reads that repeat each other (long matches from far back)"""
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 3000)
    recs = []
    for i in range(n):
        st = int(rng.integers(0, genome.size - 150))
        q = np.repeat(rng.choice(np.frombuffer(b"FF:F,#", dtype=np.uint8), 12), rng.integers(1, 40, 12))[:150]
        q = np.concatenate([q, np.full(150 - q.size, ord("F"), dtype=np.uint8)])
        recs.append(b"@M0:%d:1101:%d:%d 1:N:0:1\n" % (i, i * 7 % 30000, i * 13 % 30000) + bytes(genome[st:st + 150]) + b"\n+\n" + bytes(q) + b"\n")
    return b"".join(recs)


@pytest.mark.gpu
@pytest.mark.parametrize("form,lut", FORMS)
def test_gpu_large_members_and_sequencer_like_text(inflater, monkeypatch, form, lut):
    """members beyond BGZF's 64 KiB (the entry point takes any size: hundreds of batches of staged output and many blocks per member
    in the wavefront form, matches from 32 KiB back) and text with the statistics of real FASTQ, at every level"""
    monkeypatch.setenv("BNS_INFLATE_LUT", lut)
    monkeypatch.setenv("BNS_INFLATE_FORM", form)
    lib, h = inflater
    rng = np.random.default_rng(5)
    texts = [illumina_like_text(rng, 3000), fastq_text(rng, 2500), illumina_like_text(rng, 200)[:65536], bytes(rng.integers(0, 4, 300000).astype(np.uint8)),
             (b"0123456789abcdef" * 40000)[:600001], illumina_like_text(rng, 1)[:1], b""]
    streams = [(t, deflate(t, lv, st)) for t in texts for lv, st in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                                                      (6, zlib.Z_FIXED), (6, zlib.Z_RLE))]
    got, crc, status = gpu_inflate(lib, h, [c for _, c in streams], [len(t) for t, _ in streams])
    for i, ((t, _), g, cr, st) in enumerate(zip(streams, got, crc, status)):
        assert st == 0 and g == t and int(cr) == (zlib.crc32(t) & 0xFFFFFFFF), (i, len(t))


@pytest.mark.gpu
def test_gpu_rejects_members_outside_their_buffers(inflater):
    lib, h = inflater
    comp = np.zeros(64, dtype=np.uint8)
    text = np.zeros(64, dtype=np.uint8)
    one = lambda v, t: np.array([v], dtype=t)
    crc, status = one(0, np.uint32), one(0, np.uint32)
    rc = lib.bns_inflate_members(h, comp.ctypes.data, 64, one(60, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64)), one(10, np.uint32).ctypes.data_as(C.POINTER(C.c_uint32)),
                                 one(0, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64)), one(8, np.uint32).ctypes.data_as(C.POINTER(C.c_uint32)), 1, text.ctypes.data, 64,
                                 crc.ctypes.data_as(C.POINTER(C.c_uint32)), status.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc == -1 and b"outside" in lib.bns_inflater_error(h)


def _reader_cases(tmp_path, extra_env):
    """the host reader with a device to inflate on (set_bgzf_device: GPU threads take batches from the back of the task queue, CPU
    inflaters the front -- or, without CPU inflaters, from the front): the records are those of the plain text -- members of every
    size, small batches so that several are in flight and the window ahead of the parser fills, GPU alone and beside CPU threads;
    damage is an error"""
    import hashlib
    import subprocess
    import sys
    import synth
    from bonsai_amd import hostio
    rng = np.random.default_rng(5)
    recs = []
    for i in range(60000):
        L = int(rng.integers(30, 260))
        s_ = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), L, p=[.24, .25, .25, .25, .01]))
        recs.append(b"@q%d c%d\n" % (i, i % 7) + s_ + b"\n+\n" + bytes(rng.integers(33, 74, L).astype(np.uint8)) + b"\n")
    doc = b"".join(recs)
    plain = tmp_path / "d.fq"
    plain.write_bytes(doc)
    want, _ = hostio.read_fastx(str(plain))
    files = {}
    for tag, kw in (("std", {}), ("tiny", {"member_sizes": [1, 7, 300, 65280, 12345]}), ("lvl1", {"level": 1, "block": 40000})):
        files[tag] = str(tmp_path / ("d_%s.fq.gz" % tag))
        synth.write_bgzf(files[tag], doc, **kw)
    code = ("import sys; sys.path.insert(0, %r); from bonsai_amd import hostio; hostio.lib().bnsh_set_bgzf_device(0); "
            "r, _ = hostio.read_fastx(sys.argv[1], chunk_size=int(sys.argv[2]), block_bytes=int(sys.argv[3])); "
            "import hashlib; print(len(r), hashlib.sha256(repr(r).encode()).hexdigest())" % ROOT)
    want_line = [str(len(want)), hashlib.sha256(repr(want).encode()).hexdigest()]
    for tag, path in files.items():
        for env in ({"BNS_BGZF_GPU_BATCH": "4", "BNS_GZ_THREADS": "0"}, {"BNS_BGZF_GPU_BATCH": "2", "BNS_GZ_THREADS": "2", "BNS_BGZF_GPU_THREADS": "3"}, {}):
            for chunk, blk in ((1 << 20, 0), (5000, 70000)):
                out = subprocess.run([sys.executable, "-c", code, path, str(chunk), str(blk)], env=dict(os.environ, BNS_CLI_TIMING="1", **env, **extra_env),
                                     capture_output=True, text=True, timeout=120)
                assert out.returncode == 0, out.stderr[-800:]
                assert out.stdout.split() == want_line, (tag, env, chunk)
                if env.get("BNS_GZ_THREADS") == "0":
                    assert "BGZF on the GPU" in out.stderr and " 0 batches" not in out.stderr      # (the device did the work)
    raw = bytearray(open(files["std"], "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = tmp_path / "bad.fq.gz"
    bad.write_bytes(bytes(raw))
    out = subprocess.run([sys.executable, "-c", code, str(bad), str(1 << 20), "0"], env=dict(os.environ, BNS_GZ_THREADS="0", **extra_env), capture_output=True, text=True,
                         timeout=120)
    assert out.returncode != 0 and "BGZF" in out.stderr


def test_reader_gpu_inflate_threads_against_a_cpu_stand_in(tmp_path):
    """CPU tier: the reader's GPU-inflate threads (queue discipline, windows, staging, error path) against a stand-in for the
    bns_inflater_* entry points that decodes on the host with the product's decoder source and answers as late as the device does"""
    import subprocess
    so = str(tmp_path / "libshim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "helpers", "inflater_shim.cpp"), "-o", so],
                   check=True)
    _reader_cases(tmp_path, {"LD_PRELOAD": so, "BNS_SHIM_LATENCY_MS": "10"})


@pytest.mark.gpu
def test_reader_inflates_bgzf_on_the_device(tmp_path):
    _reader_cases(tmp_path, {})


# ---- one plain gzip stream on the device (bns_inflate_stream_device, csrc/bns_gzstream.hip) ---------------------------------------
def gzip_header_end(b, at=0):
    """offset of the DEFLATE data of the gzip member whose header starts at `at` (RFC 1952)"""
    assert b[at] == 0x1F and b[at + 1] == 0x8B and b[at + 2] == 8
    flg = b[at + 3]
    p = at + 10
    if flg & 4:
        p += 2 + (b[p] | (b[p + 1] << 8))
    if flg & 8:
        p = b.index(b"\0", p) + 1
    if flg & 16:
        p = b.index(b"\0", p) + 1
    if flg & 2:
        p += 2
    return p


class NoRoom(Exception):
    """the stream's first block inflates to more than a chunk's room for symbols (BNS_INF_OUT_OVERFLOW): the host inflater's"""


def gpu_gunzip(lib, h, ctx, gz, piece=None, text_cap=None, stats=None):
    """every member of the gzip file `gz` through bns_inflate_stream_device, `piece` compressed bytes a call -> the text; CRC-32 and
    ISIZE of every member checked against its trailer (bns_crc32_combine over the calls)"""
    from bonsai_amd._lib import GzResult
    piece = piece or len(gz)
    text_cap = text_cap or (64 << 20)
    d_text = ctx.dev_alloc(text_cap + 64)
    d_win = ctx.dev_alloc(32768)
    out = []
    try:
        at = 0
        while at < len(gz):
            data0 = gzip_header_end(gz, at)
            pos_bit = data0 * 8                      # absolute bit position in the file
            fresh = True
            crc, isize = 0, 0
            grow = piece
            while True:
                b0 = pos_bit // 8
                comp = np.frombuffer(gz[b0:b0 + grow], dtype=np.uint8).copy()
                res = GzResult()
                rc = lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, pos_bit - b0 * 8, None if fresh else d_win, d_text, text_cap, d_win, C.byref(res))
                assert rc == 0, lib.bns_inflater_error(h)
                if stats is not None:
                    stats.append((res.n_chunks, res.n_chained, res.stop_why, res.status, res.text_bytes))
                if res.status == 7 and b0 + grow < len(gz):                 # the first block does not end inside the piece: more bytes
                    grow *= 2
                    continue
                if res.status == 6:
                    raise NoRoom()
                assert res.status == 0, (res.status, res.stop_why, at, pos_bit)
                grow = piece
                t = np.zeros(res.text_bytes, dtype=np.uint8)
                if res.text_bytes:
                    ctx.dev_download(d_text, t)
                    assert res.crc32 == (zlib.crc32(t.tobytes()) & 0xFFFFFFFF)
                out.append(t.tobytes())
                crc = lib.bns_crc32_combine(crc, res.crc32, res.text_bytes)
                isize += res.text_bytes
                assert res.end_bit > pos_bit - b0 * 8 or res.member_end
                pos_bit = b0 * 8 + res.end_bit
                fresh = False
                if res.member_end:
                    break
            tr = (pos_bit + 7) // 8
            want_crc = int.from_bytes(gz[tr:tr + 4], "little"); want_isize = int.from_bytes(gz[tr + 4:tr + 8], "little")
            assert crc == want_crc and (isize & 0xFFFFFFFF) == want_isize
            at = tr + 8
    finally:
        ctx.dev_free(d_text); ctx.dev_free(d_win)
    return b"".join(out)


@pytest.fixture(scope="module")
def gz_ctx():
    import bonsai_amd
    c = bonsai_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 6, 9])
def test_gzip_stream_on_the_device(inflater, gz_ctx, level):
    """a FASTQ file as ONE gzip member: found block headers, symbolic decode, chained chunks -- the text is zlib's, whatever the piece"""
    import gzip
    lib, h = inflater
    rng = np.random.default_rng(level)
    text = fastq_text(rng, 30000)                            # ~9.5 MB
    gz = gzip.compress(text, compresslevel=level)
    stats = []
    assert gpu_gunzip(lib, h, gz_ctx, gz, stats=stats) == text
    assert stats[0][0] > 8 and stats[0][1] > 8                # chunks found headers and chained
    # small pieces: a call ends inside a block, the next one goes on from the last block boundary with the window behind it
    assert gpu_gunzip(lib, h, gz_ctx, gz, piece=300000) == text
    # little room for text: calls stop at the room, nothing lost
    assert gpu_gunzip(lib, h, gz_ctx, gz, text_cap=1 << 20) == text


@pytest.mark.gpu
def test_gzip_stream_odd_shapes(inflater, gz_ctx, monkeypatch):
    import gzip
    lib, h = inflater
    rng = np.random.default_rng(5)
    monkeypatch.setenv("BNS_GZ_CHUNK_KB", "16")
    docs = {"tiny": b"@r\nACGT\n+\nIIII\n", "empty": b"", "fixed": fastq_text(rng, 3)[:120],
            "stored": bytes(rng.integers(0, 256, 200000).astype(np.uint8)),
            "text": (b"the quick brown fox jumps over the lazy dog " * 40000),
            "ramp": bytes(range(256)) * 3000, "fastq": fastq_text(rng, 4000)}
    for name, d in docs.items():
        for level in (0, 1, 6, 9):
            gz = gzip.compress(d, compresslevel=level)
            try:
                assert gpu_gunzip(lib, h, gz_ctx, gz) == d, (name, level)
            except NoRoom:
                # (text that compresses 100:1 and more: a block holds more than sixteen times the chunk)
                assert name in ("text", "ramp") and level > 0
    # several members in one file (cat a.gz b.gz), one of them empty
    gz = gzip.compress(docs["fastq"]) + gzip.compress(b"") + gzip.compress(docs["stored"], 1) + gzip.compress(docs["tiny"])
    assert gpu_gunzip(lib, h, gz_ctx, gz) == docs["fastq"] + docs["stored"] + docs["tiny"]
    # a stream that compresses 1000:1 -- more text per chunk than the symbols' room: taken in smaller steps or refused, never wrong
    z = bytes(40 << 20)
    gz = gzip.compress(z, 6)
    from bonsai_amd._lib import GzResult
    comp = np.frombuffer(gz, dtype=np.uint8).copy()
    d_text = gz_ctx.dev_alloc(64 << 20); d_win = gz_ctx.dev_alloc(32768)
    res = GzResult()
    assert lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, gzip_header_end(gz) * 8, None, d_text, 64 << 20, d_win, C.byref(res)) == 0
    if res.status == 0:
        t = np.zeros(res.text_bytes, dtype=np.uint8); gz_ctx.dev_download(d_text, t)
        assert not t.any() and res.text_bytes <= len(z)
    else:
        assert res.status == 6
    gz_ctx.dev_free(d_text); gz_ctx.dev_free(d_win)


@pytest.mark.gpu
def test_gzip_stream_damage_is_reported(inflater, gz_ctx):
    import gzip
    lib, h = inflater
    rng = np.random.default_rng(9)
    text = fastq_text(rng, 8000)
    gz = bytearray(gzip.compress(text, 6))
    random.seed(4)
    caught = 0
    for it in range(12):
        b = bytearray(gz)
        for _ in range(3):
            b[random.randrange(200, len(b) - 8)] ^= 1 << random.randrange(8)
        try:
            got = gpu_gunzip(lib, h, gz_ctx, bytes(b))
        except (AssertionError, ValueError, IndexError):
            caught += 1
            continue
        assert got == text                                    # (passing the trailer's CRC-32 means the text is right)
    assert caught >= 10
    # arguments
    from bonsai_amd._lib import GzResult
    res = GzResult()
    comp = np.frombuffer(bytes(gz), dtype=np.uint8).copy()
    assert lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, comp.size * 8, None, 1, 1 << 20, 1, C.byref(res)) == -1
    assert lib.bns_inflate_stream_device(h, None, comp.size, 80, None, 1, 1 << 20, 1, C.byref(res)) == -1


def test_crc32_combine_matches_zlib():
    import bonsai_amd
    lib = bonsai_amd.load()
    rng = np.random.default_rng(2)
    for n1, n2 in ((0, 0), (1, 0), (0, 5), (10, 1), (1000, 77777), (123456, 1 << 20)):
        a = bytes(rng.integers(0, 256, n1).astype(np.uint8)); b = bytes(rng.integers(0, 256, n2).astype(np.uint8))
        assert lib.bns_crc32_combine(zlib.crc32(a), zlib.crc32(b), n2) == zlib.crc32(a + b)


@pytest.mark.gpu
def test_gzip_stream_many_chunks_and_a_cut_last_chunk(inflater, gz_ctx, monkeypatch):
    """more than 1024 chunks (the validity scan works in tiles of 1024), the last of them cut by the end of the bytes passed: the call
    ends at the chunk in front and says so -- the result words are written by the last chunk that counts, whichever tile it is in"""
    import gzip
    lib, h = inflater
    monkeypatch.setenv("BNS_GZ_CHUNK_KB", "4")
    rng = np.random.default_rng(11)
    m = 130000                                               # records of 314 bytes: ~41 MB of text, ~20 MB of DEFLATE, a block header every ~16 KB
    rec = np.empty((m, 314), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1:9] = np.frombuffer(b"%08d" % 0, dtype=np.uint8)
    for j in range(8):
        rec[:, 8 - j] = ord("0") + (np.arange(m) // 10 ** j) % 10
    rec[:, 9] = 10
    rec[:, 10:160] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(m, 150))
    rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
    rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
    rec[:, 313] = 10
    text = rec.tobytes()
    gz = gzip.compress(text, compresslevel=6)
    from bonsai_amd._lib import GzResult
    d_text = gz_ctx.dev_alloc(64 << 20); d_win = gz_ctx.dev_alloc(32768)
    he = gzip_header_end(gz)
    seen = set()
    for cut in (0, 3000, 9000, 20000, 33000):
        comp = np.frombuffer(gz[he:len(gz) - 8 - cut], dtype=np.uint8).copy()
        res = GzResult()
        assert lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, 0, None, d_text, 64 << 20, d_win, C.byref(res)) == 0
        assert res.status == 0 and res.n_chained > 1030 and res.n_chunks - res.n_chained <= 1, (cut, res.n_chunks, res.n_chained)
        seen.add(res.n_chunks - res.n_chained)
        t = np.zeros(res.text_bytes, dtype=np.uint8); gz_ctx.dev_download(d_text, t)
        assert res.text_bytes > 30 << 20 and t.tobytes() == text[:res.text_bytes], cut
        assert res.crc32 == (zlib.crc32(t.tobytes()) & 0xFFFFFFFF)
        assert (res.member_end == 1) == (cut == 0) and (cut or res.text_bytes == len(text))
        assert 0 < res.end_bit <= comp.size * 8
    assert seen == {0, 1}                                     # (both: the last entry whole, and cut in its first block)
    gz_ctx.dev_free(d_text); gz_ctx.dev_free(d_win)


@pytest.mark.gpu
def test_gzip_stream_from_prefetched_bytes(inflater, gz_ctx, monkeypatch):
    """bns_inflate_stream_prefetch: the calls read their bytes inside a range brought up ahead -- at any byte offset of it (the kernels read
    from the 4-byte boundary in front, every position shifted) -- and say where they stopped in the CALL's coordinates; two ranges are kept"""
    import gzip
    from bonsai_amd._lib import GzResult
    lib, h = inflater
    monkeypatch.setenv("BNS_GZ_CHUNK_KB", "8")
    rng = np.random.default_rng(21)
    texts = [fastq_text(rng, 9000), fastq_text(rng, 7000)]
    gzs = [gzip.compress(t, 6) for t in texts]
    bufs = []
    for gz in gzs:
        pc = C.c_void_p()
        assert lib.bns_inflater_host_alloc(h, len(gz) + 64, C.byref(pc)) == 0
        C.memmove(pc, gz, len(gz))
        assert lib.bns_inflate_stream_prefetch(h, pc, len(gz)) == 0
        bufs.append(pc)
    d_text = gz_ctx.dev_alloc(8 << 20); d_win = gz_ctx.dev_alloc(32768)
    for gz, text, pc in zip(gzs, texts, bufs):
        for piece in (len(gz), 200001, 77777):
            pos = gzip_header_end(gz) * 8
            fresh, got = True, []
            while True:
                b0 = pos // 8                                     # (any alignment: 10, then wherever a call stopped)
                res = GzResult()
                nb = min(piece, len(gz) - b0)
                assert lib.bns_inflate_stream_device(h, C.c_void_p(pc.value + b0), nb, pos - 8 * b0, None if fresh else d_win, d_text, 8 << 20, d_win, C.byref(res)) == 0
                assert res.status == 0, (res.status, piece, b0)
                t = np.zeros(res.text_bytes, dtype=np.uint8)
                if res.text_bytes:
                    gz_ctx.dev_download(d_text, t)
                got.append(t.tobytes())
                assert res.crc32 == (zlib.crc32(t.tobytes()) & 0xFFFFFFFF)
                pos = 8 * b0 + res.end_bit; fresh = False
                if res.member_end:
                    break
            assert b"".join(got) == text, piece
    assert lib.bns_inflate_stream_prefetch(h, None, 10) == -1 and lib.bns_inflate_stream_prefetch(h, bufs[0], 0) == -1
    for pc in bufs:
        lib.bns_inflater_host_free(h, pc)
    gz_ctx.dev_free(d_text); gz_ctx.dev_free(d_win)
