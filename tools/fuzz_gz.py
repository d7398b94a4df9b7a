#!/usr/bin/env python
"""round 6: bns_inflate_stream_device against zlib on random gzip files -- texts of every kind (FASTQ-like, random bytes, long runs, tiny), every
level / strategy / memLevel zlib has (stored blocks, fixed codes, Huffman only, RLE: streams with few or no dynamic headers to find), one member
or several (empty ones among them), pigz-style sync flushes, chunks of 4-64 KiB, calls of 40 KB to the whole file, little room for text or symbols.
What the device takes must be zlib's text byte for byte (CRC-32 and ISIZE against the trailers); what it refuses it must refuse with
BNS_INF_OUT_OVERFLOW (room) -- never a wrong byte.   usage (GPU box): python tools/fuzz_gz.py [seconds=240] [seed=1]"""
import ctypes as C, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bonsai_amd
from bonsai_amd._lib import GzResult
from test_inflate import fastq_text, gzip_header_end

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = bonsai_amd.load(); ctx = bonsai_amd.Context(0)
h = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(h)) == 0
CAP = 96 << 20
d_text = ctx.dev_alloc(CAP + 64); d_win = ctx.dev_alloc(32768)


def make_text(rng):
    kind = rng.integers(0, 6)
    n = int(rng.choice([0, 1, 50, 5000, 200000, 3000000]))
    if kind == 0: return fastq_text(rng, max(1, n // 320))
    if kind == 1: return bytes(rng.integers(0, 256, n).astype(np.uint8))
    if kind == 2: return bytes(rng.choice(np.frombuffer(b"ACGT\n", dtype=np.uint8), n))
    if kind == 3: return (b"@read\nACGTACGTAC\n+\nIIIIIIIIII\n" * (n // 28 + 1))[:n]
    if kind == 4: return bytes(n)
    return fastq_text(rng, max(1, n // 900))[:n] + bytes(rng.integers(0, 256, n // 3).astype(np.uint8))


def member(rng, data):
    level = int(rng.choice([0, 1, 1, 6, 6, 9])); strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
    ml = int(rng.choice([1, 4, 8, 8, 9]))
    if rng.random() < 0.3 and len(data) > 100000:            # pigz-style: pieces ended by a sync flush, primed with the text in front
        out, prev = [], b""
        P = int(rng.choice([30000, 250000, 1 << 20]))
        for i in range(0, len(data), P):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, ml, strat, prev[-32768:]) if prev else zlib.compressobj(level, zlib.DEFLATED, -15, ml, strat)
            piece = data[i:i + P]
            out.append(co.compress(piece) + co.flush(zlib.Z_FINISH if i + P >= len(data) else zlib.Z_SYNC_FLUSH))
            prev = piece
        body = b"".join(out)
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, ml, strat)
        body = co.compress(data) + co.flush()
    return b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


t0 = time.time(); it = n_whole = n_refused = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 7919 + it)
    os.environ["BNS_GZ_CHUNK_KB"] = str(int(rng.choice([4, 8, 16, 64])))
    os.environ["BNS_GZ_RATIO_CAP"] = str(int(rng.choice([2, 16, 16, 400])))
    texts = [make_text(rng) for _ in range(int(rng.choice([1, 1, 1, 2, 4])))]
    gz = b"".join(member(rng, t) for t in texts)
    if rng.random() < 0.2: gz += bytes(int(rng.integers(1, 50)))          # zero padding behind the last member
    piece = int(rng.choice([40000, 300000, 1 << 30]))
    cap = int(rng.choice([70000, 1 << 20, CAP]))
    at, out, refused = 0, [], False
    try:
        for text in texts:
            pos = gzip_header_end(gz, at) * 8
            fresh, crc, isize, grow = True, 0, 0, piece
            while True:
                b0 = pos // 8
                comp = np.frombuffer(gz[b0:b0 + grow], dtype=np.uint8).copy()
                res = GzResult()
                rc = lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, pos - 8 * b0, None if fresh else d_win, d_text, cap, d_win, C.byref(res))
                assert rc == 0, rc
                if res.status == 7 and b0 + grow < len(gz):
                    grow *= 2; continue
                if res.status == 6:
                    refused = True; break
                assert res.status == 0, ("status", res.status, res.stop_why)
                grow = piece
                t = np.zeros(res.text_bytes, dtype=np.uint8)
                if res.text_bytes: ctx.dev_download(d_text, t)
                out.append(t.tobytes())
                crc = lib.bns_crc32_combine(crc, res.crc32, res.text_bytes); isize += res.text_bytes
                assert res.end_bit > pos - 8 * b0 or res.member_end, "no progress"
                pos = 8 * b0 + res.end_bit; fresh = False
                if res.member_end: break
            if refused: break
            tr = (pos + 7) // 8
            assert crc == int.from_bytes(gz[tr:tr + 4], "little") and (isize & 0xFFFFFFFF) == int.from_bytes(gz[tr + 4:tr + 8], "little"), "trailer"
            at = tr + 8
        got = b"".join(out); want = b"".join(texts)
        if refused:
            n_refused += 1
            assert want.startswith(got), "refused, but what it delivered is wrong"
        else:
            n_whole += 1
            assert got == want, "text differs"
    except AssertionError as e:
        print("GZ FUZZ MISMATCH seed", seed0 * 7919 + it, e, {k: os.environ[k] for k in ("BNS_GZ_CHUNK_KB", "BNS_GZ_RATIO_CAP")}, "piece", piece, "cap", cap, "texts", [len(t) for t in texts])
        sys.exit(1)
    it += 1
print("gz fuzz ok: %d files in %.0f s (%d inflated whole, %d refused for room)" % (it, time.time() - t0, n_whole, n_refused))
