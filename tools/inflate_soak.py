#!/usr/bin/env python3
"""randomised soak of the device inflate against zlib: streams of random kinds of text (alphabets of 1..256 symbols, repeats at random
distances, FASTQ-like records, runs), random sizes (0 .. 200 KB), every level / strategy / memLevel / window size, in batches; then
the same streams with bits flipped, truncated or with a byte cut out (must be flagged by status or CRC, or inflate to the right text,
and must not disturb their neighbours).  usage (GPU box): BNS_INFLATE_FORM=wave|lane python tools/inflate_soak.py [seconds=60] [seed=1]"""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_inflate import fastq_text, gpu_inflate, illumina_like_text  # noqa: E402


def random_text(rng):
    kind = int(rng.integers(0, 7))
    n = int(rng.choice([0, 1, 2, 3, 17, 255, 256, 257, 1000, 4096, 32768, 65280, 65536, int(rng.integers(0, 200000))]))
    if kind == 0:
        return bytes(rng.integers(0, int(rng.integers(1, 257)), n).astype(np.uint8))
    if kind == 1:                                        # a short pattern repeated: matches of distance 1..300, every length
        p = bytes(rng.integers(0, 256, int(rng.integers(1, 300))).astype(np.uint8))
        return (p * (n // len(p) + 1))[:n]
    if kind == 2:
        return fastq_text(rng, n // 314 + 1)[:n]
    if kind == 3:
        return illumina_like_text(rng, n // 330 + 1)[:n]
    if kind == 4:                                        # runs of random lengths
        v = rng.integers(0, 256, n // 4 + 1).astype(np.uint8)
        return bytes(np.repeat(v, rng.integers(1, 40, v.size)))[:n]
    if kind == 5:                                        # text that repeats itself from far back (32 KiB window edge)
        a = bytes(rng.integers(0, 64, 33000).astype(np.uint8))
        return (a + a[:int(rng.integers(1, 33000))] + a)[:max(n, 1)]
    return bytes(rng.integers(0, 4, n).astype(np.uint8) + 65)


def deflate_any(rng, data):
    level = int(rng.integers(0, 10))
    strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
    co = zlib.compressobj(level, zlib.DEFLATED, -int(rng.integers(9, 16)), int(rng.integers(1, 10)), strat)
    out = b""
    # (flush points in the middle make stored / empty blocks inside a stream)
    cuts = sorted(int(c) for c in rng.integers(0, len(data) + 1, int(rng.integers(0, 3))))
    prev = 0
    for c in cuts:
        out += co.compress(data[prev:c]) + co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH])))
        prev = c
    return out + co.compress(data[prev:]) + co.flush()


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import bonsai_amd
    lib = bonsai_amd.load()
    h = C.c_void_p()
    assert lib.bns_inflater_create(0, C.byref(h)) == 0
    rng = np.random.default_rng(seed)
    t_end = time.time() + secs
    n_ok = n_dmg = n_flag = n_bytes = n_batches = 0
    while time.time() < t_end:
        texts = [random_text(rng) for _ in range(int(rng.integers(1, 600)))]
        comp = [deflate_any(rng, t) for t in texts]
        got, crc, status = gpu_inflate(lib, h, comp, [len(t) for t in texts])
        for i, t in enumerate(texts):
            assert status[i] == 0 and got[i] == t and int(crc[i]) == (zlib.crc32(t) & 0xFFFFFFFF), ("clean stream", seed, n_batches, i, len(t), int(status[i]))
        n_ok += len(texts); n_bytes += sum(len(t) for t in texts)
        # the same batch, every third stream damaged
        sent = list(comp)
        hurt = set()
        for i in range(0, len(comp), 3):
            b = bytearray(comp[i])
            if not b:
                continue
            how = int(rng.integers(0, 3))
            if how == 0:
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            elif how == 1:
                b = b[:int(rng.integers(0, len(b)))]
            else:
                k = int(rng.integers(0, len(b))); del b[k]
            sent[i] = bytes(b); hurt.add(i)
        got, crc, status = gpu_inflate(lib, h, sent, [len(t) for t in texts])
        for i, t in enumerate(texts):
            want = zlib.crc32(t) & 0xFFFFFFFF
            if i in hurt:
                n_dmg += 1
                if status[i] or int(crc[i]) != want:
                    n_flag += 1
                else:
                    assert got[i] == t, ("damaged stream passed as clean with wrong text", seed, n_batches, i)
            else:
                assert status[i] == 0 and got[i] == t and int(crc[i]) == want, ("neighbour of a damaged stream", seed, n_batches, i)
        n_batches += 1
    print("form %s, seed %d: %d batches, %d clean streams (%.1f MB of text) all equal to zlib's; %d damaged streams: %d flagged, %d inflated to the right text anyway"
          % (os.environ.get("BNS_INFLATE_FORM", "wave (default)"), seed, n_batches, n_ok, n_bytes / 1e6, n_dmg, n_flag, n_dmg - n_flag), flush=True)
    lib.bns_inflater_destroy(h)


if __name__ == "__main__":
    main()
