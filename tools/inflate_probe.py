#!/usr/bin/env python3
"""what one member costs the inflate kernels by kind of stream: N copies of one 65 280-byte member of FASTQ text deflated several ways
(level 6 as bgzip does; literals only; fixed codes; level 1), kernel time of the form named by BNS_INFLATE_FORM.
usage (GPU box): BNS_INFLATE_FORM=wave python tools/inflate_probe.py [N=1024]"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_inflate import deflate, fastq_text, gpu_inflate  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    import bonsai_amd
    lib = bonsai_amd.load()
    lib.bns_inflater_last_kernel_ms.restype = C.c_float
    h = C.c_void_p()
    assert lib.bns_inflater_create(0, C.byref(h)) == 0
    rng = np.random.default_rng(3)
    text = fastq_text(rng, 220)[:65280]
    quals = bytes(rng.integers(35, 75, 65280).astype(np.uint8))
    kinds = [("fastq level 6", text, deflate(text, 6)), ("fastq level 1", text, deflate(text, 1)),
             ("fastq literals only", text, deflate(text, 6, zlib.Z_HUFFMAN_ONLY)), ("fastq fixed codes", text, deflate(text, 6, zlib.Z_FIXED)),
             ("qualities only (literals)", quals, deflate(quals, 6)), ("zeros", bytes(65280), deflate(bytes(65280), 6))]
    for name, d, c in kinds:
        best = 1e9
        for _ in range(3):
            texts, crc, status = gpu_inflate(lib, h, [c] * n, [len(d)] * n)
            best = min(best, lib.bns_inflater_last_kernel_ms(h))
        ok = all(s == 0 for s in status) and texts[0] == d and texts[-1] == d and int(crc[0]) == (zlib.crc32(d) & 0xFFFFFFFF)
        print("%-28s compressed %6d bytes: kernel %7.2f ms for %d members (%s)" % (name, len(c), best, n, "correct" if ok else "WRONG"), flush=True)
    lib.bns_inflater_destroy(h)


if __name__ == "__main__":
    main()
