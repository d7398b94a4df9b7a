import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bonsai_amd
from bonsai_amd._lib import GzResult
lib = bonsai_amd.load(); ctx = bonsai_amd.Context(0)
h = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(h)) == 0
gz = open("/tmp/gzbench/r.random.fq.gz", "rb").read()
d_text = ctx.dev_alloc(1 << 30); d_win = ctx.dev_alloc(32768)
B = 419422520
print("bytes at B:", gz[B:B + 16].hex())
for (b0, bit, nb) in ((B, 1, 134225608), (B, 1, 1 << 20), (B - 1000000, 0, 4 << 20), (402653184, 0, 144 << 20)):
    comp = np.frombuffer(gz[b0:b0 + nb], dtype=np.uint8).copy()
    res = GzResult()
    rc = lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, bit, None, d_text, 1 << 30, d_win, C.byref(res))
    print("b0", b0, "bit", bit, "bytes", comp.size, "rc", rc, "status", res.status, "why", res.stop_why, "chunks", res.n_chunks, "chained", res.n_chained, "text", res.text_bytes, "end_bit", res.end_bit, flush=True)
