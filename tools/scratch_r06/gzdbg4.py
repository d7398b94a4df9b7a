import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["BNS_GZ_DEBUG"] = "1"
import bonsai_amd
from bonsai_amd._lib import GzResult
lib = bonsai_amd.load(); ctx = bonsai_amd.Context(0)
h = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(h)) == 0
gz = open("/tmp/gzbench/r.random.fq.gz", "rb").read()
d_text = ctx.dev_alloc(1 << 30); d_win = ctx.dev_alloc(32768)
S, END = 285176914, 400 << 20
for (b0, bit, e) in ((285208530, 1, END), (285192624, 4, END)):
    comp = np.frombuffer(gz[b0:e], dtype=np.uint8).copy()
    res = GzResult()
    rc = lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, bit, None, d_text, 1 << 30, d_win, C.byref(res))
    eb = b0 * 8 + res.end_bit
    print("b0", b0, "bit", bit, "bytes", comp.size, "rc", rc, "status", res.status, "why", res.stop_why, "chunks", res.n_chunks, "chained", res.n_chained, "text", res.text_bytes, "ends at byte", eb >> 3, "bit", eb & 7, "rel END", (eb >> 3) - END, flush=True)
