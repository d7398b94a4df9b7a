import ctypes as C, gzip, sys, os, glob
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["BNS_GZ_CHUNK_KB"] = "16"; os.environ["BNS_GZ_RATIO_CAP"] = "400"
import bonsai_amd
from bonsai_amd._lib import GzResult
from test_inflate import gzip_header_end
lib = bonsai_amd.load(); ctx = bonsai_amd.Context(0)
h = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(h)) == 0
path = sorted(glob.glob("/tmp/pytest-of-root/pytest-*/test_gzip_text_stays_on_the_de*/many_l1.fq.gz"))[-1]
gz = open(path, "rb").read()
print(path, len(gz))
d_text = ctx.dev_alloc((64 << 20) + (1 << 22)); d_win = ctx.dev_alloc(32768)
he = gzip_header_end(gz)
for cap in (300000, 1 << 21):
    for nb in (131072, len(gz)):
        for base in (0, he):
            comp = np.frombuffer(gz[base:nb], dtype=np.uint8).copy()
            res = GzResult()
            rc = lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, (he - base) * 8, None, d_text + (64 << 20), cap, d_win, C.byref(res))
            print("cap", cap, "bytes", nb, "base", base, "rc", rc, "status", res.status, "why", res.stop_why, "chunks", res.n_chunks, "chained", res.n_chained, "text", res.text_bytes, "end_bit", res.end_bit, "member_end", res.member_end)
