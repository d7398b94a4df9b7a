import ctypes as C, gzip, sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["BNS_GZ_CHUNK_KB"] = "16"; os.environ["BNS_GZ_RATIO_CAP"] = "400"
import bonsai_amd
from bonsai_amd._lib import GzResult
from test_inflate import gzip_header_end
lib = bonsai_amd.load(); ctx = bonsai_amd.Context(0)
h = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(h)) == 0
rng=np.random.default_rng(0)
reads=[rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8), int(rng.integers(80,151))) for _ in range(300)]
doc=b"".join(b"@m%d_%d/1 c\n%s\n+\n%s\n" % (rep, i, r.tobytes(), (b"@>+I" * r.size)[:r.size]) for rep in range(12) for i,r in enumerate(reads))
gz = gzip.compress(doc, 1)
d_text = ctx.dev_alloc(1 << 22); d_win = ctx.dev_alloc(32768)
for cap in (300000, 1 << 21):
    for nb in (131072, len(gz)):
        comp = np.frombuffer(gz[:nb], dtype=np.uint8).copy()
        res = GzResult()
        rc = lib.bns_inflate_stream_device(h, comp.ctypes.data, comp.size, gzip_header_end(gz) * 8, None, d_text, cap, d_win, C.byref(res))
        print("cap", cap, "bytes", nb, "rc", rc, "status", res.status, "why", res.stop_why, "chunks", res.n_chunks, "chained", res.n_chained, "text", res.text_bytes, "end_bit", res.end_bit, "member_end", res.member_end)
