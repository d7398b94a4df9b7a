"""bns_inflate_stream_device on a big FASTQ gzip member: kernel time per call, text rate"""
import ctypes as C, gzip, sys, time, zlib, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bonsai_amd
from bonsai_amd._lib import GzResult
from test_inflate import fastq_text, gzip_header_end
lib = bonsai_amd.load()
ctx = bonsai_amd.Context(0)
h = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(h)) == 0
rng = np.random.default_rng(1)
t0 = time.time()
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    gz = open(sys.argv[1], "rb").read()
    text = None
    print("file %s: %.1f MB" % (sys.argv[1], len(gz) / 1e6), flush=True)
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    base = fastq_text(rng, 20000)
    # varied text: shuffle records so that matches do not span copies trivially
    recs = [base[i:i + 322] for i in range(0, len(base), 322)]
    text = b"".join(recs[int(j)] for j in rng.integers(0, len(recs), n))
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(text) + co.flush()
    print("text %.1f MB gz %.1f MB (%.1f s to make)" % (len(text) / 1e6, len(gz) / 1e6, time.time() - t0), flush=True)
piece = int(float(os.environ.get("PIECE_MB", "256")) * (1 << 20))
cap = 1900 << 20
d_text = ctx.dev_alloc(cap + 64); d_win = ctx.dev_alloc(32768)
pc = C.c_void_p(); assert lib.bns_inflater_host_alloc(h, piece + 64, C.byref(pc)) == 0
for rep in range(3):
    pos = gzip_header_end(gz) * 8
    fresh = True; total = 0; calls = 0; kms = 0.0
    t0 = time.time()
    while True:
        b0 = pos // 8
        nb = min(piece, len(gz) - b0)
        C.memmove(pc, gz[b0:b0 + nb], nb)
        res = GzResult()
        t1 = time.time()
        rc = lib.bns_inflate_stream_device(h, pc, nb, pos - b0 * 8, None if fresh else d_win, d_text, cap, d_win, C.byref(res))
        dt = time.time() - t1
        if rc != 0 or res.status != 0:
            print("  FAILED: rc %d status %d why %d chunks %d chained %d at file bit %d (byte %d), piece %d bytes" % (rc, res.status, res.stop_why, res.n_chunks, res.n_chained, pos, b0, nb), flush=True)
            sys.exit(1)
        k = lib.bns_inflater_last_kernel_ms(h)
        print("  call %d: %d bytes in, chunks %d chained %d why %d, text %.1f MB, kernels %.2f ms, call %.2f ms" % (calls, nb, res.n_chunks, res.n_chained, res.stop_why, res.text_bytes / 1e6, k, dt * 1e3), flush=True)
        kms += k; total += res.text_bytes; calls += 1
        pos = b0 * 8 + res.end_bit; fresh = False
        if res.member_end: break
    assert text is None or total == len(text)
    print("rep %d: %d calls, kernels %.1f ms = %.1f GB/s of text, wall %.1f ms" % (rep, calls, kms, total / kms / 1e6, (time.time() - t0) * 1e3), flush=True)
