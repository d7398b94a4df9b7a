#!/usr/bin/env python
"""bns_inflate_members on batches of BGZF-sized members (65 280 bytes of FASTQ text each, zlib level 6 like bgzip): kernel time from
HIP events, call time from page-locked buffers (upload + kernel + text back), GB/s of TEXT.
usage (GPU box): python tools/inflate_bench.py [distinct=512] [sizes=1024,4096,16384,65536]"""
import ctypes as C, os, sys, time, zlib
from multiprocessing import Pool
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(seed):
    rng = np.random.default_rng(seed)
    m = 208                                                  # 208 records x 314 bytes = 65 312 -> cut to 65 280
    rec = np.empty((m, 314), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    idx = np.arange(seed * m, seed * m + m)
    for j in range(8):
        rec[:, 9 - j] = ord("0") + (idx // 10 ** j) % 10
    rec[:, 9] = 10
    rec[:, 10:160] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, 150))]
    rec[:, 160] = 10; rec[:, 161] = ord("+"); rec[:, 162] = 10
    rec[:, 163:313] = rng.integers(35, 75, size=(m, 150)).astype(np.uint8)
    rec[:, 313] = 10
    text = rec.tobytes()[:65280]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    return text, co.compress(text) + co.flush()


def main():
    import bonsai_amd
    distinct = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,4096,16384,65536").split(",")]
    with Pool(min(16, os.cpu_count() or 1)) as p:
        base = p.map(make, range(distinct))
    t0 = time.time()
    for t, c in base[:64]:
        assert zlib.decompress(c, -15) == t
    print("zlib inflate, one thread: %.2f GB/s of text; compressed / text = %.2f" % (64 * 65280 / (time.time() - t0) / 1e9, len(base[0][1]) / 65280), flush=True)
    lib = bonsai_amd.load()
    h = C.c_void_p()
    assert lib.bns_inflater_create(0, C.byref(h)) == 0
    nmax = max(sizes)
    in_len_all = np.array([len(base[i % distinct][1]) for i in range(nmax)], dtype=np.uint32)
    comp_cap = int(in_len_all.astype(np.uint64).sum()) + 64
    text_cap = nmax * 65280 + 64
    pc, pt = C.c_void_p(), C.c_void_p()
    assert lib.bns_inflater_host_alloc(h, comp_cap, C.byref(pc)) == 0 and lib.bns_inflater_host_alloc(h, text_cap, C.byref(pt)) == 0
    comp = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint8)), shape=(comp_cap,))
    text = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_uint8)), shape=(text_cap,))
    in_off_all = np.zeros(nmax, dtype=np.uint64)
    in_off_all[1:] = np.cumsum(in_len_all[:-1].astype(np.uint64))
    for i in range(nmax):
        c = base[i % distinct][1]
        comp[int(in_off_all[i]):int(in_off_all[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    want_crc = np.array([zlib.crc32(base[i % distinct][0]) & 0xFFFFFFFF for i in range(distinct)], dtype=np.uint32)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    for n in sizes:
        in_off, in_len = in_off_all[:n].copy(), in_len_all[:n].copy()
        out_len = np.full(n, 65280, dtype=np.uint32)
        out_off = np.arange(n, dtype=np.uint64) * 65280
        crc = np.zeros(n, dtype=np.uint32); status = np.zeros(n, dtype=np.uint32)
        cb = int(in_off[-1]) + int(in_len[-1])
        best, kms = 1e9, 1e9
        for rep in range(4):
            t0 = time.time()
            rc = lib.bns_inflate_members(h, pc, cb, in_off.ctypes.data_as(u64p), in_len.ctypes.data_as(u32p), out_off.ctypes.data_as(u64p), out_len.ctypes.data_as(u32p),
                                         n, pt, n * 65280, crc.ctypes.data_as(u32p), status.ctypes.data_as(u32p))
            dt = time.time() - t0
            assert rc == 0, lib.bns_inflater_error(h)
            if rep:
                best = min(best, dt); kms = min(kms, lib.bns_inflater_last_kernel_ms(h))
        ok = bool((status == 0).all() and (crc == want_crc[np.arange(n) % distinct]).all())
        ok = ok and text[:65280].tobytes() == base[0][0] and text[(n - 1) * 65280:n * 65280].tobytes() == base[(n - 1) % distinct][0]
        tb = n * 65280
        print("%6d members (%6.1f MB of text): kernel %7.2f ms = %6.2f GB/s; call %7.2f ms = %6.2f GB/s (%.2f M reads/s of 314-byte records); all correct: %s"
              % (n, tb / 1e6, kms, tb / kms / 1e6, best * 1e3, tb / best / 1e9, tb / 314 / best / 1e6, ok), flush=True)
    # several handles at once (what the reader's threads do): K threads, each its own handle and buffers, 4096-member batches
    import threading
    for K in (2, 3, 4):
        n = 4096
        hs, bufs = [], []
        for k in range(K):
            hk = C.c_void_p(); assert lib.bns_inflater_create(0, C.byref(hk)) == 0
            tk = C.c_void_p(); assert lib.bns_inflater_host_alloc(hk, n * 65280 + 64, C.byref(tk)) == 0
            hs.append(hk); bufs.append(tk)
        in_off, in_len = in_off_all[:n].copy(), in_len_all[:n].copy()
        out_len = np.full(n, 65280, dtype=np.uint32); out_off = np.arange(n, dtype=np.uint64) * 65280
        cb = int(in_off[-1]) + int(in_len[-1])
        res = [(np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)) for _ in range(K)]
        rounds = 6
        def work(k):
            for _ in range(rounds):
                rc = lib.bns_inflate_members(hs[k], pc, cb, in_off.ctypes.data_as(u64p), in_len.ctypes.data_as(u32p), out_off.ctypes.data_as(u64p), out_len.ctypes.data_as(u32p),
                                             n, bufs[k], n * 65280, res[k][0].ctypes.data_as(u32p), res[k][1].ctypes.data_as(u32p))
                assert rc == 0
        work(0)
        t0 = time.time()
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.time() - t0
        ok = all(bool((r[1] == 0).all() and (r[0] == want_crc[np.arange(n) % distinct]).all()) for r in res)
        tb = K * rounds * n * 65280
        print("%d handles x %d-member batches at once: %.2f GB/s of text (%.1f M reads/s), kernel of the last batch %.1f ms; all correct: %s"
              % (K, n, tb / dt / 1e9, tb / 314 / dt / 1e6, lib.bns_inflater_last_kernel_ms(hs[0]), ok), flush=True)
        for k in range(K):
            lib.bns_inflater_host_free(hs[k], bufs[k]); lib.bns_inflater_destroy(hs[k])
    lib.bns_inflater_destroy(h)


if __name__ == "__main__":
    main()
