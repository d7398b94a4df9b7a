"""A db too big to sit in HBM next to its clustered table: build 8e9 keys on the device (2^34 khash buckets, 210 GB), take the
arrays to the host, free the device, and load them back STREAMED (bns_load_table) into the clustered layout; classify 10 M reads,
compare a sample with the oracle.   usage (GPU box): python tools/big_stream.py [genomes=36000] [log2_buckets=34] [identity=0] [buckets=0] [dbg,dbg,...]
(the last argument: bns_debug_set values, hex, one load + classify + parity sample per value on the same arrays and reads -- 0x10 = keys in
arrival order, 0 = what the loader chooses, i.e. the group-aware fill for a table this crowded)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, bonsai_amd

NG = int(sys.argv[1]) if len(sys.argv) > 1 else 36000
LG = int(sys.argv[2]) if len(sys.argv) > 2 else 34
IDENT = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # minimizer identity bits (0 = automatic)
NBUCKETS = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # home buckets (0 = automatic)
DBGS = [int(x, 16) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
G, K, L, N = 1 << 18, 31, 150, 10_000_000
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = bonsai_amd.Context(0)
ctx.set_encoder(K, None, canonicalize=True)
parent, leaves = bench.make_taxonomy(NG)
ctx.load_taxonomy(parent)
nb = 1 << LG
t0 = time.time()
flags = torch.empty(nb >> 4, dtype=torch.int32, device=dev); keys = torch.empty(nb, dtype=torch.int64, device=dev); vals = torch.empty(nb, dtype=torch.int32, device=dev)
pool = bench.make_pool(NG, G, dev, seed=7)
pa = bench.codes_to_ascii(pool)
goff = torch.arange(NG + 1, device=dev, dtype=torch.int64) * G
taxid = torch.from_numpy(leaves.astype(np.int32)).to(dev)
torch.cuda.synchronize()
hdr = ctx.build_table_device(pa.data_ptr(), goff.data_ptr(), NG, NG * G, taxid.data_ptr(), nb, flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), None)
torch.cuda.synchronize()
print("built %d keys in %.1f s" % (int(hdr[2]), time.time() - t0), flush=True)
del pa
reads = bench.gen_reads(pool, N, L, NG, G, dev, seed=43)
offsets = torch.arange(N + 1, device=dev, dtype=torch.int64) * L
del pool
t0 = time.time()
hf = flags.cpu().numpy().view(np.uint32); hk = keys.cpu().numpy().view(np.uint64); hv = vals.cpu().numpy().view(np.uint32)
del flags, keys, vals
torch.cuda.synchronize(); torch.cuda.empty_cache()
print("arrays on the host (%.0f GB) in %.1f s; free HBM %.0f GB" % ((hf.nbytes + hk.nbytes + hv.nbytes) / 1e9, time.time() - t0, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
import oracle_lib as O
S = 200_000
table = O.Table.wrap(int(hdr[0]), int(hdr[2]), int(hdr[1]), int(hdr[3]), hf, hk, hv)
tax = O.Taxonomy(pairs=[(int(c), int(p)) for c, p in enumerate(parent) if p != 0xFFFFFFFF and c != 0])
ho = offsets[:S + 1].cpu().numpy().astype(np.uint64); hb = reads[:S * L].cpu().numpy()
res = O.classify_batch(table, tax, K, hb, ho, nthreads=16)
if IDENT:
    ctx.set_minimizer_identity(IDENT)
if NBUCKETS:
    ctx.set_table_buckets(NBUCKETS)
out = [torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(3)]
for dbg in DBGS:
    ctx.debug_set(dbg)
    t0 = time.time()
    ctx.load_table(nb, hf, hk, hv, layout=bonsai_amd.LAYOUT_MINBUCKET)
    geo = ctx.table_geometry()
    print("dbg 0x%x: streamed load in %.1f s: %s geometry %s load %.3f" % (dbg, time.time() - t0, ctx.table_stats(), geo, ctx.table_stats()["n_keys"] / (10.0 * geo["buckets"])), flush=True)
    if ctx.table_warning():
        print("table warning:", ctx.table_warning(), flush=True)
    ctx.set_timing(False); ctx.set_timing(True)
    for _ in range(5):
        ctx.classify_device(reads.data_ptr(), offsets.data_ptr(), N, N * L, L, False, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), None, None, None)
    torch.cuda.synchronize()
    ms, cnt = ctx.timing_summary()
    if hasattr(ctx.L, "bns_debug_fetch_count"):     # -DBNS_COUNT_FETCHES build (BONSAI_AMD_LIB=...): counters since the table load
        import ctypes
        c2 = (ctypes.c_ulonglong * 8)()
        ctx.L.bns_debug_fetch_count.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        ctx.L.bns_debug_fetch_count(ctx.h, c2)
        d = [int(x) / 5.0 / N for x in c2]                  # (the counters restart with every table load)
        print("dbg 0x%x: per read: %.2f bucket fetches in %.2f probe passes, %.3f overflow lookups in %.3f rounds, %.3f quad iterations" % (dbg, d[0], d[1], d[2], d[3], d[4]), flush=True)
    print("dbg 0x%x: classify_kernel %.2f ms per 10 M reads = %.0f M reads/s, frac %.3f" % (dbg, ms / cnt, N / (ms / cnt) / 1e3, 1962 * N / (ms / cnt * 1e-3) / 8e12), flush=True)
    mism = int((out[0][:S].cpu().numpy().view(np.uint32) != res["taxon"]).sum() + (out[1][:S].cpu().numpy().view(np.uint32) != res["missing"]).sum() + (out[2][:S].cpu().numpy().view(np.uint32) != res["ambig"]).sum())
    print("dbg 0x%x: parity sample %d reads: %d mismatches, classified %.4f" % (dbg, S, mism, float((res["taxon"] != 0).mean())), flush=True)
