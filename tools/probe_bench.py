#!/usr/bin/env python
"""Standalone probe-kernel roofline (SURVEY 7 "hard parts"): pre-extracted canonical k-mers resident in
HBM, table far larger than L2 + Infinity Cache.  Reports lookups/s and algorithmic (16 B/lookup) GB/s."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=1024)
    ap.add_argument("--genome-len", type=int, default=1 << 18)
    ap.add_argument("--log2-buckets", type=int, default=29)
    ap.add_argument("--layout", choices=["bucket", "khash", "minbucket"], default="bucket")
    ap.add_argument("--bucket-slots-log2", type=int, default=0)
    ap.add_argument("--n", type=int, default=1 << 28)
    ap.add_argument("--hit-frac", type=float, default=0.7)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    import bonsai_amd
    ctx = bonsai_amd.Context(0)
    ctx.set_encoder(31, None, canonicalize=True)
    parent, leaves = bench.make_taxonomy(a.genomes)
    ctx.load_taxonomy(parent)
    nb = 1 << a.log2_buckets
    stream = torch.cuda.current_stream().cuda_stream
    flags = torch.empty(max(1, nb >> 4), dtype=torch.int32, device=dev)
    keys = torch.empty(nb, dtype=torch.int64, device=dev)
    vals = torch.empty(nb, dtype=torch.int32, device=dev)
    pool = bench.make_pool(a.genomes, a.genome_len, dev, seed=7)
    pa = bench.codes_to_ascii(pool)
    goff = torch.arange(a.genomes + 1, device=dev, dtype=torch.int64) * a.genome_len
    taxid = torch.from_numpy(leaves.astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    hdr = ctx.build_table_device(pa.data_ptr(), goff.data_ptr(), a.genomes, a.genomes * a.genome_len, taxid.data_ptr(), nb,
                                 flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), stream)
    # queries: present keys (random slots that are occupied) mixed with random 62-bit keys
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    n = a.n
    idx = torch.randint(0, nb, (n,), device=dev, generator=gen)
    q = keys[idx]
    present = ((flags[idx >> 4] >> ((idx & 15) << 1)) & 3) == 0          # empty slots hold 0: replace them, or 57 % of the queries
    rnd = torch.randint(0, 1 << 62, (n,), device=dev, generator=gen, dtype=torch.int64)   # would be one cached key
    use_rnd = ~present | (torch.rand(n, device=dev, generator=gen) > a.hit_frac / max(1e-9, float(hdr[2]) / nb))
    q = torch.where(use_rnd, rnd, q).contiguous()
    del idx, rnd, use_rnd, present
    layout = {"bucket": bonsai_amd.LAYOUT_BUCKET, "khash": bonsai_amd.LAYOUT_KHASH, "minbucket": bonsai_amd.LAYOUT_MINBUCKET}[a.layout]
    if a.bucket_slots_log2:
        ctx.set_bucket_slots_log2(a.bucket_slots_log2)
    ctx.load_table_device(nb, flags.data_ptr(), keys.data_ptr(), vals.data_ptr(), layout, stream)
    out_v = torch.empty(n, dtype=torch.int32, device=dev)
    out_f = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.probe_device(q.data_ptr(), n, out_v.data_ptr(), out_f.data_ptr(), stream)
    torch.cuda.synchronize()
    ctx.set_timing(True)
    for _ in range(a.iters):
        ctx.probe_device(q.data_ptr(), n, out_v.data_ptr(), out_f.data_ptr(), stream)
    torch.cuda.synchronize()
    s, c = ctx.timing_summary()
    ms = s / c
    info = ctx.table_info()
    print(json.dumps({"kernel": "probe_kernel", "layout": a.layout, "table_gb": info["device_bytes"] / 1e9, "n": n,
                      "hit_frac": float(out_f.float().mean().item()), "ms": ms, "lookups_per_s": n / (ms * 1e-3),
                      "alg_gbs_16B": n * 16 / (ms * 1e-3) / 1e9, "sector_gbs_64B": n * 64 / (ms * 1e-3) / 1e9,
                      "frac_of_8TBs_alg": n * 16 / (ms * 1e-3) / 8e12}))


if __name__ == "__main__":
    main()
