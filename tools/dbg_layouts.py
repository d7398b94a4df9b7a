"""debug: reload the clustered table many times (its placement differs from load to load) and compare with the plain bucket layout"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_properties as T
fx = T.big
fn = getattr(fx, "_fixture_function", None) or getattr(fx, "__wrapped__", None) or fx._get_wrapped_function()
gen = fn(); b = next(gen)
ctx = b["ctx"]
b["loaded"] = None
ref = [x.cpu().numpy() for x in T.run(b, 1)]
nb, flags, keys, vals = b["tab"]
hk = keys.cpu().numpy().view(np.uint64); hf = flags.cpu().numpy().view(np.uint32); hv = vals.cpu().numpy().view(np.uint32)
reads = b["reads"].cpu().numpy(); 
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    b["loaded"] = None
    got = [x.cpu().numpy() for x in T.run(b, 2)]
    d = np.nonzero((got[0] != ref[0]) | (got[1] != ref[1]) | (got[3] != ref[3]))[0]
    if d.size:
        bad += 1
        print("load", it, "geo", ctx.table_geometry(), "mismatching reads", d.size, d[:5], flush=True)
        r = int(d[0])
        print("  read", r, "minbucket", [int(got[i][r]) for i in range(4)], "bucket", [int(ref[i][r]) for i in range(4)])
        again = [x.cpu().numpy() for x in T.run(b, 2)]
        print("  same table again:", [int(again[i][r]) for i in range(4)])
        # which k-mers of the read does the table get wrong?  (probe entry point on the same table vs the khash arrays)
        import oracle_lib as O
        seq = bytes(reads[r * 150:(r + 1) * 150])
        km = O.encode(seq, 31, canon=True)
        v, f = ctx.probe(km)
        table = O.Table.wrap(nb, 0, 0, 0, hf, hk, hv)
        ev, ef = table.get_batch(km)
        w = np.nonzero((f != ef) | (v != ev))[0]
        print("  probe disagreements:", w.size, [(hex(int(km[i])), int(f[i]), int(ef[i]), int(v[i]), int(ev[i])) for i in w[:4]])
print("bad loads:", bad)
