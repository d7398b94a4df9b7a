#!/usr/bin/env python
"""When do the 8192 wavefronts of one classify_kernel launch finish?  (measurement build -DBNS_WAVE_TIMES; run with
BONSAI_AMD_LIB=bonsai_amd/lib/libT.so)  Prints the spread of the finish times as fractions of the kernel's duration."""
import ctypes, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, bonsai_amd

sys.argv = [sys.argv[0], "--no-cpu", "--no-probe", "--steps", "3", "--warmup", "1"] + sys.argv[1:]
# run the bench in-process, then read the stamps of its last launch
import builtins
ctxs = []
orig = bonsai_amd.Context
class Keep(orig):
    def __init__(self, *a, **k):
        super().__init__(*a, **k); ctxs.append(self)
    def close(self):
        t = np.zeros(16384, dtype=np.uint64)
        self.L.bns_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.L.bns_debug_wave_times(self.h, t.ctypes.data)
        t0, t1 = t[0::2].astype(np.int64), t[1::2].astype(np.int64)
        ok = t1 > 0
        base, end = t0[ok].min(), t1[ok].max()
        fin = (t1[ok] - base) / float(end - base)
        st = (t0[ok] - base) / float(end - base)
        print(json.dumps({"waves": int(ok.sum()), "start_frac_max": float(st.max()), "finish_frac_percentiles": {str(p): float(np.percentile(fin, p)) for p in (0, 1, 5, 25, 50, 75, 95, 99, 100)},
                          "mean_alive_frac": float(((t1[ok] - t0[ok]) / float(end - base)).mean())}))
        # by XCD (block index mod 8 is the usual round-robin) and by CU-ish groups
        blk = np.arange(8192)[ok] // 4
        for x in range(8):
            sel = (blk % 8) == x
            print("xcd-ish %d: median finish %.3f  p95 %.3f" % (x, float(np.median(fin[sel])), float(np.percentile(fin[sel], 95))))
        super().close()
bonsai_amd.Context = Keep
bench.main()
