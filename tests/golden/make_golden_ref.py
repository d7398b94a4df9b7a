#!/usr/bin/env python
"""Generate golden vectors from the REFERENCE's own code (oracle/_ref/libbns_ref.so = the reference's
include/bonsai/khash64.h and linear/linear.h compiled where they lie under /root/reference).
Run in the build container only:  python tests/golden/make_golden_ref.py
Outputs (data only, committed): tests/golden/khash_ref.npz, tests/golden/counter_ref.npz
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def table_dump(R, h):
    hdr = np.zeros(4, dtype=np.uint64)
    f = O.u32p(); k = O.u64p(); v = O.u32p()
    R.ref_khc_info(h, hdr.ctypes.data_as(O.u64p), C.byref(f), C.byref(k), C.byref(v))
    nb = int(hdr[0])
    fs = 1 if nb < 16 else nb >> 4
    flags = np.ctypeslib.as_array(f, shape=(fs,)).copy()
    keys = np.ctypeslib.as_array(k, shape=(nb,)).copy()
    vals = np.ctypeslib.as_array(v, shape=(nb,)).copy()
    # zero the payload of empty/deleted slots as the reference's writer does (util.h:282-284)
    idx = np.arange(nb)
    st = (flags[idx >> 4] >> ((idx & 15) << 1)) & 3
    keys[st != 0] = 0
    vals[st != 0] = 0
    return hdr, flags, keys, vals


def main():
    R = O.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20260928)
    out = {}
    # (a) 128 keys of reference test/util.cpp:10-15, (b) 50k random 62-bit keys incl. key 0, (c) after deletes
    for name, keys in (("t128", np.array([(i << 14) | (i + 2) for i in range(128)], dtype=np.uint64)),
                       ("r50k", np.unique(np.concatenate([rng.integers(0, 1 << 62, size=50000, dtype=np.uint64),
                                                          np.array([0, 1, (1 << 62) - 1], dtype=np.uint64)])))):
        rng.shuffle(keys)
        vals = rng.integers(1, 1 << 31, size=keys.size, dtype=np.uint32)
        if name == "t128":
            vals = np.arange(128, dtype=np.uint32)
        h = R.ref_khc_new()
        R.ref_khc_insert(h, keys.ctypes.data_as(O.u64p), vals.ctypes.data_as(O.u32p), keys.size)
        hdr, f, k, v = table_dump(R, h)
        q = np.concatenate([keys, rng.integers(0, 1 << 62, size=keys.size, dtype=np.uint64)])
        rng.shuffle(q)
        qv = np.zeros(q.size, dtype=np.uint32); qf = np.zeros(q.size, dtype=np.uint8)
        R.ref_khc_get_batch(h, q.ctypes.data_as(O.u64p), q.size, qv.ctypes.data_as(O.u32p), qf.ctypes.data_as(O.u8p))
        qi = np.array([R.ref_khc_get(h, int(x)) for x in q[:2000]], dtype=np.uint64)
        out.update({name + "_ins_keys": keys, name + "_ins_vals": vals, name + "_hdr": hdr, name + "_flags": f,
                    name + "_keys": k, name + "_vals": v, name + "_q": q, name + "_qv": qv, name + "_qf": qf,
                    name + "_qslot": qi})
        if name == "r50k":
            dele = keys[::7].copy()
            for x in dele:
                R.ref_khc_del_key(h, int(x))
            hdr2, f2, k2, v2 = table_dump(R, h)
            qv2 = np.zeros(q.size, dtype=np.uint32); qf2 = np.zeros(q.size, dtype=np.uint8)
            R.ref_khc_get_batch(h, q.ctypes.data_as(O.u64p), q.size, qv2.ctypes.data_as(O.u32p), qf2.ctypes.data_as(O.u8p))
            out.update({"del_keys": dele, "del_hdr": hdr2, "del_flags": f2, "del_keys_arr": k2, "del_vals": v2,
                        "del_qv": qv2, "del_qf": qf2})
        R.ref_khc_free(h)
    out["wang_in"] = np.concatenate([np.arange(16, dtype=np.uint64), rng.integers(0, 1 << 63, size=64, dtype=np.uint64) * 2 + 1])
    out["wang_out"] = np.array([R.ref_wang64(int(x)) for x in out["wang_in"]], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "khash_ref.npz"), **out)

    adds = rng.choice(np.array([0, 1, 7, 1001, 1002, 0xFFFFFFFF, 55], dtype=np.uint32), size=300)
    ko = np.zeros(16, dtype=np.uint32); vo = np.zeros(16, dtype=np.uint16)
    n = R.ref_counter(adds.ctypes.data_as(O.u32p), adds.size, ko.ctypes.data_as(O.u32p), vo.ctypes.data_as(O.u16p))
    wrap = np.full(70000, 9, dtype=np.uint32)           # u16 wrap: 70000 mod 65536 = 4464
    kw = np.zeros(4, dtype=np.uint32); vw = np.zeros(4, dtype=np.uint16)
    nw = R.ref_counter(wrap.ctypes.data_as(O.u32p), wrap.size, kw.ctypes.data_as(O.u32p), vw.ctypes.data_as(O.u16p))
    ins = rng.integers(0, 12, size=60).astype(np.uint32)
    so = np.zeros(60, dtype=np.uint32)
    ns = R.ref_linear_set(ins.ctypes.data_as(O.u32p), ins.size, so.ctypes.data_as(O.u32p))
    np.savez_compressed(os.path.join(HERE, "counter_ref.npz"), adds=adds, keys=ko[:n], vals=vo[:n],
                        wrap_keys=kw[:nw], wrap_vals=vw[:nw], set_ins=ins, set_out=so[:ns])
    print("wrote khash_ref.npz, counter_ref.npz")


if __name__ == "__main__":
    main()
