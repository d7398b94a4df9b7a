#!/usr/bin/env python
"""Golden k-mer streams for SURVEY 8a rows 1-2 from REFERENCE code (build container only):
    python tests/golden/make_golden_stream.py      ->  tests/golden/stream_ref.npz   (data only, committed)

oracle/_ref/libbns_ref.so compiles the reference's include/bonsai/rhtraits.h + alphabet.h where they lie, so the symbol table
(DNA4, the Encoder's lutptr), rhmask<u64>(DNA, k) and mul(DNA) are the reference's own; the loop of encoder.h:246-271 that
drives them is restated in oracle/ref_harness.cpp (Encoder<> itself needs the un-vendored schism::Schismatic), and the
canonical form is the reference's canonical_representation.  "reference LUT / mask / canonicalisation + restated loop".
What is frozen here: the 256-entry table, the masks for k = 1..32, the forward and canonical 31-mer streams of phiX
(reference test fixture; 5356 k-mers, test/encoding.cpp:122) and the streams of crafted 7-bit strings for several k.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

KS = (1, 5, 16, 21, 31, 32)


def crafted(rng):
    acgt = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
    mixed = lambda n: bytes(rng.choice(np.frombuffer(b"ACGTacgtNnURYKMSWBDHV-", dtype=np.uint8), size=n,
                                       p=np.array([10] * 4 + [3] * 4 + [1] * 14) / 66.0))
    out = [b"", b"A", b"ACGT", acgt(30), acgt(31), acgt(32), acgt(33), acgt(64), acgt(65), acgt(200),
           acgt(40).lower(), b"N" * 50, acgt(31) + b"N", b"N" + acgt(31), acgt(31) + b"N" + acgt(31), acgt(30) + b"N" + acgt(30),
           acgt(35) + b"NN" + acgt(35) + b"n" + acgt(10), b"A" * 70, b"T" * 70, b"ACGT" * 20, acgt(50) + b"U" + acgt(50),
           acgt(20) + b"RYKM" + acgt(45), b"acgtNACGT" * 12, mixed(300), mixed(300), mixed(1000), acgt(33) + b"\x00" + acgt(33),
           acgt(40) + b" \t\n" + acgt(40), acgt(2100), mixed(2500)]
    return out


def stream(R, s, k, canon):
    out = np.zeros(max(1, len(s)), dtype=np.uint64)
    n = R.ref_kmer_stream(s, len(s), k, int(canon), out.ctypes.data_as(O.u64p))
    return out[:n].copy()


def main():
    R = O.ref()
    assert R is not None and hasattr(R, "ref_kmer_stream"), "build oracle/_ref first (make -C oracle)"
    R.ref_kmer_stream.restype = C.c_uint64
    R.ref_kmer_stream.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, C.c_int, O.u64p]
    R.ref_rhmask_dna.restype = C.c_uint64
    R.ref_rhmul_dna.restype = C.c_uint64
    lut = np.zeros(256, dtype=np.int8)
    assert R.ref_dna4_lut(lut.ctypes.data_as(C.c_void_p)) == 256
    assert R.ref_rhmul_dna() == 4
    out = {"lut": lut, "masks": np.array([R.ref_rhmask_dna(k) for k in range(1, 33)], dtype=np.uint64), "ks": np.array(KS, dtype=np.uint32)}
    _, phix = O.read_fasta(os.path.join(HERE, "phix.fa"))[0]
    phix = phix if isinstance(phix, bytes) else phix.encode()
    out["phix_fw31"] = stream(R, phix, 31, False)
    out["phix_cn31"] = stream(R, phix, 31, True)
    assert out["phix_fw31"].size == 5356 and np.unique(out["phix_fw31"]).size == 5356      # test/encoding.cpp:122
    rng = np.random.default_rng(20260929)
    strs = crafted(rng)
    out["str_bytes"] = np.frombuffer(b"".join(strs), dtype=np.uint8).copy()
    out["str_offs"] = np.cumsum([0] + [len(s) for s in strs]).astype(np.uint64)
    for k in KS:
        for canon in (0, 1):
            parts = [stream(R, s, k, canon) for s in strs]
            out["s_k%d_c%d" % (k, canon)] = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
            out["n_k%d_c%d" % (k, canon)] = np.array([p.size for p in parts], dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "stream_ref.npz"), **out)
    print("wrote stream_ref.npz: %d strings, phiX %d k-mers" % (len(strs), out["phix_fw31"].size))


if __name__ == "__main__":
    main()
