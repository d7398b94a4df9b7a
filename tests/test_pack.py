"""bns_pack_reads: the host-side packer of the packed-read entry points (include/bonsai_amd.h) -- pure host code in the C-ABI
library, so it runs in the CPU tier.  Checked against a character-level restatement of the format: 2-bit codes A0 C1 G2 T3
(alphabet.h:128: either case, everything else invalid), 32 bases per word, first base in the top two bits, read r's words at
(offsets[r] >> 5) + r, invalid bases flagged in a sparse (word index, 32-bit mask) list."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def A():
    from bonsai_amd.build import build_device_library
    build_device_library()
    import bonsai_amd
    return bonsai_amd


def restate(seqs, offs):
    words, bad = {}, {}
    for r, s in enumerate(seqs):
        wb = (int(offs[r]) >> 5) + r
        for j in range(0, len(s), 32):
            w = b = 0
            for i, c in enumerate(s[j:j + 32]):
                cu = chr(c & 0xDF) if c < 128 else "?"
                if cu in "ACGT" and (c | 0x20) in b"acgt":
                    w |= "ACGT".index(cu) << (62 - 2 * i)
                else:
                    b |= 1 << (31 - i)
            words[wb + j // 32] = w
            if b:
                bad[wb + j // 32] = b
    return words, bad


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_pack_reads_matches_the_format(A, threads):
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"ACGTacgtNnRYU-*\x00\xff@[`{", dtype=np.uint8)
    p = np.array([30] * 4 + [6] * 4 + [1] * (alpha.size - 8), dtype=float)
    lens = list(rng.integers(0, 400, size=500)) + [0, 1, 31, 32, 33, 63, 64, 65, 2047, 2048, 2049, 5000] + [150] * 9000
    seqs = [bytes(rng.choice(alpha, size=int(L), p=p / p.sum())) for L in lens]
    seqs[40] = b"ACGT" * 40                                   # reads without a single invalid base
    seqs[41] = b"acgt" * 33
    bases, offs = A.concat_reads(seqs)
    words, bw, bm = A.pack_reads(bases, offs, threads=threads)
    assert words.size == (int(offs[-1]) >> 5) + len(seqs) + 1
    ew, eb = restate(seqs, offs)
    for i, w in ew.items():
        assert int(words[i]) == w, i
    assert {int(i): int(m) for i, m in zip(bw, bm)} == eb
    assert np.all(np.diff(bw.astype(np.int64)) > 0)           # sorted by word, whatever the thread count


def test_pack_reads_bad_list_protocol(A):
    import ctypes as C
    L = A.load()
    seqs = [b"ACGTN" * 20, b"NNNN", b"ACGT" * 10]
    bases, offs = A.concat_reads(seqs)
    words = np.zeros(int(L.bns_packed_words(int(offs[-1]), len(seqs))), dtype=np.uint64)
    nb = C.c_uint64()
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    rc = L.bns_pack_reads(bases.ctypes.data, offs.ctypes.data_as(u64p), len(seqs), words.ctypes.data_as(u64p), None, None, 0, C.cast(C.byref(nb), u64p), 1)
    assert rc == -1 and nb.value == 5                          # too little room: BNS_ERR_ARG and the room needed (4 words of read 0 + 1)
    bw = np.zeros(5, dtype=np.uint64); bm = np.zeros(5, dtype=np.uint32)
    rc = L.bns_pack_reads(bases.ctypes.data, offs.ctypes.data_as(u64p), len(seqs), words.ctypes.data_as(u64p), bw.ctypes.data_as(u64p),
                          bm.ctypes.data_as(u32p), 5, C.cast(C.byref(nb), u64p), 1)
    assert rc == 0 and nb.value == 5 and int(bm[4]) == 0xF0000000
    clean, cbw, cbm = A.pack_reads(*A.concat_reads([b"ACGT" * 100]))
    assert cbw.size == 0 and int(clean[0]) == int("00011011" * 8, 2)


@pytest.mark.parametrize("threads", [1, 4])
def test_pack_reads_zeroes_the_slack_words(A, threads):
    """the words no read covers (the slack between two reads, the word behind the last read) are zeroed: a recycled buffer gives
    the same image as a fresh one"""
    import ctypes as C
    L = A.load()
    rng = np.random.default_rng(3)
    lens = [0, 1, 31, 32, 33, 64, 0, 0, 150, 5] + list(rng.integers(0, 300, size=20000))
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=int(n))) for n in lens]
    bases, offs = A.concat_reads(seqs)
    fresh, _, _ = A.pack_reads(bases, offs, threads=threads)
    dirty = np.full(fresh.size, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    bw = np.zeros(1 << 20, dtype=np.uint64); bm = np.zeros(1 << 20, dtype=np.uint32); nb = C.c_uint64()
    rc = L.bns_pack_reads(bases.ctypes.data, offs.ctypes.data_as(u64p), len(seqs), dirty.ctypes.data_as(u64p), bw.ctypes.data_as(u64p),
                          bm.ctypes.data_as(u32p), bw.size, C.cast(C.byref(nb), u64p), threads)
    assert rc == 0
    assert np.array_equal(dirty, fresh)
