"""Host-side file formats of the classify path through libbns_host.so (C++: bonsai_amd/csrc/host):
bns.db (database.h:33-102), nodes.dmp (util.h:766-785), FASTA/FASTQ batches (kseq_declare.h:106-175) and
the Kraken / FASTQ output records (classifier.h:45-129).  No GPU needed."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(PKG, "lib", "libbns_host.so")
_lib = None


class HostIOError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise HostIOError("%s is missing: run `make -C bonsai_amd/csrc/host` (or __graft_entry__.build())" % SO)
        L = C.CDLL(SO)
        L.bnsh_last_error.restype = C.c_char_p
        L.bnsh_db_open.restype = C.c_void_p; L.bnsh_db_open.argtypes = [C.c_char_p]
        L.bnsh_db_close.argtypes = [C.c_void_p]
        L.bnsh_db_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_int), C.POINTER(C.c_uint16)]
        for n, t in (("flags", C.c_uint32), ("keys", C.c_uint64), ("vals", C.c_uint32)):
            f = getattr(L, "bnsh_db_" + n); f.restype = C.POINTER(t); f.argtypes = [C.c_void_p]
        L.bnsh_db_write.restype = C.c_int; L.bnsh_db_write.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.bnsh_db_from_arrays.restype = C.c_void_p
        L.bnsh_db_from_arrays.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bnsh_parent_map.restype = C.c_int
        L.bnsh_parent_map.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
        L.bnsh_free.argtypes = [C.c_void_p]
        L.bnsh_parse_spacing.restype = C.c_int; L.bnsh_parse_spacing.argtypes = [C.c_char_p, C.c_uint, C.c_void_p, C.c_int]
        L.bnsh_read_fastx.restype = C.c_int
        L.bnsh_read_fastx.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.bnsh_set_bgzf_device.argtypes = [C.c_int]
        L.bnsh_genome_name.restype = C.c_size_t; L.bnsh_genome_name.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.bnsh_get_taxid.restype = C.c_int; L.bnsh_get_taxid.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32)]
        L.bnsh_kraken_line.restype = C.c_size_t
        L.bnsh_kraken_line.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
        L.bnsh_fastq_record.restype = C.c_size_t
        L.bnsh_fastq_record.argtypes = [C.c_char_p] * 6 + [C.c_uint32] * 3 + [C.c_void_p, C.c_uint32, C.c_int, C.c_char_p, C.c_size_t]
        L.bnsh_encoder_from_str.restype = C.c_long
        L.bnsh_encoder_from_str.argtypes = [C.c_uint, C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.bnsh_encoder_hash_from_str.restype = C.c_long
        L.bnsh_encoder_hash_from_str.argtypes = [C.c_uint, C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def encoder_hash_from_str(seq, k, gaps=None, canon=True, w=0, hash_k=0):
    """the C++ host class: bns::Encoder(k, gaps, canon, device 0, w).for_each_hash(func, str, len, hash_k), collected"""
    if isinstance(seq, str):
        seq = seq.encode()
    g = np.ascontiguousarray(gaps, dtype=np.uint16) if gaps is not None else None
    out = np.zeros(len(seq) + 1, dtype=np.uint64)
    n = lib().bnsh_encoder_hash_from_str(k, g.ctypes.data if g is not None else None, int(canon), w, hash_k, seq, len(seq), out.ctypes.data, out.size)
    if n < 0:
        raise HostIOError(lib().bnsh_last_error().decode())
    return out[:n].copy()


def encoder_from_str(seq, k, gaps=None, canon=True, w=0, score=0):
    """the C++ host class: bns::Encoder(k, gaps, canon, device 0, w, score).for_each(func, str, len), collected"""
    if isinstance(seq, str):
        seq = seq.encode()
    g = np.ascontiguousarray(gaps, dtype=np.uint16) if gaps is not None else None
    out = np.zeros(len(seq) + 1, dtype=np.uint64)
    n = lib().bnsh_encoder_from_str(k, g.ctypes.data if g is not None else None, int(canon), w, score, seq, len(seq), out.ctypes.data, out.size)
    if n < 0:
        raise HostIOError(lib().bnsh_last_error().decode())
    return out[:n].copy()


def read_db(path):
    """-> dict(k, w, gaps, n_buckets, n_occupied, size, upper_bound, flags, keys, vals, spacing_width)"""
    L = lib()
    h = L.bnsh_db_open(path.encode())
    if not h:
        raise HostIOError(L.bnsh_last_error().decode())
    try:
        k = C.c_uint32(); w = C.c_uint32(); hdr = (C.c_uint64 * 4)(); sw = C.c_int(); gaps = (C.c_uint16 * 64)()
        L.bnsh_db_info(h, C.byref(k), C.byref(w), hdr, C.byref(sw), gaps)
        nb = hdr[0]
        fs = 1 if nb < 16 else nb >> 4
        return {"k": k.value, "w": w.value, "gaps": np.array(gaps[:k.value - 1], dtype=np.uint16), "n_buckets": nb,
                "n_occupied": hdr[1], "size": hdr[2], "upper_bound": hdr[3], "spacing_width": sw.value,
                "flags": np.ctypeslib.as_array(L.bnsh_db_flags(h), shape=(fs,)).copy(),
                "keys": np.ctypeslib.as_array(L.bnsh_db_keys(h), shape=(nb,)).copy(),
                "vals": np.ctypeslib.as_array(L.bnsh_db_vals(h), shape=(nb,)).copy()}
    finally:
        L.bnsh_db_close(h)


def write_db(path, k, w, gaps, header4, flags, keys, vals, spacing_width=1):
    L = lib()
    g = np.ascontiguousarray(gaps if gaps is not None else np.zeros(max(0, k - 1)), dtype=np.uint16)
    hdr = np.ascontiguousarray(header4, dtype=np.uint64)
    flags = np.ascontiguousarray(flags, dtype=np.uint32); keys = np.ascontiguousarray(keys, dtype=np.uint64)
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    h = L.bnsh_db_from_arrays(k, w, g.ctypes.data, hdr.ctypes.data, flags.ctypes.data, keys.ctypes.data, vals.ctypes.data)
    try:
        if L.bnsh_db_write(h, path.encode(), spacing_width) != 0:
            raise HostIOError(L.bnsh_last_error().decode())
    finally:
        L.bnsh_db_close(h)


def read_nodes_dmp(path):
    L = lib()
    out = C.POINTER(C.c_uint32)(); n = C.c_uint32()
    if L.bnsh_parent_map(path.encode(), C.byref(out), C.byref(n)) != 0:
        raise HostIOError(L.bnsh_last_error().decode())
    arr = np.ctypeslib.as_array(out, shape=(n.value,)).copy()
    L.bnsh_free(out)
    return arr


def genome_name(header):
    """get_taxid's name extraction (util.h:898-929) from a header line without '>'."""
    buf = C.create_string_buffer(4096)
    n = lib().bnsh_genome_name(header.encode(), buf, 4096)
    return buf.raw[:n].decode()


def get_taxid(genome_path, seq2tax_path):
    out = C.c_uint32()
    if lib().bnsh_get_taxid(genome_path.encode(), seq2tax_path.encode(), C.byref(out)) != 0:
        raise HostIOError(lib().bnsh_last_error().decode())
    return out.value


def parse_spacing(s, k):
    out = np.zeros(64, dtype=np.uint16)
    n = lib().bnsh_parse_spacing(s.encode() if s else None, k, out.ctypes.data, 64)
    return out[:n].copy()


def read_fastx(path1, path2=None, chunk_size=1 << 20, block_bytes=0):
    """-> (list of (name, comment, seq, qual) bytes tuples, number of bseq_read chunks).
    block_bytes shrinks the reader's text blocks (tests: make records cross block boundaries)."""
    L = lib()
    blob = C.c_void_p(); ln = C.c_size_t(); ch = C.c_int()
    L.bnsh_read_fastx_blk.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_size_t,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    if L.bnsh_read_fastx_blk(path1.encode(), path2.encode() if path2 else None, chunk_size, block_bytes,
                             C.byref(blob), C.byref(ln), C.byref(ch)) != 0:
        raise HostIOError(L.bnsh_last_error().decode())
    raw = C.string_at(blob.value, ln.value)
    L.bnsh_free(blob)
    recs = [tuple(line.split(b"\x1f")) for line in raw.split(b"\n") if line]
    return recs, ch.value


def read_fastx_par(path, chunk_size=1 << 20, parser_threads=2, segment_bytes=0, cuts=None, path2=None):
    """ChunkSource over one plain file (or a pair of files, mates interleaved) -> (records as read_fastx gives them, stretches, fell_back)"""
    L = lib()
    blob = C.c_void_p(); ln = C.c_size_t(); info = (C.c_int * 2)()
    L.bnsh_read_fastx_par.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_size_t), C.c_void_p]
    ca = np.ascontiguousarray(cuts, dtype=np.uint64) if cuts is not None else None
    if L.bnsh_read_fastx_par(path.encode(), path2.encode() if path2 else None, chunk_size, parser_threads, segment_bytes, ca.ctypes.data if ca is not None else None,
                             ca.size if ca is not None else 0, C.byref(blob), C.byref(ln), info) != 0:
        raise HostIOError(L.bnsh_last_error().decode())
    raw = C.string_at(blob.value, ln.value)
    L.bnsh_free(blob)
    recs = [tuple(line.split(b"\x1f")) for line in raw.split(b"\n") if line]
    return recs, info[0], bool(info[1])


def find_cut_points(path, segment_bytes):
    L = lib()
    out = np.zeros(4096, dtype=np.uint64)
    L.bnsh_find_cut_points.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int]
    n = L.bnsh_find_cut_points(path.encode(), segment_bytes, out.ctypes.data, out.size)
    return out[:n].copy()


def kraken_line(name, l_seq, taxon, missing, ambig, hits):
    hits = np.ascontiguousarray(hits, dtype=np.uint32)
    cap = 128 + len(name) + 16 * max(1, hits.size)
    buf = C.create_string_buffer(cap)
    n = lib().bnsh_kraken_line(name.encode(), l_seq, taxon, missing, ambig, hits.ctypes.data, hits.size, buf, cap)
    return buf.raw[:n]


def fastq_record(m1, m2, taxon, missing, ambig, hits, verbose):
    """m1/m2 = (name, seq, qual-or-None) byte tuples; m2 None for single-end."""
    hits = np.ascontiguousarray(hits, dtype=np.uint32)
    cap = 512 + 4 * (len(m1[1]) + (len(m2[1]) if m2 else 0)) + 32 * max(1, hits.size)
    buf = C.create_string_buffer(cap)
    a = [m1[0], m1[1], m1[2]] + ([m2[0], m2[1], m2[2]] if m2 else [None, None, None])
    n = lib().bnsh_fastq_record(*a, taxon, missing, ambig, hits.ctypes.data, hits.size, int(verbose), buf, cap)
    return buf.raw[:n]
