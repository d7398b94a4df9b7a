"""ctypes loader for the C ABI declared in include/bonsai_amd.h.  Fails loudly when the
device library is missing: there is no CPU fallback in this package."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("BONSAI_AMD_LIB") or os.path.join(PKG, "lib", "libbonsai_amd.so")   # env override: A/B profiling builds

OK = 0
LAYOUT_KHASH, LAYOUT_BUCKET, LAYOUT_MINBUCKET = 0, 1, 2
SCORE_LEX, SCORE_ENTROPY_PATH, SCORE_ENTROPY_STRING = 0, 1, 2
TAX_ABSENT = 0xFFFFFFFF

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class BonsaiAmdError(RuntimeError):
    pass


# bns_classify_text (include/bonsai_amd.h): flags, status codes and the two structs
TEXT_FINAL, TEXT_TRIM_READNO, TEXT_DEVICE, TEXT_PARSE_ONLY, TEXT_DEFER = 1, 2, 4, 8, 16
TEXT_OK, TEXT_IRREGULAR, TEXT_NO_RECORD, TEXT_CAP = 0, 1, 2, 3
TEXT_WHY = {1: "CR", 2: "LEADING", 4: "AFTER_QUAL", 8: "QUAL_LEN", 16: "PLUS_RUN", 32: "LONG_RECORD", 64: "LINES"}


class TextOut(C.Structure):
    _fields_ = [("taxon", C.c_void_p), ("missing", C.c_void_p), ("ambig", C.c_void_p), ("n_hits", C.c_void_p),
                ("run_start", C.c_void_p), ("n_runs", C.c_void_p), ("seq_len", C.c_void_p), ("rec_pos", C.c_void_p),
                ("name_off", C.c_void_p), ("names", C.c_void_p), ("names_cap", C.c_uint64),
                ("run_tax", C.c_void_p), ("run_len", C.c_void_p), ("runs_cap", C.c_uint64),
                ("words", C.c_void_p), ("nmask", C.c_void_p)]


class GzResult(C.Structure):
    """bns_gz_result (include/bonsai_amd.h): what one bns_inflate_stream_device call took of a gzip stream"""
    _fields_ = [("text_bytes", C.c_uint64), ("end_bit", C.c_uint64), ("member_end", C.c_uint32), ("crc32", C.c_uint32),
                ("n_chunks", C.c_uint32), ("n_chained", C.c_uint32), ("status", C.c_uint32), ("stop_why", C.c_uint32)]


class TextInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("consumed", C.c_uint64 * 2), ("total_bases", C.c_uint64), ("names_bytes", C.c_uint64),
                ("n_runs_total", C.c_uint64), ("run_tax", C.POINTER(C.c_uint32)), ("run_len", C.POINTER(C.c_uint32)),
                ("status", C.c_int32), ("why", C.c_uint32), ("n_slices", C.c_uint32), ("n_launches", C.c_uint32),
                ("ms_parse", C.c_double), ("ms_classify", C.c_double)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise BonsaiAmdError(
            "device library %s is missing: build it with `python -m bonsai_amd.build` "
            "(hipcc --offload-arch=gfx950); bonsai_amd has no CPU fallback" % SO)
    L = C.CDLL(SO)
    sig = {
        "bns_version": (C.c_int, []),
        "bns_device_count": (C.c_int, []),
        "bns_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
        "bns_strerror": (C.c_char_p, [C.c_int]),
        "bns_last_error": (C.c_char_p, [vp]),
        "bns_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "bns_destroy": (None, [vp]),
        "bns_set_encoder": (C.c_int, [vp, C.c_uint32, u16p, C.c_int, C.c_int]),
        "bns_set_window": (C.c_int, [vp, C.c_uint32, C.c_int]),
        "bns_load_table": (C.c_int, [vp, C.c_uint64, u32p, u64p, u32p, C.c_int]),
        "bns_load_table_device": (C.c_int, [vp, C.c_uint64, vp, vp, vp, C.c_int, vp]),
        "bns_load_table_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_uint64, u32p, u64p, u32p, C.c_int]),
        "bns_set_bucket_slots_log2": (C.c_int, [vp, C.c_uint32]),
        "bns_set_table_buckets": (C.c_int, [vp, C.c_uint64]),
        "bns_set_minimizer_identity": (C.c_int, [vp, C.c_int]),
        "bns_set_table_fill": (C.c_int, [vp, C.c_int]),
        "bns_table_geometry": (C.c_int, [vp, u64p]),
        "bns_table_warning": (C.c_char_p, [vp]),
        "bns_rccl_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "bns_table_info": (C.c_int, [vp, u64p, u64p, C.POINTER(C.c_int)]),
        "bns_table_stats": (C.c_int, [vp, u64p]),
        "bns_set_minimizer_span": (C.c_int, [vp, C.c_uint32]),
        "bns_table_minimizer": (C.c_int, [vp, u32p, u64p]),
        "bns_load_taxonomy": (C.c_int, [vp, u32p, C.c_uint32]),
        "bns_classify_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_int, u32p, u32p, u32p, u32p, u32p]),
        "bns_classify_batch_runs": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_int, u32p, u32p, u32p, u32p, u64p, u32p,
                                              C.POINTER(u32p), C.POINTER(u32p), u64p]),
        "bns_classify_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                                vp, vp, vp, vp, vp, vp]),
        "bns_packed_words": (C.c_uint64, [C.c_uint64, C.c_uint64]),
        "bns_pack_reads": (C.c_int, [vp, u64p, C.c_uint64, u64p, u64p, u32p, C.c_uint64, u64p, C.c_int]),
        "bns_pack_reads_ptrs": (C.c_int, [vp, u32p, C.c_uint64, u64p, u64p, u64p, u32p, C.c_uint64, u64p, C.c_int]),
        "bns_classify_batch_packed": (C.c_int, [vp, u64p, u64p, u32p, C.c_uint64, u64p, C.c_uint64, C.c_int, u32p, u32p, u32p, u32p, u32p]),
        "bns_classify_batch_packed_runs": (C.c_int, [vp, u64p, u64p, u32p, C.c_uint64, u64p, C.c_uint64, C.c_int, u32p, u32p, u32p, u32p, u64p, u32p,
                                                     C.POINTER(u32p), C.POINTER(u32p), u64p]),
        "bns_classify_batch_packed_device": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                                       vp, vp, vp, vp, vp, vp]),
        "bns_encode_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, u64p, u32p]),
        "bns_encode_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp]),
        "bns_rolling_hash_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_uint32, C.c_int, u64p, u64p, u64p, u32p]),
        "bns_rolling_hash_windowed_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, u64p, u64p, u64p, u32p]),
        "bns_rolling_tables": (C.c_int, [C.c_uint64, C.c_uint64, u64p, u64p]),
        "bns_rolling_hash128_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_uint32, C.c_int, u64p, u64p, u64p, u32p]),
        "bns_rolling_hash128_windowed_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, u64p, u64p, u64p, u32p]),
        "bns_rolling_tables128": (C.c_int, [C.c_uint64, C.c_uint64, u64p, u64p]),
        "bns_for_each_hash_batch": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_uint32, C.c_int, u64p, u64p, u32p]),
        "bns_nthash_tables": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p]),
        "bns_probe": (C.c_int, [vp, u64p, C.c_uint64, u32p, u8p]),
        "bns_probe_device": (C.c_int, [vp, vp, C.c_uint64, vp, vp, vp]),
        "bns_resolve_batch": (C.c_int, [vp, u32p, u16p, u64p, C.c_uint64, u32p]),
        "bns_build_table_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp, vp, vp, u64p, vp]),
        "bns_set_timing": (C.c_int, [vp, C.c_int]),
        "bns_last_kernel_ms": (C.c_float, [vp]),
        "bns_timing_summary": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
        "bns_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "bns_host_free": (C.c_int, [vp, vp]),
        "bns_dev_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "bns_dev_free": (C.c_int, [vp, vp]),
        "bns_dev_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "bns_dev_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "bns_dev_sync": (C.c_int, [vp]),
        "bns_classify_text": (C.c_int, [vp, C.POINTER(vp), u64p, C.c_int, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(TextOut), C.POINTER(TextInfo)]),
        "bns_text_prefetch": (C.c_int, [vp, C.POINTER(vp), u64p, C.c_int]),
        "bns_text_finish": (C.c_int, [vp, C.POINTER(TextInfo)]),
        "bns_dev_copy_peer": (C.c_int, [vp, vp, vp, vp, C.c_size_t]),
        "bns_host_register": (C.c_int, [vp, vp, C.c_size_t]),
        "bns_host_unregister": (C.c_int, [vp, vp]),
        "bns_dev_copy": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "bns_inflater_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "bns_inflater_destroy": (None, [vp]),
        "bns_inflater_error": (C.c_char_p, [vp]),
        "bns_inflater_last_kernel_ms": (C.c_float, [vp]),
        "bns_inflater_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "bns_inflater_host_free": (C.c_int, [vp, vp]),
        "bns_inflate_members": (C.c_int, [vp, vp, C.c_uint64, u64p, u32p, u64p, u32p, C.c_uint64, vp, C.c_uint64, u32p, u32p]),
        "bns_inflate_members_device": (C.c_int, [vp, vp, C.c_uint64, u64p, u32p, u64p, u32p, C.c_uint64, vp, C.c_uint64, u32p, u32p]),
        "bns_inflate_stream_device": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint64, vp, C.POINTER(GzResult)]),
        "bns_inflate_stream_reserve": (C.c_int, [vp, C.c_uint64]),
        "bns_inflate_stream_prefetch": (C.c_int, [vp, vp, C.c_uint64]),
        "bns_inflate_stream_room": (C.c_int, [vp, C.c_uint32]),
        "bns_crc32_combine": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here == ABI drift; let it propagate
        fn.restype = res
        fn.argtypes = args
    L._bns_signatures = sig
    _lib = L
    return L
