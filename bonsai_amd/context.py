"""Host-side mirror of the reference's classify objects over the C ABI.

Context bundles what bin/bonsai.cpp:149-157 builds before process_dataset: the Database's khash
(database.h:33-56), the Spacer/Encoder (classifier.h:155-166) and the parent map (util.h:766-785).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BonsaiAmdError, u8p, u16p, u32p, u64p, vp


def _p(a, t):
    return a.ctypes.data_as(t)


def concat_reads(seqs):
    """list of bytes/str -> (uint8 bases, uint64 offsets[n+1])"""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return bases, offsets


def pack_reads(bases, offsets, threads=1):
    """bns_pack_reads (host only, no GPU needed): ASCII batch -> (words uint64, bad_word uint64, bad_mask uint32): the 2-bit image
    the classify kernel works on and the sparse list of words that hold a base other than A/C/G/T"""
    L = _lib.load()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    words = np.zeros(int(L.bns_packed_words(int(offsets[-1]) if n >= 0 else 0, n)), dtype=np.uint64)
    cap = 1024
    while True:
        bw = np.zeros(cap, dtype=np.uint64); bm = np.zeros(cap, dtype=np.uint32); nb = C.c_uint64()
        rc = L.bns_pack_reads(bases.ctypes.data, _p(offsets, u64p), n, _p(words, u64p), _p(bw, u64p), _p(bm, u32p), cap,
                              C.cast(C.byref(nb), u64p), int(threads))
        if rc == 0:
            return words, bw[:nb.value].copy(), bm[:nb.value].copy()
        if nb.value <= cap:
            raise BonsaiAmdError("bns_pack_reads: %s" % L.bns_strerror(rc).decode())
        cap = int(nb.value)


class Context:
    def __init__(self, device=0):
        self.L = _lib.load()
        h = vp()
        rc = self.L.bns_create(device, C.byref(h))
        if rc != 0:
            raise BonsaiAmdError("bns_create(%d): %s" % (device, self.L.bns_strerror(rc).decode()))
        self.h = h
        self.k = None

    def close(self):
        if getattr(self, "h", None):
            self.L.bns_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise BonsaiAmdError("%s: %s (%s)" % (what, self.L.bns_strerror(rc).decode(),
                                                   self.L.bns_last_error(self.h).decode()))

    # ---- configuration
    def set_encoder(self, k, gaps=None, canonicalize=True, spaced_intended=True):
        g = None
        gp = None
        if gaps is not None:
            g = np.ascontiguousarray(gaps, dtype=np.uint16)
            if g.size != k - 1:
                raise ValueError("gaps must have k-1 entries")
            gp = _p(g, u16p)
        self._chk(self.L.bns_set_encoder(self.h, k, gp, int(canonicalize), int(spaced_intended)), "bns_set_encoder")
        self.k = k

    def set_window(self, w, score=_lib.SCORE_LEX):
        """Spacer window + score for minimizer selection (encode / build only; classify is always w = k)."""
        self._chk(self.L.bns_set_window(self.h, w, score), "bns_set_window")

    def set_bucket_slots_log2(self, lg):
        self._chk(self.L.bns_set_bucket_slots_log2(self.h, lg), "bns_set_bucket_slots_log2")

    def set_table_buckets(self, n):
        """clustered table: exact number of home buckets (0 = sized from the key count)"""
        self._chk(self.L.bns_set_table_buckets(self.h, n), "bns_set_table_buckets")

    def set_minimizer_identity(self, bits):
        """clustered table: 32 / 52-bit minimizer identity, 0 = chosen from the key count"""
        self._chk(self.L.bns_set_minimizer_identity(self.h, bits), "bns_set_minimizer_identity")

    def set_table_fill(self, mode):
        """clustered table: 0 = the loader decides, 1 = keys in arrival order, 2 = group-aware fill"""
        self._chk(self.L.bns_set_table_fill(self.h, mode), "bns_set_table_fill")

    def table_geometry(self):
        g = (C.c_uint64 * 8)()
        self._chk(self.L.bns_table_geometry(self.h, g), "bns_table_geometry")
        return {"buckets": g[0], "m": g[1], "identity_bits": g[2], "spilled_keys": g[3], "span": g[4], "overflow_keys": g[5], "group_fill": g[6]}

    def table_warning(self):
        return self.L.bns_table_warning(self.h).decode()

    def debug_set(self, bits):
        """test / profiling switches (BNS_DBG_* in bns_api.hip); not part of the public header"""
        self.L.bns_debug_set.argtypes = [vp, C.c_int]
        self.L.bns_debug_set.restype = C.c_int
        self._chk(self.L.bns_debug_set(self.h, bits), "bns_debug_set")

    def load_table(self, n_buckets, flags, keys, vals, layout=_lib.LAYOUT_MINBUCKET):
        flags = np.ascontiguousarray(flags, dtype=np.uint32)
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vals = np.ascontiguousarray(vals, dtype=np.uint32)
        if keys.size != n_buckets or vals.size != n_buckets or flags.size != max(1, n_buckets >> 4):
            raise ValueError("khash array sizes do not match n_buckets")
        self._chk(self.L.bns_load_table(self.h, n_buckets, _p(flags, u32p), _p(keys, u64p), _p(vals, u32p), layout),
                  "bns_load_table")

    def load_table_device(self, n_buckets, d_flags, d_keys, d_vals, layout=_lib.LAYOUT_MINBUCKET, stream=None):
        self._chk(self.L.bns_load_table_device(self.h, n_buckets, d_flags, d_keys, d_vals, layout, stream),
                  "bns_load_table_device")

    def table_info(self):
        nk = C.c_uint64(); nb = C.c_uint64(); ly = C.c_int()
        self._chk(self.L.bns_table_info(self.h, C.byref(nk), C.byref(nb), C.byref(ly)), "bns_table_info")
        return {"n_keys": nk.value, "device_bytes": nb.value, "layout": ly.value}

    def table_stats(self):
        st = (C.c_uint64 * 4)()
        self._chk(self.L.bns_table_stats(self.h, st), "bns_table_stats")
        return {"n_keys": st[0], "n_overflow_keys": st[1], "main_bytes": st[2], "overflow_bytes": st[3]}

    def set_minimizer_span(self, span):
        """clustered table, contiguous seeds: minimizer window k - m; 0 = chosen from the db at load, 8 / 11 / 15 fix it"""
        self._chk(self.L.bns_set_minimizer_span(self.h, span), "bns_set_minimizer_span")

    def table_minimizer(self):
        m = C.c_uint32(); sp = C.c_uint64()
        self._chk(self.L.bns_table_minimizer(self.h, C.byref(m), C.byref(sp)), "bns_table_minimizer")
        return {"m": m.value, "spilled_keys": sp.value}

    def load_taxonomy(self, parent):
        parent = np.ascontiguousarray(parent, dtype=np.uint32)
        self._chk(self.L.bns_load_taxonomy(self.h, _p(parent, u32p), parent.size), "bns_load_taxonomy")

    # ---- hot path (host buffers)
    def classify(self, bases, offsets, paired=False, want_hits=False):
        """classify_seqs (classifier.h:269-287) over one batch; returns dict of per-unit arrays."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = offsets.size - 1
        n_units = n_reads // (2 if paired else 1)
        taxon = np.zeros(n_units, dtype=np.uint32)
        missing = np.zeros(n_units, dtype=np.uint32)
        ambig = np.zeros(n_units, dtype=np.uint32)
        n_hits = np.zeros(n_units, dtype=np.uint32)
        hits = np.zeros(int(offsets[-1]) if want_hits else 0, dtype=np.uint32)
        self._chk(self.L.bns_classify_batch(self.h, bases.ctypes.data, _p(offsets, u64p), n_reads, int(paired),
                                            _p(taxon, u32p), _p(missing, u32p), _p(ambig, u32p), _p(n_hits, u32p),
                                            _p(hits, u32p) if want_hits else None), "bns_classify_batch")
        out = {"taxon": taxon, "missing": missing, "ambig": ambig, "n_hits": n_hits}
        if want_hits:
            inc = 2 if paired else 1
            out["hits"] = [hits[int(offsets[u * inc]):int(offsets[u * inc]) + int(n_hits[u])].copy()
                           for u in range(n_units)]
        return out

    def classify_packed(self, words, bad_word, bad_mask, offsets, paired=False, want_hits=False):
        """classify() over a batch already packed by pack_reads(): same results, 40 instead of 150 bytes per 150-bp read uploaded"""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        bad_word = np.ascontiguousarray(bad_word, dtype=np.uint64); bad_mask = np.ascontiguousarray(bad_mask, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = offsets.size - 1
        n_units = n_reads // (2 if paired else 1)
        taxon = np.zeros(n_units, dtype=np.uint32); missing = np.zeros(n_units, dtype=np.uint32)
        ambig = np.zeros(n_units, dtype=np.uint32); n_hits = np.zeros(n_units, dtype=np.uint32)
        hits = np.zeros(int(offsets[-1]) if want_hits else 0, dtype=np.uint32)
        self._chk(self.L.bns_classify_batch_packed(self.h, _p(words, u64p), _p(bad_word, u64p) if bad_word.size else None,
                                                   _p(bad_mask, u32p) if bad_mask.size else None, bad_word.size, _p(offsets, u64p), n_reads,
                                                   int(paired), _p(taxon, u32p), _p(missing, u32p), _p(ambig, u32p), _p(n_hits, u32p),
                                                   _p(hits, u32p) if want_hits else None), "bns_classify_batch_packed")
        out = {"taxon": taxon, "missing": missing, "ambig": ambig, "n_hits": n_hits}
        if want_hits:
            inc = 2 if paired else 1
            out["hits"] = [hits[int(offsets[u * inc]):int(offsets[u * inc]) + int(n_hits[u])].copy() for u in range(n_units)]
        return out

    def classify_packed_device(self, d_words, d_nmask, d_offsets, n_reads, total_bases, max_read_len, paired, d_taxon,
                               d_missing=None, d_ambig=None, d_n_hits=None, d_hits=None, stream=None):
        self._chk(self.L.bns_classify_batch_packed_device(self.h, d_words, d_nmask, d_offsets, n_reads, total_bases, max_read_len,
                                                          int(paired), d_taxon, d_missing, d_ambig, d_n_hits, d_hits, stream),
                  "bns_classify_batch_packed_device")

    def classify_runs(self, bases, offsets, paired=False):
        """classify() with the hit stream run-length encoded on the device: out["runs"][u] = (taxids, lengths)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = offsets.size - 1
        n_units = n_reads // (2 if paired else 1)
        taxon = np.zeros(n_units, dtype=np.uint32)
        missing = np.zeros(n_units, dtype=np.uint32)
        ambig = np.zeros(n_units, dtype=np.uint32)
        n_hits = np.zeros(n_units, dtype=np.uint32)
        run_start = np.zeros(n_units, dtype=np.uint64)
        n_runs = np.zeros(n_units, dtype=np.uint32)
        rt = u32p(); rl = u32p(); tot = C.c_uint64()
        self._chk(self.L.bns_classify_batch_runs(self.h, bases.ctypes.data, _p(offsets, u64p), n_reads, int(paired),
                                                 _p(taxon, u32p), _p(missing, u32p), _p(ambig, u32p), _p(n_hits, u32p),
                                                 _p(run_start, u64p), _p(n_runs, u32p), C.byref(rt), C.byref(rl),
                                                 C.cast(C.byref(tot), u64p)), "bns_classify_batch_runs")
        n = int(tot.value)
        tax = np.ctypeslib.as_array(rt, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        ln = np.ctypeslib.as_array(rl, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        runs = [(tax[int(s):int(s) + int(c)], ln[int(s):int(s) + int(c)]) for s, c in zip(run_start, n_runs)]
        return {"taxon": taxon, "missing": missing, "ambig": ambig, "n_hits": n_hits, "runs": runs, "n_runs_total": n}

    def classify_text(self, texts, final=True, limit=None, trim_readno=False, parse_only=False, want_runs=False, want_words=False,
                      cap_records=None, names_cap=None, device_ptrs=None, runs_cap=None, defer=False, between=None):
        """bns_classify_text: FASTA / FASTQ TEXT (bytes, or a pair of bytes: mates) parsed, packed and classified on the device.
        device_ptrs = [(ptr, n_bytes), ...]: the text is already in HBM.  -> dict with n_records, consumed, status, why, per-unit
        results, per-record seq_len / rec_pos / names, runs as classify_runs() gives them, the packed words when asked for.
        defer: the call in two halves (BNS_TEXT_DEFER, then bns_text_finish; `between()` runs in between) -- "first_half" in the result is
        what the first half reported."""
        if isinstance(texts, (bytes, bytearray, memoryview, np.ndarray)):
            texts = [texts]
        bufs = [np.frombuffer(bytes(t), dtype=np.uint8) if not isinstance(t, np.ndarray) else np.ascontiguousarray(t, dtype=np.uint8) for t in texts]
        ns = len(device_ptrs) if device_ptrs is not None else len(bufs)
        if device_ptrs is not None:
            ptrs = (vp * ns)(*[vp(int(p)) for p, _ in device_ptrs])
            sizes = np.array([n for _, n in device_ptrs], dtype=np.uint64)
        else:
            ptrs = (vp * ns)(*[vp(b.ctypes.data if b.size else 0) for b in bufs])
            sizes = np.array([b.size for b in bufs], dtype=np.uint64)
        cap = int(cap_records) if cap_records is not None else int(sizes.sum()) // 2 + 16
        ncap = int(names_cap) if names_cap is not None else int(sizes.sum()) + 16
        nu_cap = cap
        a = {"taxon": np.zeros(nu_cap, np.uint32), "missing": np.zeros(nu_cap, np.uint32), "ambig": np.zeros(nu_cap, np.uint32),
             "n_hits": np.zeros(nu_cap, np.uint32), "seq_len": np.zeros(cap, np.uint32), "rec_pos": np.zeros(cap, np.uint64),
             "name_off": np.zeros(cap + 1, np.uint32), "names": np.zeros(ncap, np.uint8)}
        o = _lib.TextOut()
        for k in ("taxon", "missing", "ambig", "n_hits", "seq_len", "rec_pos", "name_off", "names"):
            setattr(o, k, a[k].ctypes.data)
        o.names_cap = ncap
        if want_runs:
            a["run_start"] = np.zeros(nu_cap, np.uint64); a["n_runs"] = np.zeros(nu_cap, np.uint32)
            o.run_start = a["run_start"].ctypes.data; o.n_runs = a["n_runs"].ctypes.data
        if want_runs and runs_cap is not None:                 # the caller's own run arrays instead of the context's
            a["run_tax"] = np.zeros(max(1, int(runs_cap)), np.uint32); a["run_len"] = np.zeros(max(1, int(runs_cap)), np.uint32)
            o.run_tax = a["run_tax"].ctypes.data; o.run_len = a["run_len"].ctypes.data; o.runs_cap = int(runs_cap)
        if want_words:
            nw = int(sizes.sum()) // 32 + cap + 2
            a["words"] = np.zeros(nw, np.uint64); a["nmask"] = np.zeros(nw, np.uint32)
            o.words = a["words"].ctypes.data; o.nmask = a["nmask"].ctypes.data
        info = _lib.TextInfo()
        flags = (_lib.TEXT_FINAL if final else 0) | (_lib.TEXT_TRIM_READNO if trim_readno else 0) | (_lib.TEXT_PARSE_ONLY if parse_only else 0) | \
                (_lib.TEXT_DEVICE if device_ptrs is not None else 0) | (_lib.TEXT_DEFER if defer else 0)
        lim = int(limit) if limit is not None else 0xFFFFFFFFFFFFFFFF
        self._chk(self.L.bns_classify_text(self.h, ptrs, _p(sizes, u64p), ns, lim, flags, cap, C.byref(o), C.byref(info)), "bns_classify_text")
        first_half = None
        if defer:
            first_half = {"n_records": int(info.n_records), "consumed": [int(info.consumed[i]) for i in range(ns)], "status": int(info.status),
                          "total_bases": int(info.total_bases), "names_bytes": int(info.names_bytes)}
            if between is not None:
                between()
            info = _lib.TextInfo()
            self._chk(self.L.bns_text_finish(self.h, C.byref(info)), "bns_text_finish")
        n = int(info.n_records)
        nu = n // ns
        names = a["names"].tobytes()
        res = {"n_records": n, "consumed": [int(info.consumed[i]) for i in range(ns)], "status": int(info.status), "why": int(info.why),
               "total_bases": int(info.total_bases), "n_slices": int(info.n_slices), "n_launches": int(info.n_launches), "ms_parse": float(info.ms_parse), "ms_classify": float(info.ms_classify),
               "seq_len": a["seq_len"][:n].copy(), "rec_pos": a["rec_pos"][:n].copy(),
               "names": [names[int(a["name_off"][r]):int(a["name_off"][r + 1])] for r in range(n)]}
        if not parse_only:
            for k in ("taxon", "missing", "ambig", "n_hits"):
                res[k] = a[k][:nu].copy()
            if want_runs:
                nt = int(info.n_runs_total)
                if runs_cap is not None:
                    tax, ln = a["run_tax"][:nt].copy(), a["run_len"][:nt].copy()
                else:
                    tax = np.ctypeslib.as_array(info.run_tax, shape=(nt,)).copy() if nt else np.zeros(0, np.uint32)
                    ln = np.ctypeslib.as_array(info.run_len, shape=(nt,)).copy() if nt else np.zeros(0, np.uint32)
                res["runs"] = [(tax[int(s):int(s) + int(c)], ln[int(s):int(s) + int(c)]) for s, c in zip(a["run_start"][:nu], a["n_runs"][:nu])]
        if want_words:
            nw = int(self.L.bns_packed_words(int(info.total_bases), n))
            res["words"] = a["words"][:nw].copy(); res["nmask"] = a["nmask"][:nw].copy()
        if first_half is not None:
            res["first_half"] = first_half
        return res

    def encode(self, bases, offsets):
        """Encoder::for_each over a batch (encoder.h:415-442): list of uint64 arrays, one per read."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = offsets.size - 1
        kmers = np.zeros(max(1, int(offsets[-1])), dtype=np.uint64)
        n_k = np.zeros(n_reads, dtype=np.uint32)
        self._chk(self.L.bns_encode_batch(self.h, bases.ctypes.data, _p(offsets, u64p), n_reads, _p(kmers, u64p),
                                          _p(n_k, u32p)), "bns_encode_batch")
        return [kmers[int(offsets[r]):int(offsets[r]) + int(n_k[r])].copy() for r in range(n_reads)]

    def rolling_hash(self, bases, offsets, k, canon=False, tables=None, w=0):
        """RollingHasher<u64>::for_each_hash over a batch (encoder.h:644-865): list of uint64 arrays.  w > k: with a window
        (minimizers of the hash stream by lex_score; the canonical path queues both strands' hashes).
        tables = (fwd[256], rc[256]) character tables, None = the default-seed tables (parity unpinned)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        per = 2 if (w > k and canon) else 1
        out = np.zeros(max(1, per * int(offsets[-1])), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        tf = tr = None
        if tables is not None:
            tf = np.ascontiguousarray(tables[0], dtype=np.uint64); tr = np.ascontiguousarray(tables[1], dtype=np.uint64)
            assert tf.size == 256 and tr.size == 256
        self._chk(self.L.bns_rolling_hash_windowed_batch(self.h, bases.ctypes.data, _p(offsets, u64p), n, k, int(canon), int(w),
                                                         _p(tf, u64p) if tf is not None else None, _p(tr, u64p) if tr is not None else None,
                                                         _p(out, u64p), _p(cnt, u32p)), "bns_rolling_hash_windowed_batch")
        return [out[per * int(offsets[r]):per * int(offsets[r]) + int(cnt[r])].copy() for r in range(n)]

    def rolling_hash128(self, bases, offsets, k, canon=False, tables=None, w=0):
        """RollingHasher<__uint128_t>::for_each_hash (w > k: with a window of w - k + 1 values): list of (n, 2) uint64 arrays
        [lo, hi], one per sequence.  tables = (fwd, rc), each 256 x (lo, hi) u64; None = the default-seed tables."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        per = 2 if (w > k and canon) else 1
        out = np.zeros(max(1, 2 * per * int(offsets[-1])), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        tf = tr = None
        if tables is not None:
            tf = np.ascontiguousarray(tables[0], dtype=np.uint64).reshape(-1); tr = np.ascontiguousarray(tables[1], dtype=np.uint64).reshape(-1)
            assert tf.size == 512 and tr.size == 512
        self._chk(self.L.bns_rolling_hash128_windowed_batch(self.h, bases.ctypes.data, _p(offsets, u64p), n, k, int(canon), int(w),
                                                            _p(tf, u64p) if tf is not None else None, _p(tr, u64p) if tr is not None else None,
                                                            _p(out, u64p), _p(cnt, u32p)), "bns_rolling_hash128_windowed_batch")
        return [out[2 * per * int(offsets[r]):2 * (per * int(offsets[r]) + int(cnt[r]))].reshape(-1, 2).copy() for r in range(n)]

    def for_each_hash(self, bases, offsets, k=0, canon=-1, table=None):
        """Encoder::for_each_hash over a batch (encoder.h:355-394, ntHash): list of uint64 arrays, one per sequence.
        k = 0: the encoder's k; canon = -1: the encoder's flag; table = 256 seeds in make_nthash_lut's geometry (None: published)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(max(1, int(offsets[-1])), dtype=np.uint64)
        cnt = np.zeros(n, dtype=np.uint32)
        t = None
        if table is not None:
            t = np.ascontiguousarray(table, dtype=np.uint64)
            assert t.size == 256
        self._chk(self.L.bns_for_each_hash_batch(self.h, bases.ctypes.data, _p(offsets, u64p), n, int(k), int(canon),
                                                 _p(t, u64p) if t is not None else None, _p(out, u64p), _p(cnt, u32p)),
                  "bns_for_each_hash_batch")
        return [out[int(offsets[r]):int(offsets[r]) + int(cnt[r])].copy() for r in range(n)]

    def probe(self, kmers):
        """kh_get over a batch (khash64.h:250-263): (vals, found)."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
        vals = np.zeros(kmers.size, dtype=np.uint32)
        found = np.zeros(kmers.size, dtype=np.uint8)
        self._chk(self.L.bns_probe(self.h, _p(kmers, u64p), kmers.size, _p(vals, u32p), _p(found, u8p)), "bns_probe")
        return vals, found

    def resolve(self, keys, counts, starts):
        """resolve_tree (util.h:831-869) over a batch of insertion-ordered counters."""
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        n_units = starts.size - 1
        taxon = np.zeros(n_units, dtype=np.uint32)
        self._chk(self.L.bns_resolve_batch(self.h, _p(keys, u32p), _p(counts, u16p), _p(starts, u64p), n_units,
                                           _p(taxon, u32p)), "bns_resolve_batch")
        return taxon

    # ---- device-pointer entry points (ints are raw device addresses, e.g. torch.Tensor.data_ptr())
    def classify_device(self, d_bases, d_offsets, n_reads, total_bases, max_read_len, paired, d_taxon,
                        d_missing=None, d_ambig=None, d_n_hits=None, d_hits=None, stream=None):
        self._chk(self.L.bns_classify_batch_device(self.h, d_bases, d_offsets, n_reads, total_bases, max_read_len,
                                                   int(paired), d_taxon, d_missing, d_ambig, d_n_hits, d_hits, stream),
                  "bns_classify_batch_device")

    def probe_device(self, d_kmers, n, d_vals, d_found=None, stream=None):
        self._chk(self.L.bns_probe_device(self.h, d_kmers, n, d_vals, d_found, stream), "bns_probe_device")

    def build_table_device(self, d_bases, d_offsets, n_genomes, total_bases, d_taxid, n_buckets, d_flags, d_keys,
                           d_vals, stream=None):
        hdr = np.zeros(4, dtype=np.uint64)
        self._chk(self.L.bns_build_table_device(self.h, d_bases, d_offsets, n_genomes, total_bases, d_taxid, n_buckets,
                                                d_flags, d_keys, d_vals, _p(hdr, u64p), stream), "bns_build_table_device")
        return hdr

    def set_timing(self, on=True):
        self._chk(self.L.bns_set_timing(self.h, int(on)), "bns_set_timing")

    def last_kernel_ms(self):
        return float(self.L.bns_last_kernel_ms(self.h))

    def timing_summary(self):
        """(sum_ms, count) of the dominant-kernel launches since the last summary."""
        s = C.c_double(); n = C.c_int()
        self._chk(self.L.bns_timing_summary(self.h, C.byref(s), C.byref(n)), "bns_timing_summary")
        return s.value, n.value

    # ---- raw device buffers for hosts without a HIP binding
    def dev_alloc(self, nbytes):
        p = vp()
        self._chk(self.L.bns_dev_alloc(self.h, nbytes, C.byref(p)), "bns_dev_alloc")
        return p.value

    def dev_free(self, ptr):
        self._chk(self.L.bns_dev_free(self.h, ptr), "bns_dev_free")

    def dev_upload(self, dst, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.bns_dev_upload(self.h, dst, arr.ctypes.data, arr.nbytes), "bns_dev_upload")

    def dev_download(self, src, arr):
        self._chk(self.L.bns_dev_download(self.h, arr.ctypes.data, src, arr.nbytes), "bns_dev_download")

    def dev_copy_from(self, dst, other, src, nbytes):
        """bns_dev_copy_peer: nbytes from `src` in the device memory of context `other` to `dst` in this context's (one device or two)"""
        self._chk(self.L.bns_dev_copy_peer(self.h, dst, other.h, src, nbytes), "bns_dev_copy_peer")

    def sync(self):
        self._chk(self.L.bns_dev_sync(self.h), "bns_dev_sync")
