/*
 * ref_harness.cpp -- thin C exports around the REFERENCE's own code, compiled from where it lies
 * under $(REF) (default /root/reference) into oracle/_ref/libbns_ref.so.  Test infrastructure only;
 * container-only: oracle/_ref is git-ignored, and the generated source ranges under oracle/_ref/gen are gpurun-ignored -- reference SOURCE does not
 * travel; the compiled libbns_ref.so does (git-ignored, not gpurun-ignored), for bench.py's cpu_baseline leg on the GPU box.
 *
 * (1) Reference headers that compile whole, with no stand-ins, are included in place:
 *   include/bonsai/khash64.h       (kh_init/put/get/resize/del, __ac_Wang64_hash, flag macros)
 *   linear/linear.h                (linear::counter, linear::set)
 *   include/bonsai/logutil.h       (LOG_WARNING / LOG_DEBUG)
 *   kspp/ks.h                      (ks::string, the formatters' output buffer; system <zlib.h>)
 *   include/bonsai/kseq_declare.h  (+ klib/kseq.h: kseq_read, bseq1_t, bseq_read; system <zlib.h>)
 *   include/bonsai/rhtraits.h      (+ alphabet.h, std headers only: the DNA4 symbol table the Encoder indexes -- including its
 *                                   "U:T" alias as make_lut really resolves it --, InputType, rhmask, mul)
 * (2) The hot path's functions that live in headers which do NOT compile here (util.h, kmerutil.h,
 *   classifier.h, feature_min.h pull in the un-vendored sketch / ntHash / libpopcnt submodules) are cut
 *   out BY LINE RANGE at build time (oracle/ref_extract.py -> oracle/_ref/gen/*.inc, guarded by the text
 *   each range must hold) and compiled here unmodified:
 *     kmerutil.h:83-90,137-140   reverse_complement, canonical_representation
 *     util.h:279-294             khash_write_impl (fd overload: the bns.db table section)
 *     util.h:540-551             RUNTIME_ERROR
 *     util.h:634-663             lca
 *     util.h:766-785             build_parent_map
 *     util.h:831-869             resolve_tree
 *     feature_min.h:205-228      update_lca_map
 *     classifier.h:10-129        append_taxa_run(s), append_counts, append_fastq/kraken_classification
 *   The typedefs and khash instantiations those functions see are repeated from util.h:44-47,66,126-132,
 *   158-162 (this file's only restated lines; util.h itself is unbuildable here).
 * No reference source is copied into the repository; the generated .inc files exist only under oracle/_ref.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>              /* rhtraits.h uses std::pow without including <cmath> itself */
#include <cinttypes>
#include <string>
#include <vector>
#include <fstream>
#include <stdexcept>
#include <memory>
#include <unistd.h>
#include <fcntl.h>
#include "include/bonsai/khash64.h"
#include "linear/linear.h"
#include "include/bonsai/logutil.h"
#include "kspp/ks.h"
#include "include/bonsai/kseq_declare.h"
#include "include/bonsai/rhtraits.h"

#ifndef likely
#  define likely(x) __builtin_expect((x),1)        /* util.h:44 */
#endif
#ifndef unlikely
#  define unlikely(x) __builtin_expect((x),0)      /* util.h:47 */
#endif
namespace bns {
using u8 = std::uint8_t; using u16 = std::uint16_t; using u32 = std::uint32_t; using u64 = std::uint64_t;   /* util.h:126-130 */
using tax_t = u32;                                 /* util.h:132 */
using std::size_t;
KHASH_SET_INIT_INT64(all)        /* util.h:158 */
KHASH_MAP_INIT_INT64(c, tax_t)   /* util.h:159 */
KHASH_MAP_INIT_INT(p, tax_t)     /* util.h:161 */
#include "gen/runtime_error.inc"
#include "gen/revcomp.inc"
#include "gen/canonical.inc"
#include "gen/khash_write.inc"
#include "gen/lca.inc"
#include "gen/build_parent_map.inc"
#include "gen/resolve_tree.inc"
#include "gen/update_lca_map.inc"
#include "gen/formatters.inc"
} // namespace bns
using namespace bns;

extern "C" {

uint64_t ref_wang64(uint64_t k) { return __ac_Wang64_hash(k); }

void *ref_khc_new(void) { return kh_init(c); }
void ref_khc_free(void *h) { kh_destroy(c, (khash_t(c) *)h); }
/* sequential insert exactly as feature_min.h:212-216 does (kh_get, then kh_put + assign on miss;
 * on hit the value is overwritten with `val` here -- callers pass unique keys when layout matters) */
void ref_khc_insert(void *hv, const uint64_t *keys, const uint32_t *vals, uint64_t n)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    int khr;
    for (uint64_t i = 0; i < n; ++i) {
        khint_t k2 = kh_get(c, h, keys[i]);
        if (k2 == kh_end(h)) k2 = kh_put(c, h, keys[i], &khr);
        kh_val(h, k2) = vals[i];
    }
}
int ref_khc_resize(void *hv, uint64_t nb) { return kh_resize(c, (khash_t(c) *)hv, nb); }
void ref_khc_del_key(void *hv, uint64_t key)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    khint_t k2 = kh_get(c, h, key);
    if (k2 != kh_end(h)) kh_del(c, h, k2);
}
void ref_khc_info(void *hv, uint64_t *out4, uint32_t **flags, uint64_t **keys, uint32_t **vals)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    out4[0] = h->n_buckets; out4[1] = h->size; out4[2] = h->n_occupied; out4[3] = h->upper_bound;
    *flags = h->flags; *keys = (uint64_t *)h->keys; *vals = h->vals;
}
void ref_khc_get_batch(void *hv, const uint64_t *keys, uint64_t n, uint32_t *out_val, uint8_t *out_found)
{
    khash_t(c) *h = (khash_t(c) *)hv;
    for (uint64_t i = 0; i < n; ++i) {
        khint_t k2 = kh_get(c, h, keys[i]);
        out_found[i] = (k2 != kh_end(h));
        out_val[i] = out_found[i] ? kh_val(h, k2) : 0;
    }
}
/* raw slot index returned by kh_get (kh_end == n_buckets on miss) */
uint64_t ref_khc_get(void *hv, uint64_t key) { return kh_get(c, (khash_t(c) *)hv, key); }

/* linear::counter<tax_t,u16> as classifier.h:10 declares it */
uint32_t ref_counter(const uint32_t *adds, uint32_t n, uint32_t *keys_out, uint16_t *vals_out)
{
    linear::counter<tax_t, uint16_t> ct;
    for (uint32_t i = 0; i < n; ++i) ct.add(adds[i]);
    for (uint32_t i = 0; i < ct.size(); ++i) { keys_out[i] = ct.keys()[i]; vals_out[i] = ct.vals()[i]; }
    return ct.size();
}
uint16_t ref_counter_count(const uint32_t *adds, uint32_t n, uint32_t key)
{
    linear::counter<tax_t, uint16_t> ct;
    for (uint32_t i = 0; i < n; ++i) ct.add(adds[i]);
    return ct.count(key);
}
/* linear::set insertion order (used by resolve_tree's max_taxa and lca's `nodes`) */
uint32_t ref_linear_set(const uint32_t *ins, uint32_t n, uint32_t *out)
{
    linear::set<tax_t> s;
    for (uint32_t i = 0; i < n; ++i) s.insert(ins[i]);
    uint32_t m = 0;
    for (auto v : s) out[m++] = v;
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * the line-range-extracted functions (see the header comment)
 * ---------------------------------------------------------------------------------------------- */
uint64_t ref_revcomp(uint64_t kmer, unsigned k) { return reverse_complement(kmer, (uint8_t)k); }
uint64_t ref_canonical(uint64_t kmer, unsigned k) { return canonical_representation(kmer, (uint8_t)k); }

/* khash_t(p) built with the reference's kh_put from (child, parent) pairs, in order; later pairs overwrite */
void *ref_khp_from_pairs(const uint32_t *child, const uint32_t *parent, uint32_t n)
{
    khash_t(p) *h = kh_init(p);
    int khr;
    for (uint32_t i = 0; i < n; ++i) { khint_t ki = kh_put(p, h, child[i], &khr); kh_val(h, ki) = parent[i]; }
    return h;
}
void ref_khp_free(void *h) { kh_destroy(p, (khash_t(p) *)h); }
/* build_parent_map (util.h:766-785) on a nodes.dmp; NULL when it throws */
void *ref_build_parent_map(const char *path)
{
    try { return build_parent_map(path); } catch (const std::exception &) { return nullptr; }
}
uint32_t ref_khp_size(void *hv) { return kh_size((khash_t(p) *)hv); }
/* dump as (child, parent) pairs in slot order; returns the number written */
uint32_t ref_khp_pairs(void *hv, uint32_t *child, uint32_t *parent, uint32_t cap)
{
    khash_t(p) *h = (khash_t(p) *)hv;
    uint32_t m = 0;
    for (khint_t ki = kh_begin(h); ki != kh_end(h); ++ki)
        if (kh_exist(h, ki) && m < cap) { child[m] = kh_key(h, ki); parent[m] = kh_val(h, ki); ++m; }
    return m;
}
uint32_t ref_lca(void *hv, uint32_t a, uint32_t b) { return lca((khash_t(p) *)hv, a, b); }
void ref_lca_batch(void *hv, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = lca((khash_t(p) *)hv, a[i], b[i]);
}
/* resolve_tree (util.h:831-869) over counters given as the ordered add() stream of one read each:
 * adds[offs[i] .. offs[i+1]) are the taxids the hit lambda (classifier.h:225-229) would add, in order. */
void ref_resolve_adds_batch(void *hv, const uint32_t *adds, const uint64_t *offs, uint64_t n, uint32_t *out)
{
    for (uint64_t i = 0; i < n; ++i) {
        linear::counter<tax_t, u16> ct;
        for (uint64_t j = offs[i]; j < offs[i + 1]; ++j) ct.add(adds[j]);
        out[i] = resolve_tree(ct, (khash_t(p) *)hv);
    }
}
/* the same from (key, count) pairs in insertion order: each key is add()ed count times, keys interleaved
 * so that the insertion order is the given one (first sighting of key j before first sighting of key j+1) */
uint32_t ref_resolve_pairs(void *hv, const uint32_t *keys, const uint32_t *counts, uint32_t n)
{
    linear::counter<tax_t, u16> ct;
    for (uint32_t j = 0; j < n; ++j) ct.add(keys[j]);
    for (uint32_t j = 0; j < n; ++j) for (uint32_t c = 1; c < counts[j]; ++c) ct.add(keys[j]);
    return resolve_tree(ct, (khash_t(p) *)hv);
}

/* classify_seq's body (classifier.h:225-238) from a k-mer stream: the hit lambda, the ambig arithmetic and
 * resolve_tree, as the reference composes them.  The k-mers themselves come from the caller (the encoder is
 * pinned separately; Encoder<> is unbuildable here).  kmers2 == NULL: single-end.  out4 = taxon, missing,
 * ambig, n_hits; hits (optional) receives the `taxa` vector. */
void ref_classify_kmers(void *dbv, void *taxv, unsigned comb, const uint64_t *kmers1, uint32_t n1, int l_seq1,
                        const uint64_t *kmers2, uint32_t n2, int l_seq2, uint32_t *out4, uint32_t *hits, uint32_t hits_cap)
{
    const khash_t(c) *db = (const khash_t(c) *)dbv;
    khiter_t ki;
    tax_counter hit_counts;
    u32 missing_count(0);
    tax_t taxon(0);
    std::vector<tax_t> taxa;
    auto fn = [&](u64 kmer) {
        if ((ki = kh_get(c, db, kmer)) == kh_end(db)) ++missing_count;
        else taxa.push_back(kh_val(db, ki)), hit_counts.add(kh_val(db, ki));
    };
    for (uint32_t i = 0; i < n1; ++i) fn(kmers1[i]);
    unsigned ambig_count(l_seq1 - comb + 1 - taxa.size() - missing_count);
    if (kmers2) {
        for (uint32_t i = 0; i < n2; ++i) fn(kmers2[i]);
        ambig_count += l_seq2 - (comb - 1) - taxa.size() - missing_count;
    }
    taxon = resolve_tree(hit_counts, (const khash_t(p) *)taxv);
    out4[0] = taxon; out4[1] = missing_count; out4[2] = ambig_count; out4[3] = (uint32_t)taxa.size();
    if (hits) for (size_t i = 0; i < taxa.size() && i < hits_cap; ++i) hits[i] = taxa[i];
}

/* A khash_t(c) VIEW of caller-owned arrays (no copy; free the handle with ref_khc_view_free, never kh_destroy): lets the
 * calibration below run the reference's kh_get on a table of benchmark size without re-inserting every key. */
void *ref_khc_view(uint64_t n_buckets, uint64_t size, uint64_t n_occupied, uint64_t upper_bound, uint32_t *flags, uint64_t *keys, uint32_t *vals)
{
    khash_t(c) *h = (khash_t(c) *)std::calloc(1, sizeof(khash_t(c)));
    h->n_buckets = n_buckets; h->size = size; h->n_occupied = n_occupied; h->upper_bound = upper_bound;
    h->flags = flags; h->keys = (decltype(h->keys))keys; h->vals = vals;
    return h;
}
void ref_khc_view_free(void *h) { std::free(h); }

/* CPU-baseline calibration (tools/cpu_calibrate.py): classify_seq's body (classifier.h:212-251)
 * over a whole batch INSIDE this library -- no per-read foreign call -- built from the reference's own pieces: DNA4 / rhmask /
 * mul / canonical_representation driving the loop of encoder.h:246-271 (restated as in ref_kmer_stream, here with the hit lambda
 * inlined as `func`, the way Encoder::for_each's template inlines it), the reference's kh_get, linear::counter and resolve_tree.
 * Single-end, contiguous seeds (what `bonsai classify` runs by default).  Threads: the reference fans reads out over a kt_for
 * pool (classifier.h:275); here an OpenMP dynamic loop.
 * phase: 0 = encode only (k-mers summed into *sink), 1 = encode + kh_get (hits counted), 2 = the whole of classify_seq.
 * out4: taxon, missing, ambig, n_hits per read (phase 2). */
void ref_classify_batch(void *dbv, void *taxv, unsigned k_, int canon, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                        uint32_t *out4, int nthreads, int phase, uint64_t *sink)
{
    const khash_t(c) *db = (const khash_t(c) *)dbv;
    const khash_t(p) *tax = (const khash_t(p) *)taxv;
    const int8_t *lutptr = (const int8_t *)bns::alph::DNA4.data();
    const uint64_t mask(bns::rhmask<uint64_t>(bns::DNA, (int)k_));
    const uint64_t ENCODE_OVERFLOW = uint64_t(-1);
    const size_t mul = bns::mul(bns::DNA);
    uint64_t total = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+:total)
    for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
        const char *s_ = bases + offsets[r];
        const uint64_t l_ = offsets[r + 1] - offsets[r];
        khiter_t ki;
        tax_counter hit_counts;
        u32 missing_count(0);
        std::vector<tax_t> taxa;
        uint64_t acc = 0;
        auto func = [&](u64 kmer) {
            if (phase == 0) { acc += kmer; return; }
            if ((ki = kh_get(c, db, kmer)) == kh_end(db)) ++missing_count;
            else if (phase == 1) acc += kh_val(db, ki);
            else taxa.push_back(kh_val(db, ki)), hit_counts.add(kh_val(db, ki));
        };
        uint64_t min, pos_ = 0;
        unsigned filled;
        loop_start:
        min = filled = 0;
        while (pos_ < l_) {
            while (filled < k_ && pos_ < l_) {
                const char c_at_pos = s_[pos_];
                const int8_t nv = lutptr[c_at_pos];
                ++pos_;
                if (nv == int8_t(-1)) { min = ENCODE_OVERFLOW; goto loop_start; }
                min = (min * mul) | nv;
                ++filled;
            }
            if (filled == k_) {
                min &= mask;
                func(canon ? canonical_representation(min, (uint8_t)k_) : min);
                --filled;
            }
        }
        total += acc + missing_count;
        if (phase == 2) {
            unsigned ambig_count((unsigned)l_ - k_ + 1 - taxa.size() - missing_count);
            const tax_t taxon = resolve_tree(hit_counts, tax);
            out4[4 * r] = taxon; out4[4 * r + 1] = missing_count; out4[4 * r + 2] = ambig_count; out4[4 * r + 3] = (uint32_t)taxa.size();
        }
    }
    if (sink) *sink = total;
}

/* update_lca_map (feature_min.h:205-228): `keys` become one khash_t(all) set (kh_put in order), folded into db */
void ref_update_lca_map(void *dbv, void *taxv, const uint64_t *keys, uint64_t n, uint32_t taxid)
{
    khash_t(all) *set = kh_init(all);
    int khr;
    for (uint64_t i = 0; i < n; ++i) kh_put(all, set, keys[i], &khr);
    update_lca_map((khash_t(c) *)dbv, set, (const khash_t(p) *)taxv, taxid);
    kh_destroy(all, set);
}

/* khash_write_impl(map, fd) (util.h:279-294): the table section of a bns.db, appended to `path` */
int64_t ref_khc_write(void *dbv, const char *path, int append)
{
    int fd = ::open(path, O_WRONLY | O_CREAT | (append ? O_APPEND : O_TRUNC), 0644);
    if (fd < 0) return -1;
    int64_t r = (int64_t)khash_write_impl((const khash_t(c) *)dbv, fd);
    ::close(fd);
    return r;
}

/* the reference's formatters (classifier.h:30-129).  mate2 fields may be NULL for single-end.  Returns the
 * length written to out (no NUL counted), or -1 when cap is too small.
 * The ks::string is created with `cap` bytes up front: append_fastq_classification keeps raw pointers into the
 * buffer across appends (cms/cme, classifier.h:79,91,100), so a buffer that has to grow in between makes its
 * paired branch copy from freed memory with a garbage length (seen under ASan).  With room for the whole record
 * nothing reallocates and the output is the defined one. */
static int64_t fmt_out(ks::string &bks, char *out, size_t cap)
{
    if (bks.size() + 1 > cap) return -1;
    memcpy(out, bks.data(), bks.size()); out[bks.size()] = 0;
    return (int64_t)bks.size();
}
int64_t ref_kraken_line(const char *name, uint32_t taxon, int l_seq, uint32_t missing, uint32_t ambig,
                        const uint32_t *hits, uint32_t n_hits, char *out, size_t cap)
{
    tax_counter hc;
    std::vector<tax_t> taxa(hits, hits + n_hits);
    bseq1_t bs; memset(&bs, 0, sizeof(bs));
    bs.name = (char *)name; bs.l_seq = l_seq;
    ks::string bks((uint64_t)cap);
    append_kraken_classification(hc, taxa, taxon, ambig, missing, &bs, bks);
    return fmt_out(bks, out, cap);
}
int64_t ref_fastq_record(const char *name1, const char *seq1, const char *qual1, int l1,
                         const char *name2, const char *seq2, const char *qual2, int l2,
                         uint32_t taxon, uint32_t missing, uint32_t ambig, const uint32_t *hits, uint32_t n_hits,
                         int verbose, int is_paired, char *out, size_t cap)
{
    tax_counter hc;
    std::vector<tax_t> taxa(hits, hits + n_hits);
    bseq1_t bs[2]; memset(bs, 0, sizeof(bs));
    bs[0].name = (char *)name1; bs[0].seq = (char *)seq1; bs[0].qual = (char *)qual1; bs[0].l_seq = l1;
    bs[1].name = (char *)name2; bs[1].seq = (char *)seq2; bs[1].qual = (char *)qual2; bs[1].l_seq = l2;
    ks::string bks((uint64_t)cap);
    append_fastq_classification(hc, taxa, taxon, ambig, missing, bs, bks, verbose, is_paired);
    return fmt_out(bks, out, cap);
}

/* kseq_read / bseq_read (klib/kseq.h:177-225, kseq_declare.h:106-146) over one or two files: every record of every
 * chunk, flattened.  For record i: name, comment, seq, qual ('' when absent) appended to `blob` NUL-separated;
 * returns the number of records, or -1 when blob_cap is too small.  chunk_size as process_dataset passes it. */
int64_t ref_bseq_read_all(const char *path1, const char *path2, int chunk_size, char *blob, size_t blob_cap,
                          int32_t *l_seq, int32_t *chunk_of, int64_t rec_cap)
{
    gzFile f1 = gzopen(path1, "rb"), f2 = path2 ? gzopen(path2, "rb") : nullptr;
    if (!f1 || (path2 && !f2)) return -2;
    kseq_t *k1 = kseq_init(f1), *k2 = f2 ? kseq_init(f2) : nullptr;
    int64_t nrec = 0; size_t pos = 0; int n = 0, chunk = 0;
    bseq1_t *seqs;
    bool overflow = false;
    while ((seqs = bseq_read(chunk_size, &n, k1, k2)) != nullptr) {
        for (int i = 0; i < n; ++i) {
            const char *fields[4] = {seqs[i].name, seqs[i].comment, seqs[i].seq, seqs[i].qual};
            for (int f = 0; f < 4; ++f) {
                const char *s = fields[f] ? fields[f] : "";
                size_t l = strlen(s) + 1;
                if (pos + l > blob_cap || nrec >= rec_cap) { overflow = true; break; }
                memcpy(blob + pos, s, l); pos += l;
            }
            if (overflow) break;
            l_seq[nrec] = seqs[i].l_seq; chunk_of[nrec] = chunk; ++nrec;
        }
        for (int i = 0; i < n; ++i) free(seqs[i].name);      /* one block per record (kseq_declare.h:54-73); sam unset */
        free(seqs);
        if (overflow) break;
        ++chunk;
    }
    kseq_destroy(k1); gzclose(f1);
    if (k2) { kseq_destroy(k2); gzclose(f2); }
    return overflow ? -1 : nrec;
}


/* ---- Encoder k-mer stream: reference LUT / mask / multiplier + RESTATED loop -------------------------
 * Encoder<> itself cannot be compiled here: its core loop declares a schism::Schismatic (encoder.h:241-242, un-vendored sketch
 * library; used only on the non-DNA branch).  What the DNA path computes is: the symbol table DNA4 (alphabet.h:128, the
 * Encoder's lutptr, encoder.h:127), rhmask<u64>(DNA, k) (rhtraits.h:51-68), rhmul() = mul(DNA) (rhtraits.h:69-81) -- all three
 * REFERENCE CODE, compiled above where it lies -- driven by the loop of encoder.h:246-271, which is restated below statement by
 * statement minus the Schismatic declaration and the non-DNA else-branch; the canonical wrapper (encoder.h:218-232) applies
 * the reference's own canonical_representation (kmerutil.h:137-140, cut by line range above).  So alphabet, bit order, mask and
 * canonical form come from reference code; only the control flow is a restatement.  Bytes >= 128 index the table with a
 * negative char in the reference (undefined): callers keep to 7-bit input. */
int ref_dna4_lut(int8_t *out256) { std::memcpy(out256, bns::alph::DNA4.data(), 256); return (int)bns::alph::DNA4.size(); }
uint64_t ref_rhmask_dna(int k) { return bns::rhmask<uint64_t>(bns::DNA, k); }
uint64_t ref_rhmul_dna(void) { return bns::mul(bns::DNA); }
uint64_t ref_kmer_stream(const char *s_, uint64_t l_, unsigned k_, int canon, uint64_t *out)
{
    const int8_t *lutptr = (const int8_t *)bns::alph::DNA4.data();            /* encoder.h:127 */
    const uint64_t mask(bns::rhmask<uint64_t>(bns::DNA, (int)k_));            /* :240 */
    const uint64_t ENCODE_OVERFLOW = uint64_t(-1);                              /* :119 */
    uint64_t min, n = 0, pos_ = 0;
    unsigned filled;
    const size_t mul = bns::mul(bns::DNA);                                      /* :246 rhmul() */
    loop_start:
    min = filled = 0;
    while (pos_ < l_) {
        while (filled < k_ && pos_ < l_) {
            const char c_at_pos = s_[pos_];
            const int8_t nv = lutptr[c_at_pos];
            ++pos_;
            if (nv == int8_t(-1)) { min = ENCODE_OVERFLOW; goto loop_start; }
            min = (min * mul) | nv;
            ++filled;
        }
        if (filled == k_) {
            min &= mask;                                                        /* :260, rht == DNA */
            out[n++] = canon ? canonical_representation(min, (uint8_t)k_) : min;   /* func(min) / :220-222 */
            --filled;
        }
    }
    return n;
}

} /* extern "C" */
