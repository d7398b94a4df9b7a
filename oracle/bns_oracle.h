/*
 * bns_oracle.h -- CPU restatement of the Bonsai classify hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it, and only as the checker.  The product
 * path (bonsai_amd/csrc, include/bonsai_amd.h) never links or loads this file.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * dnbaker/bonsai checkout).  Parity pinning status is documented in oracle/README.md
 * and DESIGN.md.  Pinned against the reference's OWN code compiled in the build container
 * (oracle/_ref: whole headers where they compile, line-range extracts of util.h / kmerutil.h /
 * classifier.h / feature_min.h otherwise) and frozen as .npz files under tests/golden: khash probe/insert,
 * linear counter/set, reverse_complement / canonical_representation, lca, resolve_tree,
 * build_parent_map, update_lca_map, the classify_seq body (hit lambda, ambig arithmetic), the
 * bns.db table bytes, the Kraken / FASTQ formatters.  Encoder k-mer streams: the reference
 * test's phiX vector (test/encoding.cpp:122).  PARITY UNPINNED (un-vendored third-party
 * arithmetic, stated where it occurs): score::Lex (FRev64), RollingHasher's character tables,
 * NTC64 of Encoder::for_each_hash, the last ulp of the string-overload entropy sum.
 */
#ifndef BNS_ORACLE_H
#define BNS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BO_TAX_ABSENT 0xFFFFFFFFu   /* parent[] sentinel: taxid is not a key of the parent map */
#define BO_KH_END(h) ((h)->n_buckets)

/* khash_t(c): u64 key -> u32 value; field order/types as khash64.h:213-219 + util.h:160 */
typedef struct {
    uint64_t n_buckets, size, n_occupied, upper_bound;
    uint32_t *flags;
    uint64_t *keys;
    uint32_t *vals;
} bo_khc_t;

/* flattened parent map (util.h:766-785 builds khash_t(p); we index by taxid) */
typedef struct {
    uint32_t n;        /* ids in [0,n) */
    uint32_t *parent;  /* BO_TAX_ABSENT when the id is not a key */
} bo_tax_t;

/* linear::counter<tax_t,u16> (linear/linear.h:181-264): insertion-ordered, u16 counts */
typedef struct {
    uint32_t *keys;
    uint16_t *vals;
    uint32_t n, m;
} bo_counter_t;

typedef struct {
    uint32_t taxon;     /* resolve_tree result */
    uint32_t missing;   /* k-mers not in db */
    uint32_t ambig;     /* classifier.h:232,235 (u32 arithmetic, wraps) */
    uint32_t n_hits;    /* taxa.size() */
} bo_result_t;

/* ---- scalar primitives ---- */
uint64_t bo_wang64(uint64_t key);                       /* khash64.h:202-211 */
uint64_t bo_revcomp(uint64_t kmer, unsigned k);         /* kmerutil.h:83-90 */
uint64_t bo_canonical(uint64_t kmer, unsigned k);       /* kmerutil.h:137-140 */
int      bo_dna4(unsigned char c);                      /* alphabet.h:128 (+30-59): A0 C1 G2 T3 else -1 */
uint32_t bo_comb_size(const uint16_t *gaps, unsigned k);/* spacer.h:14-19 */
int      bo_parse_spacing(const char *s, unsigned k, uint16_t *gaps_out); /* spacer.h:29-47 */

/* ---- Encoder (w == k, i.e. the classify configuration) ---- */
typedef void (*bo_kmer_cb)(uint64_t kmer, void *ud);
/* Encoder::for_each(func,str,len) encoder.h:415-442, INCLUDING defect F7 (spaced => emits nothing). */
void bo_for_each(const char *s, uint64_t l, unsigned k, const uint16_t *gaps, int canon,
                 bo_kmer_cb cb, void *ud);
/* Encoder::for_each_uncanon_spaced encoder.h:233-239 with a 1-wide window (the intended spaced path). */
void bo_for_each_uncanon_spaced(const char *s, uint64_t l, unsigned k, const uint16_t *gaps,
                                bo_kmer_cb cb, void *ud);
/* convenience: collect into an array; returns number emitted (may exceed cap; only cap stored) */
uint64_t bo_encode(const char *s, uint64_t l, unsigned k, const uint16_t *gaps, int canon,
                   int spaced_intended, uint64_t *out, uint64_t cap);

/* ---- windowed minimizers (db construction path) ---- */
#define BO_SCORE_LEX 0           /* score::Lex = FRev64 (encoder.h:47,59-60); arithmetic lives in the un-vendored sketch
                                    library: restated from its public definition, PARITY UNPINNED */
#define BO_SCORE_ENTROPY_PATH 1  /* score::Entropy through the path/kseq overloads: kmer() feeds k-1 symbols to the k-wide
                                    CircusEnt, value() is NOT_FULL=-1, score = (u64)(i64)(double(kmer)/(-1+1e-4)) */
uint64_t bo_score(uint64_t kmer, int score_kind);
/* Encoder::for_each_canon_windowed (encoder.h:211-217): one value per window of w-c+1 consecutive k-mers, the
 * canonical k-mer with the smallest (score, kmer) (qmap.h:16-29,79-87), duplicates kept.  A k-mer containing a non-ACGT
 * base is ENCODE_OVERFLOW, which canonical_representation() turns into 0 before scoring (encoder.h:624-625).
 * Returns the number emitted (= max(0, l-w+1) for w >= comb). */
uint64_t bo_encode_windowed(const char *s, uint64_t l, unsigned k, const uint16_t *gaps, unsigned w, int score_kind,
                            uint64_t *out, uint64_t cap);
/* Encoder::for_each_uncanon_unspaced_windowed (encoder.h:273-306): contiguous seed, no canonicalisation, w > k.  Windows
 * run over the stream of emitted forward k-mers (not over positions), a short sequence flushes one partial window, and
 * with k >= 31 a run of 32 T restarts the k-mer like an invalid base does (see the .c file). */
uint64_t bo_encode_uncanon_windowed(const char *s, uint64_t l, unsigned k, unsigned w, int score_kind,
                                    uint64_t *out, uint64_t cap);
/* Encoder<score::Entropy>::for_each(func, str, len), contiguous seed, w > k (encoder.h:307-353): the string overload's REAL
 * entropy score, (u64)(double(fwd_kmer) / (sum (n/k) ln(n/k) + .001)); selection on forward k-mers, the emitted value
 * canonicalised when canon.  PARITY UNPINNED to the last ulp of the sum (the reference adds in hash-map iteration order; here
 * A, C, G, T) and in the three non-A homopolymer k-mers (score > 2^64: the conversion is instruction-set dependent). */
double bo_kmer_entropy(uint64_t kmer, unsigned k);
uint64_t bo_encode_windowed_entropy_str(const char *s, uint64_t l, unsigned k, unsigned w, int canon, uint64_t *out, uint64_t cap);
/* db construction with a windowed Spacer: update_lca_map over the windowed stream (canon: Encoder's canonicalize_) */
void bo_lca_map_add_windowed(bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps, unsigned w, int score_kind,
                             int canon, const char *seq, uint64_t len, uint32_t taxid);

/* ---- khash_t(c) ---- */
bo_khc_t *bo_khc_init(void);                            /* khash64.h:231-233 */
void      bo_khc_destroy(bo_khc_t *h);
uint64_t  bo_khc_get(const bo_khc_t *h, uint64_t key);  /* khash64.h:250-263 */
int       bo_khc_resize(bo_khc_t *h, uint64_t new_n_buckets); /* khash64.h:264-326 */
uint64_t  bo_khc_put(bo_khc_t *h, uint64_t key, int *ret);    /* khash64.h:327-368 */
/* batch lookup: out_val[i] = value, out_found[i] = 1 on hit */
void      bo_khc_get_batch(const bo_khc_t *h, const uint64_t *keys, uint64_t n,
                           uint32_t *out_val, uint8_t *out_found);
/* wrap caller-owned arrays (no copy, no ownership) */
void      bo_khc_wrap(bo_khc_t *h, uint64_t n_buckets, uint64_t size, uint64_t n_occupied,
                      uint64_t upper_bound, uint32_t *flags, uint64_t *keys, uint32_t *vals);

/* ---- taxonomy ---- */
int      bo_tax_from_nodes_dmp(const char *path, bo_tax_t *out);   /* util.h:766-785 */
int      bo_tax_from_pairs(const uint32_t *child, const uint32_t *parent, uint32_t n_pairs, bo_tax_t *out);
void     bo_tax_free(bo_tax_t *t);
uint32_t bo_lca(const bo_tax_t *t, uint32_t a, uint32_t b);        /* util.h:634-663 */

/* ---- counter + resolve ---- */
void     bo_counter_init(bo_counter_t *c);
void     bo_counter_free(bo_counter_t *c);
void     bo_counter_clear(bo_counter_t *c);
uint32_t bo_counter_add(bo_counter_t *c, uint32_t key);            /* linear.h:229-240 */
uint16_t bo_counter_count(const bo_counter_t *c, uint32_t key);    /* linear.h:241-244 */
uint32_t bo_resolve_tree(const bo_counter_t *c, const bo_tax_t *t);/* util.h:831-869 */
/* known-answer helper: resolve from (key,count) arrays given in insertion order */
uint32_t bo_resolve_pairs(const uint32_t *keys, const uint16_t *counts, uint32_t n, const bo_tax_t *t);

/* ---- classify_seq (classifier.h:212-251) ---- */
/* s2==NULL => single-end.  spaced_intended: 0 = reference behaviour (F7), 1 = intended.
 * hits (optional) receives the ordered hit taxa (the reference's `taxa` vector), up to hits_cap. */
void bo_classify_seq(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps,
                     int canon, int spaced_intended,
                     const char *s1, uint32_t l1, const char *s2, uint32_t l2,
                     bo_result_t *res, uint32_t *hits, uint32_t hits_cap);

/* batch over concatenated ASCII reads; units are reads (paired==0) or adjacent mate pairs.
 * offsets has n_reads+1 entries.  res has n_units entries.  nthreads<=1 => serial. */
void bo_classify_batch(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps,
                       int canon, int spaced_intended, int paired,
                       const char *bases, const uint64_t *offsets, uint64_t n_reads,
                       bo_result_t *res, int nthreads);

/* calibration only (tools/cpu_calibrate.py): the batch loop cut after phase 0 (encode) / 1 (+ kh_get) / 2 (= bo_classify_batch) */
void bo_classify_batch_phase(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps, int canon,
                             const char *bases, const uint64_t *offsets, uint64_t n_reads, bo_result_t *res, int nthreads,
                             int phase, uint64_t *sink);
/* Kraken-style line (classifier.h:112-129 + 45-70): returns bytes written (no NUL counted). */
size_t bo_kraken_line(char *buf, size_t cap, const char *name, uint32_t taxon, int l_seq,
                      uint32_t missing, uint32_t ambig, const uint32_t *hits, uint32_t n_hits);

/* ---- db construction (feature_min.h:205-228 update_lca_map) ---- */
/* add every k-mer of one genome sequence under `taxid`: first sighting stores taxid,
 * later sightings with a different value store lca(tax, taxid, old). */
void bo_lca_map_add(bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps, int canon,
                    const char *seq, uint64_t len, uint32_t taxid);

/* ---- bns.db IO (database.h:33-56,81-102; util.h:280-364) ---- */
/* spacing_width: 1 (as the reader expects) or 2 (as the gz writer emits).  gaps are "extra gap" values. */
int bo_db_write(const char *path, uint32_t k, uint32_t w, const uint16_t *gaps, int spacing_width,
                bo_khc_t *db);   /* zeroes empty/deleted slots like util.h:282-284 */
int bo_db_read(const char *path, uint32_t *k, uint32_t *w, uint16_t *gaps /*>=63 entries*/, bo_khc_t *db);

/* get_taxid's name extraction (util.h:898-929): from the first header line of a genome file (without '>'): the field
 * between the last two '|' when the line has pipes, else the first whitespace-delimited token.  Writes into out. */
void bo_genome_name(const char *header_line, char *out, size_t cap);

/* ---- RollingHasher<u64> without a window (encoder.h:644-865).  PARITY UNPINNED: the character tables
 * are an input; bo_rolling_tables fills them from a restated generator (see bns_oracle.c). */
void bo_rolling_tables(uint64_t seed1, uint64_t seed2, uint64_t *fwd /*256*/, uint64_t *rc /*256*/);
uint64_t bo_rolling_hash(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *fwd, const uint64_t *rc,
                         uint64_t *out, uint64_t cap);
/* with a window of w bases (w <= k: none): minimizers by (FRev64(v), v) over w-k+1 consecutive hashes; the canonical path
 * queues both strands' hashes as separate entries; a queue that never fills flushes its minimum (encoder.h:706-736,771-795) */
uint64_t bo_rolling_hash_windowed(const char *s, uint64_t l, unsigned k, int canon, unsigned w, const uint64_t *fwd,
                                  const uint64_t *rc, uint64_t *out, uint64_t cap);

/* RollingHasher<__uint128_t> without a window: values and table entries as (lo, hi) u64 pairs; tables have 256 entries = 512 u64. */
void bo_rolling_tables128(uint64_t seed1, uint64_t seed2, uint64_t *fwd_lohi, uint64_t *rc_lohi);
uint64_t bo_rolling_hash128(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *fwd_lohi, const uint64_t *rc_lohi,
                            uint64_t *out_lohi, uint64_t cap);
/* the same hasher with a window (w > k); its queue's score function is restated (parity unpinned: sketch's CEHasher) */
uint64_t bo_rolling_hash128_windowed(const char *s, uint64_t l, unsigned k, int canon, unsigned w, const uint64_t *fwd_lohi,
                                     const uint64_t *rc_lohi, uint64_t *out_lohi, uint64_t cap);

/* ---- Encoder::for_each_hash (encoder.h:355-394): ntHash (NTC64) stream of a contiguous, unwindowed seed.  PARITY UNPINNED:
 * NTC64 lives in the un-vendored bcgsc/ntHash submodule (.gitmodules, version unpinned); restated from the published
 * definition (ntHash 1.0.x rol/ror form), the 256-entry seed table is an input.  bo_nthash_tables fills it in make_nthash_lut's
 * in-tree geometry (encoder.h:93-103): seed of a base at its letter (both cases), seed of its complement at letter & 7. */
void bo_nthash_tables(uint64_t seed_a, uint64_t seed_c, uint64_t seed_g, uint64_t seed_t, uint64_t *table256);
uint64_t bo_for_each_hash(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *table256, uint64_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
