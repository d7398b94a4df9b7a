/*
 * bonsai_amd.h -- C ABI of the MI355X-native Bonsai classify hot path.
 *
 * Drop-in boundary.  The reference (dnbaker/bonsai) has no FFI layer: the path sits
 * behind C++ templates.  Each entry point below names the reference interface it replaces
 * (file:line into the reference checkout).  Plain pointers and sizes only, int error codes, no
 * exceptions cross this boundary, one context per device, calls on one context are serialised by the
 * caller.  All multi-byte data is little-endian host order.
 *
 * Host-pointer entry points copy in/out and synchronise before returning.  The `_device` entry points
 * take HIP device pointers plus a hipStream_t (passed as void*) and only enqueue work.
 */
#ifndef BONSAI_AMD_H
#define BONSAI_AMD_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bns_ctx bns_ctx;

enum {
    BNS_OK = 0,
    BNS_ERR_ARG = -1,            /* bad argument (k out of [1,32], NULL pointer, ...) */
    BNS_ERR_HIP = -2,            /* HIP runtime failure; see bns_last_error() */
    BNS_ERR_NOMEM = -3,          /* device or host allocation failed */
    BNS_ERR_STATE = -4,          /* table / taxonomy / spacer not loaded yet */
    BNS_ERR_TAX_CYCLE = -5,      /* parent[] contains a cycle (the reference would loop forever) */
    BNS_ERR_TAX_RANGE = -6,      /* taxid >= 2^28: flat parent[] cap */
    BNS_ERR_TABLE = -7,          /* khash arrays inconsistent (n_buckets not a power of two, unreachable key) */
    BNS_ERR_NO_DEVICE = -8       /* no usable gfx950 device */
};

/* table layouts in HBM (bns_load_table*) */
enum {
    BNS_LAYOUT_KHASH  = 0,  /* probe the on-disk SoA arrays as they are: Wang64 + triangular probing, khash64.h:250-263 */
    BNS_LAYOUT_BUCKET = 1,  /* re-hash on device into 64-byte buckets of 4 x {key,val,occ}; same key->value map */
    BNS_LAYOUT_MINBUCKET = 2 /* (default) 128-byte buckets {u64 keys[10], u32 vals[10], u32 count|occupancy, u32 S}.  The home
                                bucket of a key comes from its minimizer -- the smallest hash among the canonical m-mers inside
                                the k-mer, m = k-15, k-11 or k-8 as the db allows (never below 16 / 19 / 19: m = k for small k;
                                spaced seeds: m-mers of the mask's longest run; see bns_set_minimizer_span) -- so neighbouring k-mers
                                of a read share a 128-byte line; inside a bucket the key sits at the slot a per-bucket
                                perfect-hash multiplier S assigns it; a full bucket spills to the next one (at most 4), then to
                                a small plain-hashed overflow table.  Same key->value map.  Needs bns_set_encoder() before the
                                table is loaded (the layout depends on k). */
};

#define BNS_TAX_ABSENT 0xFFFFFFFFu   /* parent[] entry of an id that is not a key of the parent map */

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* Replaces: ClassifierGeneric ctor (classifier.h:155-166) + ForPool (util.h:109-118): the context owns
 * the device, its stream and all device allocations. */
int  bns_create(int device, bns_ctx **out);
int  bns_device_count(void);            /* visible HIP devices (0 when there is none) */
/* PCI address of a device ("0000:05:00.0", what /sys/bus/pci/devices/ lists it under): lets a host bind its threads to the CPUs
 * next to the GPU (.../local_cpulist) before it creates a context.  cap >= 16. */
int  bns_device_pci_bus_id(int device, char *out, int cap);
void bns_destroy(bns_ctx *ctx);
const char *bns_strerror(int code);
const char *bns_last_error(const bns_ctx *ctx);     /* detail of the last failure on this context */
int  bns_version(void);

/* ---- encoder configuration -------------------------------------------------------------------- */
/* Replaces: Spacer(k, w=k, spaces) (spacer.h:58-71) + Encoder(const Spacer&, bool canonicalize)
 * (encoder.h:153) as classify builds them (bin/bonsai.cpp:152-153).  gaps = k-1 "extra gap" values as
 * stored in bns.db (database.h:46-48), NULL = contiguous.  A spaced seed is never canonicalised
 * (encoder.h:148-150).  spaced_intended: 0 reproduces the reference's string for_each (a spaced seed
 * emits nothing); 1 = for_each_uncanon_spaced semantics (encoder.h:233-239). */
int bns_set_encoder(bns_ctx *ctx, uint32_t k, const uint16_t *gaps, int canonicalize, int spaced_intended);

/* Replaces: the window argument of Spacer(k, w, spaces) (spacer.h:58-71) and the Encoder's ScoreType template
 * argument (encoder.h:113): windowed minimizer selection as `bonsai build -w <w> [-e]` uses it
 * (bin/bonsai.cpp:226-261 -> feature_min.h:67-82 -> Encoder::for_each_canon_windowed, encoder.h:211-217; with
 * canonicalize = 0 for_each_uncanon_unspaced_windowed, encoder.h:273-306, whose windows run over the emitted k-mers;
 * a spaced seed goes through for_each_uncanon_spaced, encoder.h:233-239).
 * Honoured by bns_encode_batch* and bns_build_table_device; classify always looks up every k-mer (w = k,
 * bin/bonsai.cpp:152).  w <= comb size = unwindowed.  At most 1024 k-mers per window.
 *   BNS_SCORE_LEX           score::Lex = FRev64 (encoder.h:47) -- restated from the un-vendored sketch library,
 *                           parity unpinned
 *   BNS_SCORE_ENTROPY_PATH  score::Entropy as the path overloads compute it: (u64)(i64)(double(kmer)/(-1+1e-4))
 *                           
 *   BNS_SCORE_ENTROPY_STRING score::Entropy as the STRING overload computes it for a contiguous seed
 *                           (for_each_[un]canon_unspaced_windowed_entropy_, encoder.h:307-353): (u64)(double(fwd_kmer) /
 *                           (sum over the k-mer's bases of (n/k) ln(n/k) + .001)); like the -C windowed stream its windows
 *                           run over the emitted k-mers, selection is on forward k-mers and the emitted value is
 *                           canonicalised when canonicalize = 1.  Parity unpinned to the last ulp of the sum (the
 *                           reference adds in hash-map order; here A, C, G, T) */
enum { BNS_SCORE_LEX = 0, BNS_SCORE_ENTROPY_PATH = 1, BNS_SCORE_ENTROPY_STRING = 2 };
int bns_set_window(bns_ctx *ctx, uint32_t w, int score);

/* ---- database --------------------------------------------------------------------------------- */
/* Replaces: Database<khash_t(c)>(path).db_ (database.h:33-56 -> util.h:334-364 khash_load_impl): the
 * three khash arrays exactly as they sit in bns.db.  flags has max(1, n_buckets>>4) words.  For the re-hashed layouts the arrays
 * are uploaded whole when they fit the HBM next to the table they are turned into, and streamed from these host buffers in
 * chunks of 2^27 slots when they do not (an 8e9-key db: 210 GB of arrays, 200+ GB of table); BNS_LAYOUT_KHASH keeps them resident. */
int bns_load_table(bns_ctx *ctx, uint64_t n_buckets, const uint32_t *flags, const uint64_t *keys,
                   const uint32_t *vals, int layout);
/* Same, arrays already resident in HBM (e.g. after an RCCL broadcast).  With BNS_LAYOUT_KHASH the
 * context BORROWS the arrays (caller keeps them alive); with BNS_LAYOUT_BUCKET they are only read
 * during the call's enqueued kernels. */
int bns_load_table_device(bns_ctx *ctx, uint64_t n_buckets, const uint32_t *d_flags, const uint64_t *d_keys,
                          const uint32_t *d_vals, int layout, void *stream);
/* Size of the re-hashed table.  Default (nothing set), BNS_LAYOUT_MINBUCKET: sized from the number of PRESENT keys -- one
 * 128-byte bucket per 0.83 keys (buckets of 10 filled to 1/12: the knee of the load sweep) or what fits in three
 * quarters of the free HBM, whichever is smaller; any bucket count will do (the bucket index is a multiply-high, not a mask), so
 * a RefSeq-scale db takes exactly the memory there is.  BNS_LAYOUT_BUCKET: 2x the khash bucket count in 16-byte slots.
 *   bns_set_bucket_slots_log2  the exact log2 of the size in 16-byte slots (MINBUCKET: 2^(log2-3) buckets), 0 = automatic
 *   bns_set_table_buckets      MINBUCKET: the exact number of buckets a key can call home (>= keys / 9.7), 0 = automatic
 * No reference counterpart: the key -> value map is the same whatever the size. */
int bns_set_bucket_slots_log2(bns_ctx *ctx, uint32_t log2_slots);
int bns_set_table_buckets(bns_ctx *ctx, uint64_t n_home_buckets);
/* Multi-GPU load (the seam is process_dataset, classifier.h:296-337): the same table in n_ctx contexts, one per
 * device.  The khash arrays cross PCIe ONCE (into ctxs[0]'s device) and reach the other devices by an RCCL broadcast over xGMI
 * (librccl is opened on first use, only when two different devices take part); every device then builds its own clustered
 * layout, all of them at the size and with the minimizer window ctxs[0] chose.  Contexts that share a device (a one-GPU box exercising this path) are fed
 * by device-to-device copies instead -- RCCL admits a device once per communicator.  n_ctx == 1 is bns_load_table.  On failure no
 * context keeps a table and bns_last_error(ctxs[0]) holds the failing context's message.  A db whose
 * arrays do not fit the HBM next to its table is not replicated array by array: every context streams the host buffers into its own
 * table (as bns_load_table does), the first alone -- its table size is everyone's -- the others side by side. */
int bns_load_table_multi(bns_ctx **ctxs, int n_ctx, uint64_t n_buckets, const uint32_t *flags, const uint64_t *keys,
                         const uint32_t *vals, int layout);

/* What the last bns_load_table_multi broadcast ran on: librccl's version code (ncclGetVersion: e.g. 22606; 0 before the library was
 * opened) and the number of ranks of its communicator (0: no broadcast yet -- one context, or a db streamed per device).  A host prints
 * it so that a multi-GPU record says which collective library replicated the table over how many devices. */
int bns_rccl_info(int *version, int *n_ranks);
/* number of present keys / device bytes of the active table */
int bns_table_info(const bns_ctx *ctx, uint64_t *n_keys, uint64_t *device_bytes, int *layout);
/* stats4 = {present keys, keys in the MINBUCKET overflow table, main table bytes, overflow table bytes} */
int bns_table_stats(const bns_ctx *ctx, uint64_t *stats4);

/* The MINBUCKET table's minimizer window for contiguous seeds, span = k - m.  0 (default): chosen when the table is loaded --
 * the widest of 15, 11, 8 with which fewer than 1 key in 100 misses its home bucket (a db of window minimizers, bonsai build
 * -w 50, takes 15: fewer bucket fetches per read; a db of every k-mer needs 8).  8 / 11 / 15 fix it.  No reference counterpart:
 * the key -> value map is the same whatever the window (tests/test_gpu_ref_golden.py runs all three).  Call before
 * bns_load_table*.  bns_table_minimizer reports m and the number of keys that are not in their home bucket. */
int bns_set_minimizer_span(bns_ctx *ctx, uint32_t span);
int bns_table_minimizer(const bns_ctx *ctx, uint32_t *m, uint64_t *spilled_keys);
/* Width of the minimizer identity the MINBUCKET table orders its m-mers by and derives the bucket from (contiguous seeds):
 * 32 = a 32-bit hash (cheapest per lookup; beyond a few 1e8 minimizer groups distinct groups share hash values, hence buckets,
 * whatever the table size), 52 = hash and m-mer carried through the window minimum as one double (no sharing; what dbs of
 * RefSeq scale need), 0 (default) = chosen when the table is loaded: both forms are tried on a sample of the buckets, and 52 is
 * taken when the 32-bit form leaves more than 2 keys in 100 outside their home bucket and 52 bits at most 60 % of that.  Call
 * before bns_load_table*.  Same key -> value map either way. */
int bns_set_minimizer_identity(bns_ctx *ctx, int bits);
/* How the clustered table is filled: 0 (default) = chosen when the table is loaded, 1 = keys in arrival order, 2 = group by group
 * (whole minimizer groups keep their home bucket, largest first; the header's tag bits say which groups have keys elsewhere:
 * docs/TABLE_LAYOUT.md) -- what the loader takes by itself for a crowded table (more than 1 key in 100 outside its home bucket
 * in its trial pass).  Same key -> value map either way; replicas of a multi-GPU load take the root's choice.  Call before
 * bns_load_table*.  No reference counterpart (khash has one layout: khash64.h:198-263). */
int bns_set_table_fill(bns_ctx *ctx, int mode);
/* geo8 = {buckets a key can call home (MINBUCKET) / buckets (BUCKET, KHASH), minimizer length m, identity bits (32 / 52),
 * keys that are not in their home bucket, the window the table was built with as bns_set_minimizer_span names it (15 / 11 / 8;
 * 0 for a spaced seed), keys in the overflow table, 1 when the table was filled group by group (crowded tables: whole minimizer
 * groups keep their home bucket, largest first), 0}; zeros where the layout has no such thing.  Feeding entries 0, 4, 2 and
 * 6 + 1 to bns_set_table_buckets / bns_set_minimizer_span / bns_set_minimizer_identity / bns_set_table_fill reproduces the table
 * elsewhere. */
int bns_table_geometry(const bns_ctx *ctx, uint64_t *geo8);
/* "" or what the last bns_load_table* had to say about the table it built (e.g. a forced minimizer window whose groups
 * outgrow their buckets: correct results, slower lookups).  Valid until the next load on this context. */
const char *bns_table_warning(const bns_ctx *ctx);

/* Replaces: build_parent_map(nodes.dmp) (util.h:766-785) as a flat array: parent[id] for id in [0,n),
 * BNS_TAX_ABSENT where id is not a key.  parent[1] must already be 0 (util.h:780-781). */
int bns_load_taxonomy(bns_ctx *ctx, const uint32_t *parent, uint32_t n);

/* ---- hot path --------------------------------------------------------------------------------- */
/* Replaces: the kt_forpool fan-out in classify_seqs (classifier.h:275) over classify_seq
 * (classifier.h:212-251), i.e. per read (or mate pair): Encoder::for_each -> kh_get(c) ->
 * linear::counter -> resolve_tree.
 *   bases    concatenated ASCII sequences (bseq1_t::seq, kseq_declare.h:40-44), no terminators needed
 *   offsets  n_reads+1 byte offsets into bases
 *   paired   0: one unit per read; 1: reads 2u,2u+1 are mates (one vote per pair, classifier.h:233-236)
 * Outputs, one entry per unit (n_reads or n_reads/2):
 *   taxon    resolve_tree result (0 = unclassified)
 *   missing  k-mers absent from the table                      (classifier.h:227)
 *   ambig    the reference's u32 ambig_count arithmetic        (classifier.h:232,235)
 *   n_hits   hits (taxa.size()); may be NULL
 *   hits     optional (NULL to skip): ordered hit taxids = the reference's `taxa` vector
 *            (classifier.h:228), unit u's hits start at hits[offsets[first read of u]]; array has
 *            offsets[n_reads] entries. */
int bns_classify_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                       int paired, uint32_t *taxon, uint32_t *missing, uint32_t *ambig,
                       uint32_t *n_hits, uint32_t *hits);
/* Same call with the hit stream returned run-length encoded -- exactly what the Kraken formatter prints
 * (append_taxa_runs / classifier.h:45-61: "taxid:count" per run of equal consecutive hits), computed on the device so that
 * 4 bytes per k-mer do not cross PCIe.  Unit u's runs are run_tax[run_start[u] + j], run_len[run_start[u] + j] for
 * j < n_runs[u]; run_tax / run_len point into buffers owned by the context (valid until the next call on it), holding
 * *n_runs_total entries.  Where a unit's runs sit in them may differ from call to call; their content does not. */
int bns_classify_batch_runs(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads, int paired,
                            uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint64_t *run_start,
                            uint32_t *n_runs, const uint32_t **run_tax, const uint32_t **run_len, uint64_t *n_runs_total);
/* Device-resident variant: every pointer is a device pointer; max_read_len is the caller's upper
 * bound on any read length in the batch (0 = unknown: the call measures it, costing one sync).
 * d_bases must be 4-byte aligned and readable up to the next 4-byte boundary past total_bases (any hipMalloc'd or
 * framework-allocated buffer is). */
int bns_classify_batch_device(bns_ctx *ctx, const char *d_bases, const uint64_t *d_offsets,
                              uint64_t n_reads, uint64_t total_bases, uint32_t max_read_len, int paired,
                              uint32_t *d_taxon, uint32_t *d_missing, uint32_t *d_ambig,
                              uint32_t *d_n_hits, uint32_t *d_hits, void *stream);

/* ---- packed reads (north_star: "reads batched and packed 2-bit into coalesced HBM loads"; SURVEY 8b's proposed
 * classify_batch(ctx, packed_bases, valid_mask, read_offsets, ...)) ------------------------------------------------------------------
 * The same batch as above in the form the kernel works on: 2-bit codes A0 C1 G2 T3 (alphabet.h:128), 32 bases per uint64_t
 * word, FIRST base in the TOP two bits (the reference's own k-mer bit order, encoder.h:255: a k-mer is a funnel shift of two
 * adjacent words).  offsets[] are still base offsets; read r's words start at word (offsets[r] >> 5) + r (monotone, never
 * overlapping, no prefix sum needed) and the whole batch takes bns_packed_words(total_bases, n_reads) words.  Bases that are not
 * A/C/G/T (N, IUPAC, ...; their code bits are ignored) are flagged per word: bit (31 - i) of the word's 32-bit flag word = base
 * i of that word is invalid.  40 bytes per 150-bp read cross PCIe and HBM instead of 150.
 *   bns_pack_reads            host: ASCII batch -> words + the SPARSE list (word index, flags) of the words that hold an invalid
 *                             base (AVX2/BMI2 when the CPU has them; `threads` workers).  *n_bad = entries written; when it
 *                             exceeds bad_cap nothing is stored there and BNS_ERR_ARG comes back with *n_bad = the room needed.
 *   bns_pack_reads_ptrs       the same from reads that are NOT contiguous in memory (seqs[r], lens[r]: e.g. bseq1_t::seq of a chunk,
 *                             kseq_declare.h:40-44): fills offsets[n_reads + 1] itself -- packing replaces the gather copy a host
 *                             would otherwise make.
 *   bns_classify_batch_packed / _packed_runs   host buffers, as bns_classify_batch / _runs; the flags travel as the sparse list
 *                             (n_bad may be 0) and are scattered into a dense array on the device.
 *   bns_classify_batch_packed_device          everything resident; d_nmask = one flag word per packed word (dense), or NULL when
 *                             no read of the batch holds an invalid base.
 * Replaces the same reference code as bns_classify_batch (classifier.h:212-251); results are identical for the same reads. */
uint64_t bns_packed_words(uint64_t total_bases, uint64_t n_reads);
int bns_pack_reads(const char *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t *words, uint64_t *bad_word,
                   uint32_t *bad_mask, uint64_t bad_cap, uint64_t *n_bad, int threads);
int bns_pack_reads_ptrs(const char *const *seqs, const uint32_t *lens, uint64_t n_reads, uint64_t *offsets, uint64_t *words,
                        uint64_t *bad_word, uint32_t *bad_mask, uint64_t bad_cap, uint64_t *n_bad, int threads);
int bns_classify_batch_packed(bns_ctx *ctx, const uint64_t *words, const uint64_t *bad_word, const uint32_t *bad_mask, uint64_t n_bad,
                              const uint64_t *offsets, uint64_t n_reads, int paired,
                              uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint32_t *hits);
int bns_classify_batch_packed_runs(bns_ctx *ctx, const uint64_t *words, const uint64_t *bad_word, const uint32_t *bad_mask, uint64_t n_bad,
                                   const uint64_t *offsets, uint64_t n_reads, int paired,
                                   uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint64_t *run_start,
                                   uint32_t *n_runs, const uint32_t **run_tax, const uint32_t **run_len, uint64_t *n_runs_total);
int bns_classify_batch_packed_device(bns_ctx *ctx, const uint64_t *d_words, const uint32_t *d_nmask, const uint64_t *d_offsets,
                                     uint64_t n_reads, uint64_t total_bases, uint32_t max_read_len, int paired, uint32_t *d_taxon,
                                     uint32_t *d_missing, uint32_t *d_ambig, uint32_t *d_n_hits, uint32_t *d_hits, void *stream);

/* Replaces: Encoder<score::Lex,u64>::for_each(func, str, len) (encoder.h:415-442) over a batch; also
 * what python/bns.cpp from_str/seqlist return (python/bns.cpp:87-129).  kmers has offsets[n_reads]
 * entries; read r's k-mers start at kmers[offsets[r]], n_kmers[r] of them, in sequence order. */
int bns_encode_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                     uint64_t *kmers, uint32_t *n_kmers);
int bns_encode_batch_device(bns_ctx *ctx, const char *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                            uint64_t total_bases, uint64_t *d_kmers, uint32_t *d_n_kmers, void *stream);

/* Replaces: RollingHasher<uint64_t>(k, canon)::for_each_hash(func, s, l) without a window (encoder.h:644-865: canonical
 * path :692-760, other :762-796; CyclicHash rollinghash/cyclichash.h:23-154), as python/bns.cpp:42-77 uses it.  One 64-bit
 * value per k-window of every run of A/C/G/T the reference's loop visits (an invalid character skips k + 1 characters and
 * restarts; the canonical path stops when fewer than 2k characters remain after it), in order; sequence r's values start at
 * hashes[offsets[r]], n_hashes[r] of them.  The 256-entry character tables are an INPUT: the reference draws them from
 * wy::WyRand, a third-party generator absent from its checkout (parity unpinned); NULL, NULL = tables from
 * bns_rolling_tables(1337, 137, ...), the constructor's default seeds through a restated generator. */
int bns_rolling_hash_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                           const uint64_t *fwd_table, const uint64_t *rc_table, uint64_t *hashes, uint32_t *n_hashes);
/* Replaces: RollingHasher<uint64_t>(k, canon, DNA, w) -- the same hasher with a window (encoder.h:664-671; windowed branches
 * :706-736 and :771-795): every hash goes through a QueueMap of w-k+1 entries scored by lex_score (FRev64) and the
 * minimum by (score, value) comes out once the queue is full; the canonical path queues BOTH strands' hashes (forward, then
 * reverse) as separate entries; the queue survives a restart at an invalid character; a sequence that never fills it yields
 * one value.  w <= k: no window (bns_rolling_hash_batch).  Layout: with P = 2 for (w > k and canon) else 1, hashes must hold
 * P * offsets[n_seqs] values and sequence r's values start at hashes[P * offsets[r]], n_hashes[r] of them. */
int bns_rolling_hash_windowed_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                                    uint32_t w, const uint64_t *fwd_table, const uint64_t *rc_table, uint64_t *hashes,
                                    uint32_t *n_hashes);
/* The default tables: 256 + 256 values seeded the way encoder.h:682-683 seeds the forward / reverse hashers. */
int bns_rolling_tables(uint64_t seed1, uint64_t seed2, uint64_t *fwd, uint64_t *rc);
/* Replaces: RollingHasher<__uint128_t, CyclicHash<__uint128_t>>(k, canon)::for_each_hash(func, s, l) (encoder.h:644-865 with a
 * 128-bit word; the instantiation test/encoding.cpp:152 constructs) without a window: the same visiting rules as
 * bns_rolling_hash_batch, 128-bit rotations, myr = k % 128.  Every value and every table entry is a (lo, hi) pair of u64:
 * tables hold 256 entries = 512 u64, hashes_lohi must hold 2 * offsets[n_seqs] u64 and sequence r's values start at
 * hashes_lohi[2 * offsets[r]].  NULL, NULL = the constructor's default seeds through the restated generator, including
 * CharacterHash<u128>'s habit of keeping only the low word of each entry (characterhash.h:82-97).  Parity unpinned as for the
 * 64-bit hasher.
 * bns_rolling_hash128_windowed_batch: the same hasher with a window, RollingHasher<__uint128_t>(k, canon, DNA, w) -- the form the
 * reference's one RollingHasher test constructs (test/encoding.cpp:152-156; window branches encoder.h:706-736, 771-795): a
 * QueueMap of w - k + 1 entries, minimum by (score, value), both strands queued separately on the canonical path, the queue
 * surviving restarts, one flushed value for a stream that never fills it.  The queue's score is lex_score(u128) =
 * sketch::hash::CEHasher in the reference (un-vendored: restated as FRev64 over the folded halves, parity unpinned); the COUNT of
 * values -- what that test pins: len - w + 1 -- does not depend on it.  Layout: P = 2 for (w > k and canon) else 1; hashes_lohi
 * holds 2 * P * offsets[n_seqs] u64, sequence r's values start at hashes_lohi[2 * P * offsets[r]].  Not built: RollingHasherSet. */
int bns_rolling_hash128_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                              const uint64_t *fwd_lohi, const uint64_t *rc_lohi, uint64_t *hashes_lohi, uint32_t *n_hashes);
int bns_rolling_hash128_windowed_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                                       uint32_t w, const uint64_t *fwd_lohi, const uint64_t *rc_lohi, uint64_t *hashes_lohi,
                                       uint32_t *n_hashes);
int bns_rolling_tables128(uint64_t seed1, uint64_t seed2, uint64_t *fwd_lohi, uint64_t *rc_lohi);

/* Replaces: Encoder<>::for_each_hash(func, str, len, k = 0) (encoder.h:355-394) -- the ntHash stream of a contiguous,
 * unwindowed seed, as bin/kmercnt.cpp:78, bin/setsketcher.cpp:98,129-130 select it ("k > 32 implies nthash").  One value per
 * k-window the reference's loop visits: every window of every run of A/C/G/T (either case) at least k long, in order --
 * except that a run whose first window ends exactly at the end of the sequence yields nothing (the NUL test comes first at
 * encoder.h:378-379; so a sequence of exactly k bases is silent) and a NUL byte ends the sequence.  canon: 1 = NTC64's
 * canonical value min(forward, reverse), 0 = the forward value, -1 = the encoder's canonicalize flag (encoder.h:383,392).
 * k = 0 = the encoder's k; k may exceed 32 (the hash is 64-bit whatever k is).  A spaced or windowed encoder is refused as
 * encoder.h:363-364 does.  The arithmetic is NTC64 of bcgsc/ntHash, an un-vendored submodule (.gitmodules, version unpinned):
 * restated from the published definition (rol/ror form of ntHash 1.0.x) with the 256-entry seed table as an INPUT --
 * PARITY UNPINNED.  table256 follows make_nthash_lut's in-tree geometry (encoder.h:93-103: a base's seed at
 * its letter, its complement's seed at letter & 7); NULL = ntHash's published seeds.  Sequence r's values start at
 * hashes[offsets[r]], n_hashes[r] of them. */
int bns_for_each_hash_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                            const uint64_t *table256, uint64_t *hashes, uint32_t *n_hashes);
int bns_nthash_tables(uint64_t seed_a, uint64_t seed_c, uint64_t seed_g, uint64_t seed_t, uint64_t *table256);

/* Replaces: kh_get(c, db, kmer) + kh_val (khash64.h:250-263) over a batch of keys.
 * found[i] = 1 and vals[i] = value on a hit; found[i] = 0, vals[i] = 0 on a miss. */
int bns_probe(bns_ctx *ctx, const uint64_t *kmers, uint64_t n, uint32_t *vals, uint8_t *found);
int bns_probe_device(bns_ctx *ctx, const uint64_t *d_kmers, uint64_t n, uint32_t *d_vals, uint8_t *d_found,
                     void *stream);

/* Replaces: resolve_tree(hit_counts, taxmap) (util.h:831-869) over a batch of counters given in
 * insertion order: unit u owns keys/counts[starts[u] .. starts[u+1]). */
int bns_resolve_batch(bns_ctx *ctx, const uint32_t *keys, const uint16_t *counts, const uint64_t *starts,
                      uint64_t n_units, uint32_t *taxon);

/* ---- database construction on device (feature_min.h:205-228 update_lca_map) ------ */
/* Builds khash_t(c) arrays (valid for kh_get and for bns.db) in HBM from n_genomes sequences:
 * every k-mer the encoder emits for genome g maps to taxid[g]; a k-mer seen under several taxids maps
 * to their lca().  n_buckets must be a power of two with load <= 0.77 (khash64.h:198).  Requires a
 * loaded taxonomy and encoder.  Outputs are device arrays sized like bns_load_table's inputs;
 * header4 (host) receives {n_buckets, n_occupied, size, upper_bound}. */
int bns_build_table_device(bns_ctx *ctx, const char *d_bases, const uint64_t *d_offsets, uint64_t n_genomes,
                           uint64_t total_bases, const uint32_t *d_taxid, uint64_t n_buckets,
                           uint32_t *d_flags, uint64_t *d_keys, uint32_t *d_vals, uint64_t *header4,
                           void *stream);

/* ---- instrumentation -------------------------------------------------------------------------- */
/* HIP-event timing of the dominant kernel (classify_kernel / probe_kernel), recorded on the stream the
 * kernel is launched on.  bns_set_timing(ctx, 1) enables and clears.  bns_last_kernel_ms: most recent
 * launch (< 0 if none).  bns_timing_summary: sum and count over the launches recorded since the last
 * summary (ring of 64), waits for them, then clears. */
int   bns_set_timing(bns_ctx *ctx, int enabled);
float bns_last_kernel_ms(const bns_ctx *ctx);
int   bns_timing_summary(bns_ctx *ctx, double *sum_ms, int *count);

/* raw device memory helpers so non-HIP hosts (ctypes tests) can stage buffers */
/* Page-locked host memory: the host-buffer entry points (bns_classify_batch, ...) copy from / to pageable memory through
 * the runtime's staging buffers at roughly a third of the PCIe rate; buffers obtained here go at full rate. */
int bns_host_alloc(bns_ctx *ctx, size_t bytes, void **out);
/* ... and memory the HOST allocated (and touched), page-locked where it lies: registering 96 MiB of huge-page memory that is already
 * resident takes 0.5 ms where bns_host_alloc takes 15-45 ms (it allocates, faults in and pins under the runtime's lock, with every other
 * thread's calls waiting: the first 0.15 s of a BGZF file were five such slots, tools/micro/pin_bench.hip); copies from it run at
 * 44-50 GB/s against 57.  Nothing in the reference (its buffers are kseq's: klib/kseq.h:62-75). */
int bns_host_register(bns_ctx *ctx, void *p, size_t bytes);
int bns_host_unregister(bns_ctx *ctx, void *p);
int bns_host_free(bns_ctx *ctx, void *p);
int bns_dev_alloc(bns_ctx *ctx, size_t bytes, void **out);
int bns_dev_free(bns_ctx *ctx, void *p);
int bns_dev_upload(bns_ctx *ctx, void *dst, const void *src, size_t bytes);
int bns_dev_download(bns_ctx *ctx, void *dst, const void *src, size_t bytes);
int bns_dev_sync(bns_ctx *ctx);

/* ---- FASTA / FASTQ text parsed on the device (host ingest) -------------------------------------------------------
 * Replaces: kseq_read (klib/kseq.h:177-225) + bseq_read's loop (kseq_declare.h:112-145) + the chunk's classify_seqs fan-out
 * (classifier.h:269-287) for text that is handed over AS TEXT: record boundaries, names, sequence lengths and the 2-bit words the
 * classify kernel reads are made by kernels from the raw bytes of the file -- uploaded from a read(2) buffer, or already in HBM
 * (BNS_TEXT_DEVICE: e.g. left there by bns_inflate_members_device) -- so neither a host parser nor a host packer is on the path.
 *   text[s], text_bytes[s]   n_streams = 1: one file's bytes; 2: a pair of files, record i of stream 0 and record i of stream 1 are
 *                            mates (kseq_declare.h:116-131).  text[s][0] must be the first byte of a record (or of the file).
 *   limit                    records of stream 0 that START at or behind this offset are left alone (a stretch of a file handed to
 *                            this device: the record that straddles its nominal end is this call's, the one that starts behind it
 *                            the next stretch's); >= text_bytes[0]: no limit
 *                            Device text (BNS_TEXT_DEVICE) may start at any address; it must be readable from the 64-byte boundary in front of it to
 *                            the 64-byte boundary behind its end.
 *   flags                    BNS_TEXT_FINAL: the text ends the input (kseq's end-of-file rules close the last record); otherwise the
 *                            last record that STARTS in the text is never taken (a FASTA record ends at the next header) -- it is
 *                            where consumed[] points.  BNS_TEXT_TRIM_READNO: trim_readno (kseq_declare.h:106-110) on every name.
 *   cap_records, out         room in the caller's arrays, in records.  Per unit (record; pair when n_streams = 2): taxon, missing,
 *                            ambig, n_hits as bns_classify_batch; run_start / n_runs as bns_classify_batch_runs (NULL: no runs).
 *                            Per record, mates interleaved: seq_len; rec_pos (offset of the record's header line in its text[s]);
 *                            names[name_off[r] .. name_off[r + 1]) = kseq's name field (first token of the header line).  Any
 *                            pointer but taxon may be NULL.
 *   info                     n_records taken (all streams together); consumed[s] = offset of the first byte of text[s] NOT taken:
 *                            the caller's next call starts there.  status:
 *     BNS_TEXT_OK            everything up to consumed[] was classified
 *     BNS_TEXT_IRREGULAR     the text at consumed[] is not in the form the kernels parse (below): that stretch is the host parser's
 *     BNS_TEXT_NO_RECORD     no complete record starts in what is left (a record longer than the text handed over)
 *     BNS_TEXT_CAP           the caller's arrays are full; call again from consumed[]
 * The REGULAR FORM (everything kseq_read parses the same way line by line; proof in docs/INGEST_NOTES.md): records are a header line
 * ('>' or '@' first) / sequence lines (first byte none of '>', '@', '+'; blank lines skipped) / optionally a '+' line and quality
 * lines -- one or several, whatever they start with -- that are together exactly as long as the sequence / blank lines; line ends
 * '\n' or '\r\n' (one trailing '\r' per appended line is dropped as ks_getuntil2 drops it, klib/kseq.h:135); at most 4096 lines per
 * record.  Wrapped FASTA, wrapped FASTQ sequences and quality (round 6), quality lines that start with '@' or '+', CRLF text (round 6),
 * a missing final newline, empty sequences are all regular; text between records and quality of the wrong length (kseq's error -2)
 * are not (status IRREGULAR: nothing is guessed).
 * Results are those of bns_classify_batch on the records kseq_read yields. */
#define BNS_TEXT_FINAL        1
#define BNS_TEXT_TRIM_READNO  2
#define BNS_TEXT_DEVICE       4
#define BNS_TEXT_PARSE_ONLY   8     /* records only (seq_len, rec_pos, names): nothing is classified; no table needed */
#define BNS_TEXT_DEFER        16    /* parse and pack now -- the call returns when the records are known: info->n_records, consumed[], status,
                                       why, total_bases, names_bytes; the caller's host buffers are his again -- and leave the classify launch, the
                                       hit runs and the copies of EVERY result array (seq_len, names, ... included) to bns_text_finish.  For a caller
                                       that hands blocks of one input to several devices in turn: where block b + 1 starts is known as soon as
                                       block b is parsed, so the next device parses while this one classifies (classifier.h:296-337 reads its
                                       chunks in order too; nothing here is guessed).  Until bns_text_finish the context takes no other
                                       bns_classify_text call (BNS_ERR_STATE; bns_text_prefetch is fine).  Not with out->words / nmask. */
#define BNS_TEXT_OK           0
#define BNS_TEXT_IRREGULAR    1
#define BNS_TEXT_NO_RECORD    2
#define BNS_TEXT_CAP          3
typedef struct bns_text_out {
    uint32_t *taxon, *missing, *ambig, *n_hits;      /* per unit */
    uint64_t *run_start; uint32_t *n_runs;           /* per unit; both or neither */
    uint32_t *seq_len; uint64_t *rec_pos;            /* per record */
    uint32_t *name_off; char *names; uint64_t names_cap;   /* name_off: cap_records + 1 entries */
    uint32_t *run_tax, *run_len; uint64_t runs_cap;  /* optional: where the hit runs go (entries; with run_start / n_runs).  NULL: buffers of the
                                                        context (info->run_tax / run_len, valid until its next call).  Too small: BNS_TEXT_CAP */
    uint64_t *words; uint32_t *nmask;                /* the records' 2-bit image, bns_pack_reads' layout with DENSE flag words
                                                        (bns_packed_words(total_bases, n_records) entries each): calls of one
                                                        piece only (<= 64 MiB of text per stream), BNS_ERR_ARG otherwise */
} bns_text_out;
typedef struct bns_text_info {
    uint64_t n_records, consumed[2], total_bases, names_bytes, n_runs_total;
    const uint32_t *run_tax, *run_len;               /* owned by the context, valid until its next call */
    int32_t status;
    uint32_t why;                                    /* IRREGULAR: BNS_TEXT_WHY_* bits of the first offending stretch */
    uint32_t n_slices, n_launches;                   /* parses (one per upload piece, and more where a window filled up) / classify launches (one per ~2 M records) */
    double ms_parse, ms_classify;                    /* device time of the parse kernels / of classify (HIP events; bns_set_timing) */
} bns_text_info;
#define BNS_TEXT_WHY_CR          1u    /* (round 5: a line ends in '\r'.  Round 6 reads CRLF text on the device: not reported any more) */
#define BNS_TEXT_WHY_LEADING     2u    /* text in front of the first header */
#define BNS_TEXT_WHY_AFTER_QUAL  4u    /* a sequence or '+' line where only a header or a blank line may stand */
#define BNS_TEXT_WHY_QUAL_LEN    8u    /* the quality lines do not end where they are as long as the sequence (kseq's error -2: too long, or the input ends first) */
#define BNS_TEXT_WHY_PLUS_RUN    16u   /* (round 5: more than 16 consecutive lines that start with '+'.  Not reported any more: quality may span lines) */
#define BNS_TEXT_WHY_LONG_RECORD 32u   /* more than 4096 lines in one record */
#define BNS_TEXT_WHY_LINES       64u   /* more lines than one per 8 bytes of text (+ 1024), or more lines that start with '>' / '@' than one per 16 bytes (+ 512) */
int bns_classify_text(bns_ctx *ctx, const char *const *text, const uint64_t *text_bytes, int n_streams, uint64_t limit, int flags,
                      uint64_t cap_records, const bns_text_out *out, bns_text_info *info);
/* Optional: start the upload of the text of the NEXT bns_classify_text call now, so that it travels while the current call
 * computes (a call uploads its own text in pieces and overlaps them with ITS compute; what it cannot overlap is its first piece and
 * its last piece's compute -- a host that calls `prefetch(block b + 1); classify(block b)` keeps the link busy across calls).
 * Returns at once.  The next call's text[s] may be any sub-range of what was prefetched (a block whose first record's offset is
 * only known when the block in front is done); text that was not prefetched is uploaded by the call as before.  The host buffers
 * must stay untouched until the call that consumes them returns; at most one prefetch may be outstanding beside the text of the
 * call in progress (BNS_ERR_STATE otherwise).  n_streams = 0 (text, text_bytes ignored): wait for uploads in flight and forget them --
 * a caller that gives its buffers up without the call that would have consumed them (an input that ended early, an error). */
int bns_text_prefetch(bns_ctx *ctx, const char *const *text, const uint64_t *text_bytes, int n_streams);
/* The second half of a bns_classify_text(..., BNS_TEXT_DEFER, ...) call: classifies what that call parsed, fills the result arrays it was
 * given (they must still be there) and `info` like a call without the flag would have.  status BNS_TEXT_CAP with fewer records than the
 * first half reported: the caller's run arrays were too small -- call bns_classify_text again on the same text with larger ones (the records
 * and consumed[] will be the same).  Replaces nothing in the reference: its classify_seqs (classifier.h:269-287) returns when a chunk is done. */
int bns_text_finish(bns_ctx *ctx, bns_text_info *info);
/* device -> device copy on the context's stream (a caller that keeps text in HBM moves the unconsumed tail in front of the next batch) */
int bns_dev_copy(bns_ctx *ctx, void *dst, const void *src, size_t bytes);
/* ... and between two contexts, src in src_ctx's device memory, dst in dst_ctx's (one device or two: the unconsumed tail of a block that
 * the NEXT device goes on with, when the blocks of one input go to several devices in turn).  Synchronous; waits for no stream of src_ctx.
 * Replaces nothing in the reference (one address space: classifier.h:296-337 keeps its chunk in host memory). */
int bns_dev_copy_peer(bns_ctx *dst_ctx, void *dst, bns_ctx *src_ctx, const void *src, size_t bytes);

/* ---- BGZF members inflated on the device (host ingest) ------------------------------------
 * Replaces, for blocked-gzip input, the reference's one zlib stream (gzFile behind kseq: kseq_declare.h:112-145,
 * klib/kseq.h:177-225 -- ks_getuntil over gzread).  A BGZF file is a sequence of independent gzip members of at most 64 KiB of
 * text; the caller finds them from their 'BC' subfields (no inflating needed) and hands a batch of raw-DEFLATE payloads over:
 * one member per WAVEFRONT, its 64 lanes decoding speculatively at every bit position of a round (csrc/bns_inflate_wave.hpp; 53 GB/s
 * of text at 4 k members per call; BNS_INFLATE_FORM=lane: the round-4 form, one member per lane).  A handle of its own -- stream and staging buffers
 * -- independent of any bns_ctx, so a reader thread inflates while classify calls run; one call at a time per handle.
 *   comp[in_off[i] .. + in_len[i])   member i's DEFLATE payload (between the gzip header and the CRC32/ISIZE trailer)
 *   text[out_off[i] .. + out_len[i]) where its text goes; out_len[i] = the member's ISIZE
 *   crc32[i]                         CRC-32 of the bytes written (compare with the member's trailer)
 *   status[i]                        0, or a BNS_INF_* code (damaged stream, size mismatch): the member's text is then unspecified
 * Host buffers; page-locked ones (bns_inflater_host_alloc) travel at the full link rate.  Returns BNS_OK when the batch ran --
 * per-member failures are reported through status[], not the return value. */
typedef struct bns_inflater bns_inflater;
#define BNS_INF_OK            0
#define BNS_INF_BAD_BLOCK     1
#define BNS_INF_BAD_STORED    2
#define BNS_INF_BAD_LENGTHS   3
#define BNS_INF_BAD_CODE      4
#define BNS_INF_BAD_DISTANCE  5
#define BNS_INF_OUT_OVERFLOW  6
#define BNS_INF_IN_OVERRUN    7
#define BNS_INF_OUT_SHORT     8
int bns_inflater_create(int device, bns_inflater **out);
void bns_inflater_destroy(bns_inflater *h);
const char *bns_inflater_error(const bns_inflater *h);
float bns_inflater_last_kernel_ms(const bns_inflater *h);      /* HIP-event time of the last batch's kernel (< 0: none) */
int bns_inflater_host_alloc(bns_inflater *h, size_t bytes, void **out);
int bns_inflater_host_free(bns_inflater *h, void *p);
int bns_inflate_members(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                        const uint64_t *out_off, const uint32_t *out_len, uint64_t n_members, uint8_t *text, uint64_t text_bytes,
                        uint32_t *crc32, uint32_t *status);

/* The same batch with the text LEFT ON THE DEVICE: d_text (device memory of the same GPU, e.g. bns_dev_alloc) receives member i's
 * text at d_text + out_off[i]; only crc32[] and status[] come back.  What bns_classify_text(..., BNS_TEXT_DEVICE) then parses and
 * classifies where it lies: a BGZF file's text never crosses PCIe (compressed bytes up, results down). */
int bns_inflate_members_device(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                               const uint64_t *out_off, const uint32_t *out_len, uint64_t n_members, void *d_text, uint64_t text_bytes,
                               uint32_t *crc32, uint32_t *status);

/* ---- ONE plain gzip stream inflated on the device (csrc/bns_gzstream.hip).  Replaces gzread under kseq for a .gz file that is
 * not BGZF (kseq_declare.h:112-145, klib/kseq.h:177-225: one zlib inflate per file).  A DEFLATE stream has no entry points, but a
 * dynamic-Huffman block header is redundant enough to be found: the call's compressed bytes are cut into chunks, a wavefront per chunk
 * finds the first header in it and decodes from there to the next chunk's header -- writing 16-bit symbols, a byte or "byte j of the
 * 32 KiB in front of my entry point" --, one block then walks the chunks in order (each must end exactly where the next begins, else
 * the rest is left for the next call), resolves every chunk's 32 KiB window and the symbols become text, compacted into d_text.
 *   comp[0, comp_bytes)  HOST: a stretch of ONE gzip member's DEFLATE data (< 2 GiB); page-locked memory travels at the link rate
 *   start_bit            bit position in comp of a block header: the member's first block, or where the call before ended
 *   d_window             DEVICE: the 32 KiB of text in front of start_bit (NULL: the member starts there, nothing lies in front)
 *   d_text / text_cap    DEVICE: where the text goes, and its room
 *   d_window_out         DEVICE: receives the 32 KiB of text behind what the call took (the next call's d_window; may be d_window)
 * The call takes whole blocks only.  out->end_bit: behind the last block taken -- a block header (go on from there with more bytes:
 * comp of the next call must hold the stream from bit end_bit & ~7 on), or, with member_end, the first bit behind the member's final
 * block (the trailer follows at the next byte boundary: CRC-32 and ISIZE are the caller's to check, out->crc32 is the CRC-32 of the
 * text_bytes bytes this call wrote; bns_crc32_combine joins the calls of a member).  out->status != BNS_INF_OK: the first chunk
 * made no progress -- BNS_INF_IN_OVERRUN: its first block does not end inside comp (pass more bytes); BNS_INF_OUT_OVERFLOW: it
 * does not fit text_cap or the per-chunk room (BNS_GZ_RATIO_CAP x chunk, default 16 x 64 KiB of symbols); anything else: damaged
 * data.  Returns BNS_OK when the kernels ran. */
typedef struct bns_gz_result {
    uint64_t text_bytes;
    uint64_t end_bit;
    uint32_t member_end;
    uint32_t crc32;
    uint32_t n_chunks;      /* chunks that found a block header (the first one included) */
    uint32_t n_chained;     /* ... whose text this call took */
    uint32_t status;
    uint32_t stop_why;      /* 0 every chunk taken; 1 a chunk did not end at the next header (a false header: decoded again by the next call);
                               2 text room; 3 the member ended; 4 a chunk made no progress */
} bns_gz_result;
int bns_inflate_stream_device(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, uint64_t start_bit, const void *d_window, void *d_text,
                              uint64_t text_cap, void *d_window_out, bns_gz_result *out);
/* The device buffers of bns_inflate_stream_device sized NOW for calls of up to comp_bytes (symbols: 2 x BNS_GZ_RATIO_CAP bytes per byte
 * of the stream; a 256 MiB call: ~10 GB): a buffer that grows between two calls is freed and allocated again with the device drained. */
int bns_inflate_stream_reserve(bns_inflater *h, uint64_t comp_bytes);
/* A chunk's room for symbols, per byte of the chunk (2-1024; 0: the default, BNS_GZ_RATIO_CAP or 16).  A call whose first block inflates
 * beyond its chunk's room reports BNS_INF_OUT_OVERFLOW: the caller may ask again with more room and fewer bytes (the buffers are
 * chunks x room: eight times the room on an eighth of the bytes is the same memory) -- text that compresses 100:1 and more. */
int bns_inflate_stream_room(bns_inflater *h, uint32_t symbols_per_byte);
/* A range of host bytes brought up AHEAD, on a stream of its own (two ranges are kept): a later bns_inflate_stream_device call whose
 * comp[0, comp_bytes) lies inside it reads the bytes where they are instead of copying them up first -- the next call's upload under
 * this call's kernels.  The host bytes must not change between the prefetch and the calls that use it; a range is forgotten when the
 * second prefetch after it takes its buffer. */
int bns_inflate_stream_prefetch(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes);
/* zlib's crc32_combine: the CRC-32 of A followed by B from crc(A), crc(B) and B's length */
uint32_t bns_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

#ifdef __cplusplus
}
#endif
#endif /* BONSAI_AMD_H */
