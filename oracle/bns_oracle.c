/*
 * bns_oracle.c -- CPU restatement of the Bonsai classify hot path (see bns_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the product
 * library (bonsai_amd/csrc).  Callers: tests/, __graft_entry__.smoke(), bench.py cpu_baseline.
 *
 * Written from the reference's documented behaviour, function by function; citations are
 * `path:line` into dnbaker/bonsai.  Where the reference has undefined behaviour the
 * defined behaviour chosen here is marked DEFINED-BEHAVIOUR and mirrored by the HIP path.
 */
#include "bns_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ primitives */

/* khash64.h:202-211  Thomas Wang 64-bit mix used as the hash of khash_t(c) keys. */
uint64_t bo_wang64(uint64_t key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

/* kmerutil.h:83-90  reverse the order of the 2-bit symbols, complement, right-align to k symbols. */
uint64_t bo_revcomp(uint64_t kmer, unsigned k)
{
    kmer = ((kmer >> 2)  & 0x3333333333333333ULL) | ((kmer & 0x3333333333333333ULL) << 2);
    kmer = ((kmer >> 4)  & 0x0F0F0F0F0F0F0F0FULL) | ((kmer & 0x0F0F0F0F0F0F0F0FULL) << 4);
    kmer = ((kmer >> 8)  & 0x00FF00FF00FF00FFULL) | ((kmer & 0x00FF00FF00FF00FFULL) << 8);
    kmer = ((kmer >> 16) & 0x0000FFFF0000FFFFULL) | ((kmer & 0x0000FFFF0000FFFFULL) << 16);
    kmer = (kmer >> 32) | (kmer << 32);
    return (~kmer) >> (64 - (k << 1));
}

/* kmerutil.h:137-140 */
uint64_t bo_canonical(uint64_t kmer, unsigned k)
{
    const uint64_t rc = bo_revcomp(kmer, k);
    return kmer < rc ? kmer : rc;
}

/* alphabet.h:128 DNA4("A,C,G,T", aliases "U:T") through make_lut alphabet.h:30-59.
 * The alias loop stores arr[arr['T']] (= arr[3] = -1) into 'U'/'u', so U stays -1.
 * DEFINED-BEHAVIOUR: bytes >= 128 index the LUT with a negative `char` in the reference
 * (encoder.h:251-252, out of bounds); here they are non-ACGT (-1). */
int bo_dna4(unsigned char c)
{
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

/* spacer.h:14-19  comb size = k + sum of extra gaps */
uint32_t bo_comb_size(const uint16_t *gaps, unsigned k)
{
    uint32_t ret = k;
    if (gaps) for (unsigned i = 0; i + 1 < k; ++i) ret += gaps[i];
    return ret;
}

/* spacer.h:29-47  "a,b,c" or "gapxrepeat,..."; empty/NULL => k-1 zeros.  Returns #entries or -1. */
int bo_parse_spacing(const char *ss, unsigned k, uint16_t *out)
{
    if (!ss || *ss == '\0') { for (unsigned i = 0; i + 1 < k; ++i) out[i] = 0; return (int)k - 1; }
    int n = 0;
    char *p = (char *)ss;
    for (; *p; ++p) {
        const int j = (int)strtoul(p, &p, 10);
        if (n >= 63) return -1;
        out[n++] = (uint16_t)j;
        if (*p == 'x') {
            p = strchr(p, 'x') + 1;
            int rep = (int)strtoul(p, &p, 10) - 1;
            if (rep < 0) rep = 0;
            for (int r = 0; r < rep; ++r) { if (n >= 63) return -1; out[n++] = (uint16_t)j; }
        }
        p = strchr(p, ',');
        if (!p) break;
    }
    return n;
}

static int gaps_unspaced(const uint16_t *gaps, unsigned k)
{   /* Spacer::unspaced spacer.h:82-84 (offsets all 1 <=> extra gaps all 0) */
    if (!gaps) return 1;
    for (unsigned i = 0; i + 1 < k; ++i) if (gaps[i]) return 0;
    return 1;
}

/* ------------------------------------------------------------------ Encoder */

/* encoder.h:240-272 for_each_uncanon_unspaced_unwindowed (DNA branch), optionally wrapped by
 * for_each_canon_unwindowed encoder.h:218-232. */
static void enc_unspaced(const char *s, uint64_t l, unsigned k, int canon, bo_kmer_cb cb, void *ud)
{
    const uint64_t mask = ~UINT64_C(0) >> (64 - (k << 1));   /* rhtraits.h:55 */
    uint64_t pos = 0, min;
    unsigned filled;
loop_start:
    min = 0; filled = 0;
    while (pos < l) {
        while (filled < k && pos < l) {
            const int nv = bo_dna4((unsigned char)s[pos]);
            ++pos;
            if (nv < 0) goto loop_start;          /* encoder.h:254 */
            min = (min * 4) | (uint64_t)nv;       /* rhmul()==4 for DNA, rhtraits.h:70-72 */
            ++filled;
        }
        if (filled == k) {
            min &= mask;
            cb(canon ? bo_canonical(min, k) : min, ud);
            --filled;
        }
    }
}

/* encoder.h:547-592 kmer(start) for DNA: gather k symbols at cumulative offsets; -1 on any non-ACGT. */
static int enc_kmer_at(const char *s, uint64_t start, unsigned k, const uint16_t *gaps, uint64_t *out)
{
    int nv = bo_dna4((unsigned char)s[start]);
    if (nv < 0) return 0;
    uint64_t km = (uint64_t)nv;
    for (unsigned i = 0; i + 1 < k; ++i) {
        start += (uint64_t)(gaps ? gaps[i] : 0) + 1;       /* Spacer ctor converts gaps to offsets: spacer.h:64 */
        nv = bo_dna4((unsigned char)s[start]);
        if (nv < 0) return 0;
        km = (km << 2) | (uint64_t)nv;
    }
    *out = km;
    return 1;
}

/* encoder.h:233-239 with w == c (qmap of one entry, qmap.h:79-87 returns the element itself). */
void bo_for_each_uncanon_spaced(const char *s, uint64_t l, unsigned k, const uint16_t *gaps,
                                bo_kmer_cb cb, void *ud)
{
    const uint64_t c = bo_comb_size(gaps, k);
    uint64_t km;
    for (uint64_t pos = 0; pos + c - 1 < l; ++pos)       /* has_next_kmer encoder.h:594-597 */
        if (enc_kmer_at(s, pos, k, gaps, &km) && km != ~UINT64_C(0)) cb(km, ud);
}

/* encoder.h:415-442, Spacer(k, w=k, gaps): unspaced => unwindowed canonical/uncanonical stream;
 * spaced => ctor forced canonicalize_=false (encoder.h:148-150) and the branch at :437-440 is
 * guarded by if(canonicalize_) => nothing is emitted (defect F7). */
void bo_for_each(const char *s, uint64_t l, unsigned k, const uint16_t *gaps, int canon,
                 bo_kmer_cb cb, void *ud)
{
    const uint64_t c = bo_comb_size(gaps, k);
    if (!(c - 1 < l)) return;                             /* :418 */
    if (gaps_unspaced(gaps, k)) enc_unspaced(s, l, k, canon, cb, ud);
    /* else: F7 */
}

typedef struct { uint64_t *out; uint64_t cap, n; } collect_t;
static void collect_cb(uint64_t km, void *ud)
{
    collect_t *c = (collect_t *)ud;
    if (c->n < c->cap) c->out[c->n] = km;
    ++c->n;
}

uint64_t bo_encode(const char *s, uint64_t l, unsigned k, const uint16_t *gaps, int canon,
                   int spaced_intended, uint64_t *out, uint64_t cap)
{
    collect_t c = {out, cap, 0};
    if (spaced_intended && !gaps_unspaced(gaps, k)) bo_for_each_uncanon_spaced(s, l, k, gaps, collect_cb, &c);
    else bo_for_each(s, l, k, gaps, canon, collect_cb, &c);
    return c.n;
}

/* ------------------------------------------------------------------ windowed minimizers */

/* FRev64 = CEIFused<CEIXOR<0x533f8c2151b20f97>, CEIMul<0x9a98567ed20c127d>, RotL<31>, CEIXOR<0x691a9d706391077a>>
 * (encoder.h:47).  The combinators live in dnbaker/sketch hash.h (absent from the snapshot); restated from their
 * public definition: applied left to right, xor / multiply / rotate-left / xor.  PARITY UNPINNED. */
static uint64_t frev64(uint64_t h)
{
    h ^= UINT64_C(0x533f8c2151b20f97);
    h *= UINT64_C(0x9a98567ed20c127d);
    h = (h << 31) | (h >> 33);
    h ^= UINT64_C(0x691a9d706391077a);
    return h;
}

/* ent_score (encoder.h:55-58) with CircusEnt::value() == NOT_FULL (entropy.h:44-48): the u64 k-mer is
 * converted to double, divided by (-1 + 1e-4), and converted back to u64 through the signed path (x86-64 cvttsd2si). */
uint64_t bo_score(uint64_t kmer, int score_kind)
{
    if (score_kind == BO_SCORE_ENTROPY_PATH) {
        const double x = (double)kmer / (-1.0 + 1e-4);
        return (uint64_t)(int64_t)x;
    }
    return frev64(kmer);
}

uint64_t bo_encode_windowed(const char *s, uint64_t l, unsigned k, const uint16_t *gaps, unsigned w, int score_kind,
                            uint64_t *out, uint64_t cap)
{
    const uint64_t c = bo_comb_size(gaps, k);
    if (w <= c)                                                     /* Spacer: w_ = max(c, w) spacer.h:61; k_ == w_ is the */
        return bo_encode(s, l, k, gaps, 1, 1, out, cap);            /* unwindowed stream: canonical (encoder.h:420-421,448-449), or the spaced one of the path overloads */
    const uint64_t ws = (uint64_t)w - c + 1;                        /* QueueMap(sp_.w_ - sp_.c_ + 1) encoder.h:141 */
    const int spaced = !gaps_unspaced(gaps, k);
    if (l < c) return 0;
    const uint64_t npos = l - c + 1;
    uint64_t *el = (uint64_t *)malloc(npos * sizeof(uint64_t)), *sc = (uint64_t *)malloc(npos * sizeof(uint64_t));
    for (uint64_t p = 0; p < npos; ++p) {
        uint64_t km;
        if (!enc_kmer_at(s, p, k, gaps, &km)) km = ~UINT64_C(0);    /* ENCODE_OVERFLOW */
        /* contiguous seed: next_canonicalized_minimizer, encoder.h:622-628 (canonical_representation is applied to the
         * overflow value too, :625, -> 0).  Spaced seed: the constructor forces canonicalize_ = false (:148-150) and the path
         * overloads take for_each_uncanon_spaced -> next_minimizer (:233-239,615-620): the raw value, ~0 included. */
        if (spaced) { /* as is */ } else km = bo_canonical(km, k);
        el[p] = km; sc[p] = bo_score(km, score_kind);
    }
    uint64_t n = 0;
    for (uint64_t i = 0; i + ws <= npos; ++i) {                      /* list_.size() == wsz_ from the ws-th k-mer on */
        uint64_t b = i;
        for (uint64_t p = i + 1; p < i + ws; ++p)
            if (sc[p] < sc[b] || (sc[p] == sc[b] && el[p] < el[b])) b = p;     /* ElScore::operator< qmap.h:22-24 */
        if (el[b] != ~UINT64_C(0)) { if (n < cap) out[n] = el[b]; ++n; }      /* :215 skips ENCODE_OVERFLOW */
    }
    free(el); free(sc);
    return n;
}

/* Encoder::for_each_uncanon_unspaced_windowed (encoder.h:273-306): what the path overloads run for a contiguous seed with
 * canonicalize_ == false and w > k (`bonsai build -C -w ...`), and what the string overload runs for score::Lex.  Kept in
 * the shape of the reference's loop, because three of its details are not what a closed form would suggest:
 *   - the base is OR-ed into `min` BEFORE the validity test (:286-287): an invalid base (LUT -1, sign-extended) makes
 *     min == ENCODE_OVERFLOW and restarts the k-mer -- but so does a valid T that completes 2k+2 >= 64 one-bits: with
 *     k = 31 the 32nd T of a T run, with k = 32 the all-T 32-mer.  (`lutptr[..] != 'T'` compares a code 0..3 or -1 with
 *     the character 'T' and is always true, so the k = 32 exemption it was meant to be never applies.)
 *   - a restart does NOT reset the window queue (only assign() does, :201-204): windows run over the stream of emitted
 *     k-mers, across N gaps.
 *   - a sequence that never fills the window still emits one value, the minimum of what it has (:304-305). */
uint64_t bo_encode_uncanon_windowed(const char *s, uint64_t l, unsigned k, unsigned w, int score_kind,
                                    uint64_t *out, uint64_t cap)
{
    if (w <= k) return bo_encode(s, l, k, NULL, 0, 0, out, cap);    /* unwindowed: for_each_uncanon_unspaced_unwindowed */
    const uint64_t ws = (uint64_t)w - k + 1;
    const uint64_t mask = ~UINT64_C(0) >> (64 - (k << 1));
    uint64_t *q_el = (uint64_t *)malloc(ws * sizeof(uint64_t)), *q_sc = (uint64_t *)malloc(ws * sizeof(uint64_t));
    uint64_t q_n = 0, q_head = 0, n = 0;                             /* list_ as a ring of ws entries */
    uint64_t pos = 0, min;
    unsigned filled;
windowed_loop_start:
    min = 0; filled = 0;
    while (pos < l) {
        while (filled < k && pos < l) {
            min *= 4;
            min |= (uint64_t)(int64_t)bo_dna4((unsigned char)s[pos++]);      /* int8_t(-1) -> all ones */
            if (min == ~UINT64_C(0)) goto windowed_loop_start;
            ++filled;
        }
        if (filled == k) {
            min &= mask;
            /* qmap_.next_value(min, score) qmap.h:79-87 */
            if (q_n == ws) { q_head = (q_head + 1) % ws; --q_n; }            /* emplace_back then pop_front when over wsz_ */
            q_el[(q_head + q_n) % ws] = min; q_sc[(q_head + q_n) % ws] = bo_score(min, score_kind); ++q_n;
            if (q_n == ws) {
                uint64_t b = 0;
                for (uint64_t i = 1; i < ws; ++i)
                    if (q_sc[i] < q_sc[b] || (q_sc[i] == q_sc[b] && q_el[i] < q_el[b])) b = i;
                if (q_el[b] != ~UINT64_C(0)) { if (n < cap) out[n] = q_el[b]; ++n; }
            }
            --filled;
        }
    }
    if (q_n > 0 && q_n < ws) {                                       /* partially_full(): max_in_queue() is map_.begin() = the minimum */
        uint64_t b = q_head;
        for (uint64_t i = 1; i < q_n; ++i) {
            const uint64_t j = (q_head + i) % ws;
            if (q_sc[j] < q_sc[b] || (q_sc[j] == q_sc[b] && q_el[j] < q_el[b])) b = j;
        }
        if (n < cap) out[n] = q_el[b];
        ++n;
    }
    free(q_el); free(q_sc);
    return n;
}

/* double -> u64 as gcc compiles it for x86-64 without AVX-512 (the conversion of the score handed to QueueMap::next_value(T el,
 * T score), qmap.h:79): below 2^63 one cvttsd2si (negative values come out as their two's complement, NaN and anything below
 * -2^63 as the "integer indefinite" 0x8000000000000000); from 2^63 up cvttsd2si(x - 2^63) ^ 2^63 (so >= 2^64 gives 0). */
static uint64_t dbl_to_u64_x86(double x)
{
    const double two63 = 9223372036854775808.0;
    if (x != x) return UINT64_C(0x8000000000000000);
    if (x < two63) return x >= -two63 ? (uint64_t)(int64_t)x : UINT64_C(0x8000000000000000);
    const double y = x - two63;
    return (y < two63 ? (uint64_t)(int64_t)y : UINT64_C(0x8000000000000000)) ^ UINT64_C(0x8000000000000000);
}

/* CircusEnt::value() (entropy.h:44-48) once the tracker holds exactly the k-mer's k bases: sum over the symbols present of
 * (n/k) ln(n/k).  The reference adds the terms in ska::flat_hash_map iteration order (un-vendored: PARITY UNPINNED to the
 * last ulp of the sum); here they are added in the order A, C, G, T. */
double bo_kmer_entropy(uint64_t kmer, unsigned k)
{
    unsigned cnt[4] = {0, 0, 0, 0};
    for (unsigned i = 0; i < k; ++i) ++cnt[(kmer >> (2 * i)) & 3];
    const double qi = 1. / (double)k;
    double v = 0.;
    for (int c = 0; c < 4; ++c)
        if (cnt[c]) v = v + (double)cnt[c] * qi * log((double)cnt[c] * qi);
    return v;
}

/* Encoder<score::Entropy>::for_each(func, str, len) with a contiguous seed and w > k: for_each_uncanon_unspaced_windowed_entropy_
 * (encoder.h:307-346), wrapped by for_each_canon_unspaced_windowed_entropy_ (:347-353) when canonicalising.  The score of a
 * k-mer is (u64)(double(fwd_kmer) / (entropy + .001)) with the REAL entropy of its bases; selection runs on the FORWARD
 * k-mers and only the emitted value is canonicalised; an invalid base restarts k-mer and tracker (explicit -1 test: no
 * T-run restart here) but not the queue; a queue that never filled flushes its minimum. */
uint64_t bo_encode_windowed_entropy_str(const char *s, uint64_t l, unsigned k, unsigned w, int canon, uint64_t *out, uint64_t cap)
{
    if (w <= k) return bo_encode(s, l, k, NULL, canon, 0, out, cap);
    if (!(k - 1 < l)) return 0;                                      /* has_next_kmer() test of for_each, encoder.h:418 */
    const uint64_t ws = (uint64_t)w - k + 1;
    const uint64_t mask = ~UINT64_C(0) >> (64 - (k << 1));
    uint64_t *q_el = (uint64_t *)malloc(ws * sizeof(uint64_t)), *q_sc = (uint64_t *)malloc(ws * sizeof(uint64_t));
    uint64_t q_n = 0, q_head = 0, n = 0, pos = 0, min;
    unsigned filled;
windowed_loop_start:
    filled = 0; min = 0;
    while (pos < l) {
        while (filled < k && pos < l) {
            const int nc = bo_dna4((unsigned char)s[pos++]);
            if (nc < 0) goto windowed_loop_start;
            min = (4 * min) | (uint64_t)nc;
            ++filled;
        }
        if (filled == k) {
            min &= mask;
            const uint64_t sc = dbl_to_u64_x86((double)min / (bo_kmer_entropy(min, k) + .001));
            if (q_n == ws) { q_head = (q_head + 1) % ws; --q_n; }
            q_el[(q_head + q_n) % ws] = min; q_sc[(q_head + q_n) % ws] = sc; ++q_n;
            if (q_n == ws) {
                uint64_t b = 0;
                for (uint64_t i = 1; i < ws; ++i)
                    if (q_sc[i] < q_sc[b] || (q_sc[i] == q_sc[b] && q_el[i] < q_el[b])) b = i;
                if (q_el[b] != ~UINT64_C(0)) { if (n < cap) out[n] = canon ? bo_canonical(q_el[b], k) : q_el[b]; ++n; }
            }
            --filled;
        }
    }
    if (q_n > 0 && q_n < ws) {
        uint64_t b = q_head;
        for (uint64_t i = 1; i < q_n; ++i) {
            const uint64_t j = (q_head + i) % ws;
            if (q_sc[j] < q_sc[b] || (q_sc[j] == q_sc[b] && q_el[j] < q_el[b])) b = j;
        }
        if (n < cap) out[n] = canon ? bo_canonical(q_el[b], k) : q_el[b];
        ++n;
    }
    free(q_el); free(q_sc);
    return n;
}

/* ------------------------------------------------------------------ khash_t(c) */

/* flag macros khash64.h:169-177: 2 bits per slot, 16 slots per u32; bit1 = empty, bit0 = deleted */
#define FL_ISEMPTY(f, i)  (((f)[(i) >> 4] >> (((i) & 0xfU) << 1)) & 2)
#define FL_ISDEL(f, i)    (((f)[(i) >> 4] >> (((i) & 0xfU) << 1)) & 1)
#define FL_ISEITHER(f, i) (((f)[(i) >> 4] >> (((i) & 0xfU) << 1)) & 3)
#define FL_SET_DEL_TRUE(f, i)    ((f)[(i) >> 4] |= (uint32_t)(1ull << (((i) & 0xfU) << 1)))
#define FL_SET_EMPTY_FALSE(f, i) ((f)[(i) >> 4] &= (uint32_t)~(2ull << (((i) & 0xfU) << 1)))
#define FL_SET_BOTH_FALSE(f, i)  ((f)[(i) >> 4] &= (uint32_t)~(3ull << (((i) & 0xfU) << 1)))
#define FL_FSIZE(m) ((m) < 16 ? 1 : (m) >> 4)           /* khash64.h:179 */
static const double HASH_UPPER = 0.77;                  /* khash64.h:198 */

bo_khc_t *bo_khc_init(void) { return (bo_khc_t *)calloc(1, sizeof(bo_khc_t)); }

void bo_khc_destroy(bo_khc_t *h)
{
    if (h) { free(h->keys); free(h->flags); free(h->vals); free(h); }
}

void bo_khc_wrap(bo_khc_t *h, uint64_t n_buckets, uint64_t size, uint64_t n_occupied,
                 uint64_t upper_bound, uint32_t *flags, uint64_t *keys, uint32_t *vals)
{
    h->n_buckets = n_buckets; h->size = size; h->n_occupied = n_occupied; h->upper_bound = upper_bound;
    h->flags = flags; h->keys = keys; h->vals = vals;
}

/* khash64.h:250-263 */
uint64_t bo_khc_get(const bo_khc_t *h, uint64_t key)
{
    if (h->n_buckets) {
        uint64_t k, i, last, mask, step = 0;
        mask = h->n_buckets - 1;
        k = bo_wang64(key); i = k & mask;
        last = i;
        while (!FL_ISEMPTY(h->flags, i) && (FL_ISDEL(h->flags, i) || h->keys[i] != key)) {
            i = (i + (++step)) & mask;
            if (i == last) return h->n_buckets;
        }
        return FL_ISEITHER(h->flags, i) ? h->n_buckets : i;
    }
    return 0;
}

void bo_khc_get_batch(const bo_khc_t *h, const uint64_t *keys, uint64_t n, uint32_t *out_val, uint8_t *out_found)
{
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t i = bo_khc_get(h, keys[j]);
        const int f = (h->n_buckets != 0 && i != h->n_buckets);
        out_found[j] = (uint8_t)f;
        out_val[j] = f ? h->vals[i] : 0;
    }
}

static uint64_t roundup64(uint64_t x)
{   /* kroundup64 khash64.h:181-183 */
    --x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; x |= x >> 32; ++x;
    return x;
}

/* khash64.h:264-326 */
int bo_khc_resize(bo_khc_t *h, uint64_t new_n_buckets)
{
    uint32_t *new_flags = 0;
    uint64_t j = 1;
    {
        new_n_buckets = roundup64(new_n_buckets);
        if (new_n_buckets < 4) new_n_buckets = 4;
        if (h->size >= (uint64_t)(new_n_buckets * HASH_UPPER + 0.5)) j = 0;
        else {
            new_flags = (uint32_t *)malloc(FL_FSIZE(new_n_buckets) * sizeof(uint32_t));
            if (!new_flags) return -1;
            memset(new_flags, 0xaa, FL_FSIZE(new_n_buckets) * sizeof(uint32_t));
            if (h->n_buckets < new_n_buckets) {
                uint64_t *nk = (uint64_t *)realloc(h->keys, new_n_buckets * sizeof(uint64_t));
                if (!nk) { free(new_flags); return -1; }
                h->keys = nk;
                uint32_t *nv = (uint32_t *)realloc(h->vals, new_n_buckets * sizeof(uint32_t));
                if (!nv) { free(new_flags); return -1; }
                h->vals = nv;
            }
        }
    }
    if (j) {
        for (j = 0; j != h->n_buckets; ++j) {
            if (FL_ISEITHER(h->flags, j) == 0) {
                uint64_t key = h->keys[j];
                uint32_t val = h->vals[j];
                const uint64_t new_mask = new_n_buckets - 1;
                FL_SET_DEL_TRUE(h->flags, j);
                for (;;) {
                    uint64_t k, i, step = 0;
                    k = bo_wang64(key);
                    i = k & new_mask;
                    while (!FL_ISEMPTY(new_flags, i)) i = (i + (++step)) & new_mask;
                    FL_SET_EMPTY_FALSE(new_flags, i);
                    if (i < h->n_buckets && FL_ISEITHER(h->flags, i) == 0) {
                        { uint64_t t = h->keys[i]; h->keys[i] = key; key = t; }
                        { uint32_t t = h->vals[i]; h->vals[i] = val; val = t; }
                        FL_SET_DEL_TRUE(h->flags, i);
                    } else {
                        h->keys[i] = key;
                        h->vals[i] = val;
                        break;
                    }
                }
            }
        }
        if (h->n_buckets > new_n_buckets) {
            h->keys = (uint64_t *)realloc(h->keys, new_n_buckets * sizeof(uint64_t));
            h->vals = (uint32_t *)realloc(h->vals, new_n_buckets * sizeof(uint32_t));
        }
        free(h->flags);
        h->flags = new_flags;
        h->n_buckets = new_n_buckets;
        h->n_occupied = h->size;
        h->upper_bound = (uint64_t)(h->n_buckets * HASH_UPPER + 0.5);
    }
    return 0;
}

/* khash64.h:327-368 */
uint64_t bo_khc_put(bo_khc_t *h, uint64_t key, int *ret)
{
    uint64_t x;
    if (h->n_occupied >= h->upper_bound) {
        if (h->n_buckets > (h->size << 1)) {
            if (bo_khc_resize(h, h->n_buckets - 1) < 0) { *ret = -1; return h->n_buckets; }
        } else if (bo_khc_resize(h, h->n_buckets + 1) < 0) { *ret = -1; return h->n_buckets; }
    }
    {
        uint64_t k, i, site, last, mask = h->n_buckets - 1, step = 0;
        x = site = h->n_buckets; k = bo_wang64(key); i = k & mask;
        if (FL_ISEMPTY(h->flags, i)) x = i;
        else {
            last = i;
            while (!FL_ISEMPTY(h->flags, i) && (FL_ISDEL(h->flags, i) || h->keys[i] != key)) {
                if (FL_ISDEL(h->flags, i)) site = i;
                i = (i + (++step)) & mask;
                if (i == last) { x = site; break; }
            }
            if (x == h->n_buckets) {
                if (FL_ISEMPTY(h->flags, i) && site != h->n_buckets) x = site;
                else x = i;
            }
        }
    }
    if (FL_ISEMPTY(h->flags, x)) {
        h->keys[x] = key;
        FL_SET_BOTH_FALSE(h->flags, x);
        ++h->size; ++h->n_occupied;
        *ret = 1;
    } else if (FL_ISDEL(h->flags, x)) {
        h->keys[x] = key;
        FL_SET_BOTH_FALSE(h->flags, x);
        ++h->size;
        *ret = 2;
    } else *ret = 0;
    return x;
}

/* ------------------------------------------------------------------ taxonomy */

static int tax_reserve(bo_tax_t *t, uint32_t id)
{
    if (id >= (1u << 28)) return -1;           /* flat parent[] cap: 2^28 ids (documented deviation) */
    if (id >= t->n) {
        uint32_t nn = id + 1;
        uint32_t *p = (uint32_t *)realloc(t->parent, (size_t)nn * sizeof(uint32_t));
        if (!p) return -1;
        for (uint32_t i = t->n; i < nn; ++i) p[i] = BO_TAX_ABSENT;
        t->parent = p; t->n = nn;
    }
    return 0;
}

int bo_tax_from_pairs(const uint32_t *child, const uint32_t *parent, uint32_t n_pairs, bo_tax_t *out)
{
    out->n = 0; out->parent = NULL;
    uint32_t n_keys = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (parent[i] == BO_TAX_ABSENT) return -2;        /* malformed (reference only warns, util.h:778) */
        if (tax_reserve(out, child[i]) < 0) return -1;
        if (parent[i] != BO_TAX_ABSENT && tax_reserve(out, parent[i]) < 0) return -1;
        if (out->parent[child[i]] == BO_TAX_ABSENT) ++n_keys;
        out->parent[child[i]] = parent[i];               /* kh_put on an existing key overwrites the value */
    }
    if (tax_reserve(out, 1) < 0) return -1;
    if (out->parent[1] == BO_TAX_ABSENT) ++n_keys;
    out->parent[1] = 0;                                   /* util.h:780-781 root */
    if (n_keys < 2) return -3;                            /* util.h:782 */
    return 0;
}

/* util.h:766-785: child = atoi(line); parent = atoi(strchr(line,'|') + 2); skip lines starting with
 * '\0' or '#'; force parent[1] = 0; fewer than 2 entries is an error. */
int bo_tax_from_nodes_dmp(const char *path, bo_tax_t *out)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return -4;
    size_t cap = 1024, n = 0;
    uint32_t *ch = (uint32_t *)malloc(cap * 4), *pa = (uint32_t *)malloc(cap * 4);
    char *line = NULL; size_t lcap = 0; ssize_t len;
    while ((len = getline(&line, &lcap, fp)) >= 0) {
        if (len && line[len - 1] == '\n') line[--len] = 0;
        if (line[0] == '\0' || line[0] == '#') continue;
        const char *p = strchr(line, '|');
        if (n == cap) { cap <<= 1; ch = (uint32_t *)realloc(ch, cap * 4); pa = (uint32_t *)realloc(pa, cap * 4); }
        ch[n] = (uint32_t)atoi(line);
        pa[n] = p ? (uint32_t)atoi(p + 2) : BO_TAX_ABSENT;
        ++n;
    }
    free(line); fclose(fp);
    int rc = bo_tax_from_pairs(ch, pa, (uint32_t)n, out);
    free(ch); free(pa);
    return rc;
}

void bo_tax_free(bo_tax_t *t) { free(t->parent); t->parent = NULL; t->n = 0; }

static inline int tax_present(const bo_tax_t *t, uint32_t id) { return id < t->n && t->parent[id] != BO_TAX_ABSENT; }

/* util.h:634-663.  linear::set<tax_t> nodes == the ancestor chain of a, in order. */
uint32_t bo_lca(const bo_tax_t *t, uint32_t a, uint32_t b)
{
    if (a == b) return a;
    if (b == 0) return a;
    if (a == 0) return b;
    uint32_t chain_static[128], *chain = chain_static;
    size_t n = 0, cap = 128;
    uint32_t ret;
    while (a) {
        if (n == cap) {
            uint32_t *nc = (uint32_t *)malloc(cap * 2 * sizeof(uint32_t));
            memcpy(nc, chain, n * sizeof(uint32_t));
            if (chain != chain_static) free(chain);
            chain = nc; cap <<= 1;
        }
        chain[n++] = a;
        if (!tax_present(t, a)) { ret = 0xFFFFFFFFu; goto done; }   /* "Missing taxid" => (tax_t)-1 */
        a = t->parent[a];
    }
    while (b) {
        for (size_t i = 0; i < n; ++i) if (chain[i] == b) { ret = b; goto done; }
        if (!tax_present(t, b)) { ret = 0xFFFFFFFFu; goto done; }
        b = t->parent[b];
    }
    ret = 1;
done:
    if (chain != chain_static) free(chain);
    return ret;
}

/* ------------------------------------------------------------------ counter + resolve */

void bo_counter_init(bo_counter_t *c) { c->keys = NULL; c->vals = NULL; c->n = c->m = 0; }
void bo_counter_free(bo_counter_t *c) { free(c->keys); free(c->vals); bo_counter_init(c); }
void bo_counter_clear(bo_counter_t *c) { c->n = 0; }

/* linear.h:229-240 add(const K&, inc=1): linear search, append at the end, u16 += wraps. */
uint32_t bo_counter_add(bo_counter_t *c, uint32_t key)
{
    uint32_t i;
    for (i = 0; i < c->n; ++i) if (c->keys[i] == key) break;
    if (i == c->n) {
        if (c->n == c->m) {
            c->m = c->m ? c->m << 1 : 16;
            c->keys = (uint32_t *)realloc(c->keys, c->m * sizeof(uint32_t));
            c->vals = (uint16_t *)realloc(c->vals, c->m * sizeof(uint16_t));
        }
        c->keys[c->n] = key; c->vals[c->n] = 1; ++c->n;
    } else c->vals[i] = (uint16_t)(c->vals[i] + 1);
    return i;
}

/* linear.h:241-244 */
uint16_t bo_counter_count(const bo_counter_t *c, uint32_t key)
{
    for (uint32_t i = 0; i < c->n; ++i) if (c->keys[i] == key) return c->vals[i];
    return 0;
}

/* util.h:831-869.  score and max_score are tax_t (u32).  max_taxa is an insertion-ordered set.
 * DEFINED-BEHAVIOUR: when `node` is not a key of the parent map the reference reads
 * kh_val(map, kh_end) (out of bounds, util.h:843); here the node's own count is added and the walk stops. */
typedef struct { uint32_t stat[64], *v; uint32_t n, cap; } lset_t;   /* linear::set<tax_t>, linear.h:32-175 */
static void lset_init(lset_t *s) { s->v = s->stat; s->n = 0; s->cap = 64; }
static void lset_free(lset_t *s) { if (s->v != s->stat) free(s->v); }
static void lset_insert(lset_t *s, uint32_t val)
{   /* linear.h:94-97: append unless already present */
    for (uint32_t i = 0; i < s->n; ++i) if (s->v[i] == val) return;
    if (s->n == s->cap) {
        uint32_t *nv = (uint32_t *)malloc((size_t)s->cap * 2 * sizeof(uint32_t));
        memcpy(nv, s->v, (size_t)s->n * sizeof(uint32_t));
        lset_free(s);
        s->v = nv; s->cap <<= 1;
    }
    s->v[s->n++] = val;
}

uint32_t bo_resolve_tree(const bo_counter_t *c, const bo_tax_t *t)
{
    uint32_t max_taxon = 0, max_score = 0;
    lset_t max_taxa; lset_init(&max_taxa);
    for (uint32_t i = 0; i < c->n; ++i) {
        const uint32_t taxon = c->keys[i];
        uint32_t node = taxon, score = 0;
        while (node) {
            score += bo_counter_count(c, node);
            if (!tax_present(t, node)) break;
            node = t->parent[node];
        }
        if (score > max_score) {
            max_taxa.n = 0;
            max_score = score;
            max_taxon = taxon;
        } else if (score == max_score) {
            if (max_taxa.n == 0) lset_insert(&max_taxa, max_taxon);
            lset_insert(&max_taxa, taxon);
        }
    }
    if (max_taxa.n) {
        max_taxon = max_taxa.v[0];
        for (uint32_t i = 1; i < max_taxa.n; ++i) max_taxon = bo_lca(t, max_taxon, max_taxa.v[i]);
    }
    lset_free(&max_taxa);
    return max_taxon;
}

uint32_t bo_resolve_pairs(const uint32_t *keys, const uint16_t *counts, uint32_t n, const bo_tax_t *t)
{
    bo_counter_t c;
    c.keys = (uint32_t *)keys; c.vals = (uint16_t *)counts; c.n = c.m = n;
    return bo_resolve_tree(&c, t);
}

/* ------------------------------------------------------------------ classify_seq */

typedef struct {
    const bo_khc_t *db;
    bo_counter_t *hc;
    uint32_t missing, n_hits;
    uint32_t *hits; uint32_t hits_cap;
} cls_ctx_t;

/* the lambda at classifier.h:225-229 */
static void cls_cb(uint64_t kmer, void *ud)
{
    cls_ctx_t *x = (cls_ctx_t *)ud;
    const uint64_t ki = bo_khc_get(x->db, kmer);
    if (x->db->n_buckets == 0 || ki == x->db->n_buckets) ++x->missing;
    else {
        const uint32_t v = x->db->vals[ki];
        if (x->hits && x->n_hits < x->hits_cap) x->hits[x->n_hits] = v;
        ++x->n_hits;
        bo_counter_add(x->hc, v);
    }
}

static void classify_with(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps,
                          int canon, int spaced_intended, const char *s1, uint32_t l1,
                          const char *s2, uint32_t l2, bo_result_t *res, uint32_t *hits, uint32_t hits_cap,
                          bo_counter_t *hc)
{
    const int spaced = !gaps_unspaced(gaps, k);
    const uint32_t c = bo_comb_size(gaps, k);
    cls_ctx_t x = {db, hc, 0, 0, hits, hits_cap};
    bo_counter_clear(hc);
    if (spaced && spaced_intended) bo_for_each_uncanon_spaced(s1, l1, k, gaps, cls_cb, &x);
    else bo_for_each(s1, l1, k, gaps, canon, cls_cb, &x);
    uint32_t ambig = l1 - c + 1 - x.n_hits - x.missing;                     /* classifier.h:232 */
    if (s2) {
        if (spaced && spaced_intended) bo_for_each_uncanon_spaced(s2, l2, k, gaps, cls_cb, &x);
        else bo_for_each(s2, l2, k, gaps, canon, cls_cb, &x);
        ambig += l2 - (c - 1) - x.n_hits - x.missing;                       /* classifier.h:235 (cumulative totals) */
    }
    res->taxon = bo_resolve_tree(hc, tax);                                   /* classifier.h:238 */
    res->missing = x.missing; res->ambig = ambig; res->n_hits = x.n_hits;
}

void bo_classify_seq(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps,
                     int canon, int spaced_intended,
                     const char *s1, uint32_t l1, const char *s2, uint32_t l2,
                     bo_result_t *res, uint32_t *hits, uint32_t hits_cap)
{
    bo_counter_t hc; bo_counter_init(&hc);
    classify_with(db, tax, k, gaps, canon, spaced_intended, s1, l1, s2, l2, res, hits, hits_cap, &hc);
    bo_counter_free(&hc);
}

void bo_classify_batch(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps,
                       int canon, int spaced_intended, int paired,
                       const char *bases, const uint64_t *offsets, uint64_t n_reads,
                       bo_result_t *res, int nthreads)
{
    const int64_t inc = paired ? 2 : 1;
    const int64_t n_units = (int64_t)(n_reads / (uint64_t)inc);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        bo_counter_t hc; bo_counter_init(&hc);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t u = 0; u < n_units; ++u) {
            const uint64_t r = (uint64_t)(u * inc);
            const char *s1 = bases + offsets[r];
            const uint32_t l1 = (uint32_t)(offsets[r + 1] - offsets[r]);
            const char *s2 = paired ? bases + offsets[r + 1] : NULL;
            const uint32_t l2 = paired ? (uint32_t)(offsets[r + 2] - offsets[r + 1]) : 0;
            classify_with(db, tax, k, gaps, canon, spaced_intended, s1, l1, s2, l2, &res[u], NULL, 0, &hc);
        }
        bo_counter_free(&hc);
    }
    (void)nthreads;
}

/* CPU-baseline calibration (tools/cpu_calibrate.py): the batch loop above cut after a phase, so that the port's per-phase cost
 * can be set beside the reference-compiled path's (oracle/ref_harness.cpp ref_classify_batch, same phases).
 * phase 0 = encode only (k-mers summed), 1 = encode + kh_get (hit values summed), 2 = bo_classify_batch.  Single-end. */
typedef struct { const bo_khc_t *db; uint64_t acc; } phase_ctx_t;
static void phase0_cb(uint64_t kmer, void *ud) { ((phase_ctx_t *)ud)->acc += kmer; }
static void phase1_cb(uint64_t kmer, void *ud)
{
    phase_ctx_t *x = (phase_ctx_t *)ud;
    const uint64_t ki = bo_khc_get(x->db, kmer);
    x->acc += (x->db->n_buckets == 0 || ki == x->db->n_buckets) ? 1u : x->db->vals[ki];
}
void bo_classify_batch_phase(const bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps, int canon,
                             const char *bases, const uint64_t *offsets, uint64_t n_reads, bo_result_t *res, int nthreads,
                             int phase, uint64_t *sink)
{
    if (phase >= 2) { bo_classify_batch(db, tax, k, gaps, canon, 0, 0, bases, offsets, n_reads, res, nthreads); return; }
    uint64_t total = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+:total)
#endif
    for (int64_t r = 0; r < (int64_t)n_reads; ++r) {
        phase_ctx_t x = {db, 0};
        bo_for_each(bases + offsets[r], (uint32_t)(offsets[r + 1] - offsets[r]), k, gaps, canon, phase == 0 ? phase0_cb : phase1_cb, &x);
        total += x.acc;
    }
    if (sink) *sink = total;
}

/* ------------------------------------------------------------------ Kraken line */

static size_t put_u(char *b, size_t pos, size_t cap, uint32_t x)
{   /* ks.h:337-354 putuw_: decimal, "0" for zero */
    char tmp[16]; int len = 0;
    if (x == 0) tmp[len++] = '0';
    for (; x > 0; x /= 10) tmp[len++] = (char)('0' + x % 10);
    while (len) { if (pos < cap) b[pos] = tmp[len - 1]; ++pos; --len; }
    return pos;
}
static size_t put_i(char *b, size_t pos, size_t cap, int c)
{   /* ks.h:318-336 putw_ */
    char tmp[16]; int len = 0; unsigned x = (unsigned)c;
    if (c < 0) x = -x;
    do { tmp[len++] = (char)('0' + x % 10); x /= 10; } while (x > 0);
    if (c < 0) tmp[len++] = '-';
    while (len) { if (pos < cap) b[pos] = tmp[len - 1]; ++pos; --len; }
    return pos;
}
#define PUTC(ch) do { if (pos < cap) buf[pos] = (ch); ++pos; } while (0)

/* classifier.h:112-129 append_kraken_classification, :63-70 append_counts, :45-61 append_taxa_runs,
 * :30-42 append_taxa_run. */
size_t bo_kraken_line(char *buf, size_t cap, const char *name, uint32_t taxon, int l_seq,
                      uint32_t missing, uint32_t ambig, const uint32_t *hits, uint32_t n_hits)
{
    size_t pos = 0;
    PUTC(taxon ? 'C' : 'U'); PUTC('\t');
    for (const char *p = name; *p; ++p) PUTC(*p);
    PUTC('\t');
    pos = put_u(buf, pos, cap, taxon); PUTC('\t');
    pos = put_i(buf, pos, cap, l_seq); PUTC('\t');
    if (missing) { PUTC('M'); PUTC(':'); pos = put_u(buf, pos, cap, missing); PUTC('\t'); }
    if (ambig)   { PUTC('A'); PUTC(':'); pos = put_u(buf, pos, cap, ambig);   PUTC('\t'); }
    if (taxon) {
        uint32_t last = hits[0], run = 1;
        for (uint32_t i = 1; i <= n_hits; ++i) {
            if (i < n_hits && hits[i] == last) { ++run; continue; }
            if (last == 0) PUTC('U');
            else if (last == 0xFFFFFFFFu) PUTC('A');
            else pos = put_u(buf, pos, cap, last);
            PUTC(':'); pos = put_u(buf, pos, cap, run); PUTC('\t');
            if (i < n_hits) { last = hits[i]; run = 1; }
        }
        if (pos - 1 < cap) buf[pos - 1] = '\n';     /* classifier.h:58: trailing tab -> newline */
    } else { PUTC('0'); PUTC(':'); PUTC('0'); PUTC('\n'); }
    return pos;
}
#undef PUTC

/* ------------------------------------------------------------------ db construction */

typedef struct { bo_khc_t *db; const bo_tax_t *tax; uint32_t taxid; } lca_ctx_t;

/* feature_min.h:205-228 update_lca_map applied k-mer by k-mer.  The reference first collects one
 * genome's k-mers into a set (fill_set_genome :67-82) so each distinct k-mer is applied once per
 * genome; applying duplicates is idempotent because val==taxid after the first application
 * (lca(t, x, x) is skipped by the `!= taxid` test, and lca(taxid, lca(taxid, old)) == lca(taxid, old)). */
static void lca_cb(uint64_t km, void *ud)
{
    lca_ctx_t *x = (lca_ctx_t *)ud;
    uint64_t k2 = bo_khc_get(x->db, km);
    if (x->db->n_buckets == 0 || k2 == x->db->n_buckets) {
        int khr;
        k2 = bo_khc_put(x->db, km, &khr);
        x->db->vals[k2] = x->taxid;
    } else if (x->db->vals[k2] != x->taxid) {
        x->db->vals[k2] = bo_lca(x->tax, x->taxid, x->db->vals[k2]);
    }
}

void bo_lca_map_add(bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps, int canon,
                    const char *seq, uint64_t len, uint32_t taxid)
{
    lca_ctx_t x = {db, tax, taxid};
    if (!gaps_unspaced(gaps, k)) bo_for_each_uncanon_spaced(seq, len, k, gaps, lca_cb, &x);
    else bo_for_each(seq, len, k, gaps, canon, lca_cb, &x);
}

void bo_lca_map_add_windowed(bo_khc_t *db, const bo_tax_t *tax, unsigned k, const uint16_t *gaps, unsigned w, int score_kind,
                             int canon, const char *seq, uint64_t len, uint32_t taxid)
{
    uint64_t cap = len + 1;
    uint64_t *buf = (uint64_t *)malloc(cap * sizeof(uint64_t));
    /* Encoder::for_each(func, path) encoder.h:497-504: canonicalize_ picks for_each_canon / for_each_uncanon; a spaced seed
     * has canonicalize_ forced off (:148-150) and goes through for_each_uncanon_spaced, which bo_encode_windowed covers */
    const uint64_t n = (!canon && gaps_unspaced(gaps, k)) ? bo_encode_uncanon_windowed(seq, len, k, w, score_kind, buf, cap)
                                                          : bo_encode_windowed(seq, len, k, gaps, w, score_kind, buf, cap);
    lca_ctx_t x = {db, tax, taxid};
    for (uint64_t i = 0; i < n; ++i) lca_cb(buf[i], &x);
    free(buf);
}

/* util.h:898-929 get_taxid: which part of the first header line names the genome */
void bo_genome_name(const char *line, char *out, size_t cap)
{
    const char *p = line, *e;
    if (strchr(p, '|')) {
        const char *last = strrchr(p, '|');
        const char *q = last;
        while (q > p && *--q != '|') {}
        p = (*q == '|') ? q + 1 : q;
        e = strchr(p, '|');
    } else {
        e = p;
        while (*e && !(*e == ' ' || *e == '\t' || *e == '\n' || *e == '\r' || *e == '\v' || *e == '\f')) ++e;
    }
    size_t n = (size_t)(e - p);
    if (n >= cap) n = cap - 1;
    memcpy(out, p, n);
    out[n] = 0;
}

/* ------------------------------------------------------------------ bns.db IO */

/* database.h:81-102 (header) + util.h:280-293 (khash dump).  Layout, little-endian:
 *   u32 k; u32 w; spacing[k-1] (u8 as read at database.h:46-48, u16 as the gz writer emits at :89);
 *   u64 n_buckets, n_occupied, size, upper_bound; u32 flags[fsize]; u64 keys[n_buckets]; u32 vals[n_buckets].
 * A ".gz" suffix selects whole-file gzip. */
int bo_db_write(const char *path, uint32_t k, uint32_t w, const uint16_t *gaps, int spacing_width, bo_khc_t *db)
{
    for (uint64_t i = 0; i < db->n_buckets; ++i)
        if (FL_ISEITHER(db->flags, i)) { db->keys[i] = 0; db->vals[i] = 0; }     /* util.h:282-284 */
    const size_t plen = strlen(path);
    const int gz = plen > 3 && strcmp(path + plen - 3, ".gz") == 0;
    gzFile fp = gzopen(path, gz ? "wb" : "wbT");      /* 'T' = transparent (no compression) */
    if (!fp) return -1;
    gzwrite(fp, &k, 4); gzwrite(fp, &w, 4);
    for (uint32_t i = 0; i + 1 < k; ++i) {
        if (spacing_width == 1) { uint8_t b = (uint8_t)gaps[i]; gzwrite(fp, &b, 1); }
        else gzwrite(fp, &gaps[i], 2);
    }
    gzwrite(fp, &db->n_buckets, 8); gzwrite(fp, &db->n_occupied, 8);
    gzwrite(fp, &db->size, 8); gzwrite(fp, &db->upper_bound, 8);
    const uint64_t fs = FL_FSIZE(db->n_buckets);
    /* gzwrite takes unsigned len: chunk large arrays */
    const char *ptrs[3] = {(const char *)db->flags, (const char *)db->keys, (const char *)db->vals};
    const uint64_t lens[3] = {fs * 4, db->n_buckets * 8, db->n_buckets * 4};
    for (int a = 0; a < 3; ++a)
        for (uint64_t off = 0; off < lens[a];) {
            const uint64_t chunk = lens[a] - off > (1u << 30) ? (1u << 30) : lens[a] - off;
            if (gzwrite(fp, ptrs[a] + off, (unsigned)chunk) != (int)chunk) { gzclose(fp); return -2; }
            off += chunk;
        }
    gzclose(fp);
    return 0;
}

static int gzread_full(gzFile fp, void *dst, uint64_t n)
{
    char *d = (char *)dst;
    while (n) {
        const unsigned chunk = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        const int r = gzread(fp, d, chunk);
        if (r <= 0) return -1;
        d += r; n -= (uint64_t)r;
    }
    return 0;
}

/* Reads both spacing widths.  The width is disambiguated by trying width 1 then width 2 and keeping
 * the one whose khash header is self-consistent (n_buckets a power of two, size <= n_occupied <=
 * n_buckets, upper_bound == (u64)(n_buckets*0.77+0.5)); SURVEY 8a row 8. */
int bo_db_read(const char *path, uint32_t *k, uint32_t *w, uint16_t *gaps, bo_khc_t *db)
{
    for (int width = 1; width <= 2; ++width) {
        gzFile fp = gzopen(path, "rb");
        if (!fp) return -1;
        uint64_t hdr[4];
        int ok = gzread_full(fp, k, 4) == 0 && gzread_full(fp, w, 4) == 0 && *k >= 1 && *k <= 32;
        for (uint32_t i = 0; ok && i + 1 < *k; ++i) {
            if (width == 1) { uint8_t b; ok = gzread_full(fp, &b, 1) == 0; gaps[i] = b; }
            else ok = gzread_full(fp, &gaps[i], 2) == 0;
        }
        ok = ok && gzread_full(fp, hdr, 32) == 0;
        if (ok) {
            const uint64_t nb = hdr[0], nocc = hdr[1], sz = hdr[2], ub = hdr[3];
            ok = nb && (nb & (nb - 1)) == 0 && sz <= nocc && nocc <= nb && ub == (uint64_t)(nb * HASH_UPPER + 0.5);
            if (ok) {
                db->n_buckets = nb; db->n_occupied = nocc; db->size = sz; db->upper_bound = ub;
                const uint64_t fs = FL_FSIZE(nb);
                db->flags = (uint32_t *)malloc(fs * 4);
                db->keys = (uint64_t *)malloc(nb * 8);
                db->vals = (uint32_t *)malloc(nb * 4);
                ok = db->flags && db->keys && db->vals &&
                     gzread_full(fp, db->flags, fs * 4) == 0 &&
                     gzread_full(fp, db->keys, nb * 8) == 0 &&
                     gzread_full(fp, db->vals, nb * 4) == 0;
                char extra;
                if (ok && gzread(fp, &extra, 1) == 1) ok = 0;     /* trailing bytes => wrong width */
                if (!ok) { free(db->flags); free(db->keys); free(db->vals); db->flags = NULL; db->keys = NULL; db->vals = NULL; }
            }
        }
        gzclose(fp);
        if (ok) return width;
    }
    return -2;
}

/* ------------------------------------------------------------------ RollingHasher
 * RollingHasher<u64, CyclicHash<u64>> without a window (wsz = -1, the constructor default; encoder.h:672-684): k
 * characters are folded into a 64-bit word by  h = rotl1(h) ^ T[c]  (CyclicHash::eat, rollinghash/cyclichash.h:137-140)
 * and rolled by  h = rotl1(h) ^ rotl_{k mod 64}(T[out]) ^ T[in]  (update, :122-128).  The canonical path keeps a second
 * hasher over the reverse strand and emits min(forward, reverse).
 *
 * PARITY UNPINNED: the 256-entry character tables come from wy::WyRand (aesctr/wy.h, un-vendored, version
 * unpinned; characterhash.h:90-103): the tables are therefore an INPUT here.  bo_rolling_tables() fills them from a
 * generator restated from its public definition (Lemire's wyhash64: state += 0x60bee2bee120fc15, two 64x64->128 multiply
 * folds) seeded as the reference seeds its two hashers (encoder.h:682-683: seed1 ^ seed2, and seed2*seed1 ^ (seed2 ^ seed1),
 * both truncated to 32 bits by CharacterHash::seed, characterhash.h:85); a caller holding the reference's real tables can
 * pass those instead and everything downstream is the reference's arithmetic.
 *
 * Restated as written, quirks included:
 *  - an invalid character at i skips to i + k + 1 and restarts (encoder.h:713-718,771-773: `i += k_` in the loop body
 *    plus the loop's own ++i); the canonical path first gives up if i + 2k >= l (:714), the other path never does;
 *  - while the canonical path fills its reverse hasher it eats cstr_rc_lut[s[i - nf + k - 1]] (:721): i - nf is the
 *    window start for the whole fill, so that is the complement of the window's LAST base, k times over;
 *  - the roll is reverse_update(rc(s[i]), rc(s[i-k])) (:730; cyclichash.h:130-135): h ^= rotl_{k mod 64}(T[rc s_i]) ^
 *    T[rc s_{i-k}], then rotate right by one. */
static uint64_t rotl64(uint64_t x, unsigned r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }
static uint64_t rotr64(uint64_t x, unsigned r) { r &= 63; return r ? (x >> r) | (x << (64 - r)) : x; }

static uint64_t wyhash64_next(uint64_t *state)
{
    *state += UINT64_C(0x60bee2bee120fc15);
    __uint128_t t = (__uint128_t)*state * UINT64_C(0xa3b195354a39b70d);
    const uint64_t m1 = (uint64_t)(t >> 64) ^ (uint64_t)t;
    t = (__uint128_t)m1 * UINT64_C(0x1b03738712fad5c9);
    return (uint64_t)(t >> 64) ^ (uint64_t)t;
}

void bo_rolling_tables(uint64_t seed1, uint64_t seed2, uint64_t *fwd, uint64_t *rc)
{
    uint64_t sf = (uint32_t)(seed1 ^ seed2);                        /* hasher_.seed(seed1, seed2) */
    uint64_t sr = (uint32_t)((seed2 * seed1) ^ (seed2 ^ seed1));    /* rchasher_.seed(seed2 * seed1, seed2 ^ seed1) */
    for (int i = 0; i < 256; ++i) fwd[i] = wyhash64_next(&sf);      /* wordsize 64: every draw is accepted (characterhash.h:92-97) */
    for (int i = 0; i < 256; ++i) rc[i] = wyhash64_next(&sr);
}

static int rc_code(unsigned char c) { const int v = bo_dna4(c); return v < 0 ? 255 : 3 - v; }   /* cstr_rc_lut, -1 -> (unsigned char)255 */

/* The two loops of RollingHasher::for_each_canon / for_each_uncanon (encoder.h:692-796) with the per-position action left
 * to the caller: use(h, g) gets the forward hash and (canonical path) the reverse-complement hash. */
typedef void (*rh_use_fn)(uint64_t h, uint64_t g, void *ud);
static void rh_core(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *fwd, const uint64_t *rc, rh_use_fn use, void *ud)
{
    if (l < k || k == 0) return;
    const unsigned myr = k % 64;
    uint64_t i = 0;
    for (;;) {
        uint64_t h = 0, g = 0;
        unsigned nf = 0;
        while (nf < k && i < l) {                                   /* the fill loop, encoder.h:711-722 / 770-775 */
            const int v = bo_dna4((unsigned char)s[i]);
            if (v < 0) {
                if (canon && i + 2 * (uint64_t)k >= l) return;
                i += k; nf = 0; h = 0; g = 0;
            } else {
                h = rotl64(h, 1) ^ fwd[v];
                if (canon) g = rotl64(g, 1) ^ rc[rc_code((unsigned char)s[i - nf + k - 1])];
                ++nf;
            }
            ++i;
        }
        if (nf < k) return;
        use(h, g, ud);
        int restart = 0;
        for (; i < l; ++i) {                                        /* the roll, encoder.h:726-732 / 778-783 */
            const int v = bo_dna4((unsigned char)s[i]);
            if (v < 0) { restart = 1; break; }
            h = rotl64(h, 1) ^ rotl64(fwd[bo_dna4((unsigned char)s[i - k])], myr) ^ fwd[v];
            if (canon) {
                g ^= rotl64(rc[rc_code((unsigned char)s[i])], myr) ^ rc[rc_code((unsigned char)s[i - k])];
                g = rotr64(g, 1);
            }
            use(h, g, ud);
        }
        if (!restart) return;
        if (canon && i + 2 * (uint64_t)k >= l) return;              /* `goto fixup` lands on the same test */
        i += (uint64_t)k + 1;                                       /* i += k_, then the fill loop's ++i */
    }
}

typedef struct { uint64_t *out, cap, n; int canon; } rh_plain_t;
static void rh_plain_use(uint64_t h, uint64_t g, void *ud)
{
    rh_plain_t *x = (rh_plain_t *)ud;
    if (x->n < x->cap) x->out[x->n] = x->canon ? (h < g ? h : g) : h;   /* func(std::min(hasher_.hashvalue, rchasher_.hashvalue)) :742,749 */
    ++x->n;
}

uint64_t bo_rolling_hash(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *fwd, const uint64_t *rc,
                         uint64_t *out, uint64_t cap)
{
    rh_plain_t x = {out, cap, 0, canon};
    rh_core(s, l, k, canon, fwd, rc, rh_plain_use, &x);
    return x.n;
}

/* ---- RollingHasher<__uint128_t, CyclicHash<__uint128_t>> without a window (the instantiation test/encoding.cpp:152 uses):
 * the same two loops over a 128-bit word -- rotations are 128-bit, myr = k % 128.  Values and table entries travel as
 * (lo, hi) pairs of u64.  Default tables: CharacterHash<u128>::clear_hashvalues draws TWO 64-bit words per entry
 * (`hack`, characterhash.h:72-74: x = rng() << 64 | rng()) and then masks with tmaxval = roundup(maxval_) - 1, where maxval_ is
 * a uint64_t MEMBER (characterhash.h:82,84,90): the 128-bit mask was truncated to 2^64 - 1 on the way in, roundup of that wraps
 * to 0, minus one is 2^64 - 1 again -- so only the low word (the SECOND draw) survives and every default entry has hi = 0.
 * PARITY UNPINNED as for the 64-bit hasher (wy::WyRand un-vendored); the tables are an input. */
typedef unsigned __int128 u128_t;
static inline u128_t rotl128(u128_t x, unsigned r) { r &= 127u; return r ? (x << r) | (x >> (128u - r)) : x; }
static inline u128_t rotr128(u128_t x, unsigned r) { r &= 127u; return r ? (x >> r) | (x << (128u - r)) : x; }
static inline u128_t ld128(const uint64_t *t, int i) { return ((u128_t)t[2 * i + 1] << 64) | t[2 * i]; }

void bo_rolling_tables128(uint64_t seed1, uint64_t seed2, uint64_t *fwd_lohi, uint64_t *rc_lohi)
{
    uint64_t sf = (uint32_t)(seed1 ^ seed2);
    uint64_t sr = (uint32_t)((seed2 * seed1) ^ (seed2 ^ seed1));
    for (int i = 0; i < 256; ++i) { (void)wyhash64_next(&sf); fwd_lohi[2 * i] = wyhash64_next(&sf); fwd_lohi[2 * i + 1] = 0; }
    for (int i = 0; i < 256; ++i) { (void)wyhash64_next(&sr); rc_lohi[2 * i] = wyhash64_next(&sr); rc_lohi[2 * i + 1] = 0; }
}

/* score of a 128-bit hash in the windowed hasher's queue: the reference's lex_score(u128) is sketch::hash::CEHasher
 * (encoder.h:49-51), a class of the un-vendored sketch library, truncated to the queue's u64 score type (QueueMap<IntType,
 * uint64_t>, encoder.h:654).  PARITY UNPINNED: restated as FRev64 over the folded halves.  What the reference's own test pins for
 * this instantiation is the NUMBER of values (test/encoding.cpp:152-156: len - w + 1), which no score function changes. */
static uint64_t frev64(uint64_t x);
static inline uint64_t lex_score128(u128_t v) { return frev64((uint64_t)v ^ frev64((uint64_t)(v >> 64))); }

/* the queue of the windowed 128-bit hasher (as rh_win_t for the 64-bit one) */
typedef struct { uint64_t *out, cap, n; uint64_t ws, q_n, q_head; u128_t *q_el; uint64_t *q_sc; } rh_win128_t;
static void rh_win128_push(rh_win128_t *x, u128_t v)
{
    if (x->q_n == x->ws) { x->q_head = (x->q_head + 1) % x->ws; --x->q_n; }
    const uint64_t at = (x->q_head + x->q_n) % x->ws;
    x->q_el[at] = v; x->q_sc[at] = lex_score128(v); ++x->q_n;
    if (x->q_n == x->ws) {
        uint64_t b = 0;
        for (uint64_t i = 1; i < x->ws; ++i)
            if (x->q_sc[i] < x->q_sc[b] || (x->q_sc[i] == x->q_sc[b] && x->q_el[i] < x->q_el[b])) b = i;
        if (x->q_el[b] != ~(u128_t)0) {
            if (x->n < x->cap) { x->out[2 * x->n] = (uint64_t)x->q_el[b]; x->out[2 * x->n + 1] = (uint64_t)(x->q_el[b] >> 64); }
            ++x->n;
        }
    }
}

static uint64_t rolling_hash128_core(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *fwd, const uint64_t *rc,
                                     uint64_t *out_lohi, uint64_t cap, rh_win128_t *win)
{
    uint64_t n = 0;
#define RH128_USE() do { if (win) { rh_win128_push(win, h); if (canon) rh_win128_push(win, g); } else { \
        const u128_t v_ = canon ? (h < g ? h : g) : h; if (n < cap) { out_lohi[2 * n] = (uint64_t)v_; out_lohi[2 * n + 1] = (uint64_t)(v_ >> 64); } ++n; } } while (0)
    if (l < k || k == 0) return 0;
    const unsigned myr = k % 128;
    uint64_t i = 0;
    for (;;) {
        u128_t h = 0, g = 0;
        unsigned nf = 0;
        while (nf < k && i < l) {                                   /* the fill loop, encoder.h:711-722 / 770-775 */
            const int v = bo_dna4((unsigned char)s[i]);
            if (v < 0) {
                if (canon && i + 2 * (uint64_t)k >= l) return n;
                i += k; nf = 0; h = 0; g = 0;
            } else {
                h = rotl128(h, 1) ^ ld128(fwd, v);
                if (canon) g = rotl128(g, 1) ^ ld128(rc, rc_code((unsigned char)s[i - nf + k - 1]));
                ++nf;
            }
            ++i;
        }
        if (nf < k) return n;
        RH128_USE();
        int restart = 0;
        for (; i < l; ++i) {                                        /* the roll, encoder.h:726-732 / 778-783 */
            const int v = bo_dna4((unsigned char)s[i]);
            if (v < 0) { restart = 1; break; }
            h = rotl128(h, 1) ^ rotl128(ld128(fwd, bo_dna4((unsigned char)s[i - k])), myr) ^ ld128(fwd, v);
            if (canon) {
                g ^= rotl128(ld128(rc, rc_code((unsigned char)s[i])), myr) ^ ld128(rc, rc_code((unsigned char)s[i - k]));
                g = rotr128(g, 1);
            }
            RH128_USE();
        }
        if (!restart) return n;
        if (canon && i + 2 * (uint64_t)k >= l) return n;
        i += (uint64_t)k + 1;
    }
#undef RH128_USE
}

uint64_t bo_rolling_hash128(const char *s, uint64_t l, unsigned k, int canon, const uint64_t *fwd, const uint64_t *rc,
                            uint64_t *out_lohi, uint64_t cap)
{
    return rolling_hash128_core(s, l, k, canon, fwd, rc, out_lohi, cap, NULL);
}

/* RollingHasher<__uint128_t>(k, canon, DNA, w) with w > k (the windowed form test/encoding.cpp:152 constructs; window branches
 * encoder.h:706-736 / 771-795): as bo_rolling_hash_windowed over 128-bit values -- both strands queued separately on the canonical
 * path, the queue surviving restarts, one flushed value for a stream that never fills it. */
uint64_t bo_rolling_hash128_windowed(const char *s, uint64_t l, unsigned k, int canon, unsigned w, const uint64_t *fwd,
                                     const uint64_t *rc, uint64_t *out_lohi, uint64_t cap)
{
    if (w <= k) return bo_rolling_hash128(s, l, k, canon, fwd, rc, out_lohi, cap);
    rh_win128_t x = {out_lohi, cap, 0, (uint64_t)w - k + 1, 0, 0, NULL, NULL};
    x.q_el = (u128_t *)malloc(x.ws * sizeof(u128_t)); x.q_sc = (uint64_t *)malloc(x.ws * sizeof(uint64_t));
    (void)rolling_hash128_core(s, l, k, canon, fwd, rc, NULL, 0, &x);
    if (x.q_n > 0 && x.q_n < x.ws) {
        uint64_t b = x.q_head;
        for (uint64_t i = 1; i < x.q_n; ++i) {
            const uint64_t j = (x.q_head + i) % x.ws;
            if (x.q_sc[j] < x.q_sc[b] || (x.q_sc[j] == x.q_sc[b] && x.q_el[j] < x.q_el[b])) b = j;
        }
        if (x.n < x.cap) { x.out[2 * x.n] = (uint64_t)x.q_el[b]; x.out[2 * x.n + 1] = (uint64_t)(x.q_el[b] >> 64); }
        ++x.n;
    }
    free(x.q_el); free(x.q_sc);
    return x.n;
}

/* RollingHasher with a window (wsz > k: qmap_ of wsz-k+1 entries, encoder.h:664-671): every hash goes through
 * qmap_.next_value(v, lex_score(v)) and what comes out is the entry with the smallest (FRev64(v), v) once the queue is full
 * (:706-710,:771-776; qmap.h:79-87).  The canonical path pushes BOTH strands' hashes, forward then reverse, as separate
 * entries (add_hashes(hasher_); add_hashes(rchasher_), :724-725,:730-731) -- it does not take their minimum first.  The
 * queue survives a restart at an invalid character, and a queue that never filled flushes its minimum (:735-736,:794-795).
 * lex_score is FRev64 (parity unpinned), on top of the unpinned tables (F10). */
typedef struct { uint64_t *out, cap, n; int canon; uint64_t ws, q_n, q_head, *q_el, *q_sc; } rh_win_t;
static void rh_win_push(rh_win_t *x, uint64_t v)
{
    if (x->q_n == x->ws) { x->q_head = (x->q_head + 1) % x->ws; --x->q_n; }
    const uint64_t at = (x->q_head + x->q_n) % x->ws;
    x->q_el[at] = v; x->q_sc[at] = frev64(v); ++x->q_n;
    if (x->q_n == x->ws) {
        uint64_t b = 0;
        for (uint64_t i = 1; i < x->ws; ++i)
            if (x->q_sc[i] < x->q_sc[b] || (x->q_sc[i] == x->q_sc[b] && x->q_el[i] < x->q_el[b])) b = i;
        if (x->q_el[b] != ~UINT64_C(0)) { if (x->n < x->cap) x->out[x->n] = x->q_el[b]; ++x->n; }
    }
}
static void rh_win_use(uint64_t h, uint64_t g, void *ud)
{
    rh_win_t *x = (rh_win_t *)ud;
    rh_win_push(x, h);
    if (x->canon) rh_win_push(x, g);
}

uint64_t bo_rolling_hash_windowed(const char *s, uint64_t l, unsigned k, int canon, unsigned w, const uint64_t *fwd,
                                  const uint64_t *rc, uint64_t *out, uint64_t cap)
{
    if (w <= k) return bo_rolling_hash(s, l, k, canon, fwd, rc, out, cap);      /* window(): w <= k_ -> no window, :664-665 */
    rh_win_t x = {out, cap, 0, canon, (uint64_t)w - k + 1, 0, 0, NULL, NULL};
    x.q_el = (uint64_t *)malloc(x.ws * sizeof(uint64_t)); x.q_sc = (uint64_t *)malloc(x.ws * sizeof(uint64_t));
    rh_core(s, l, k, canon, fwd, rc, rh_win_use, &x);
    if (x.q_n > 0 && x.q_n < x.ws) {                                            /* partially_full(): the queue's minimum */
        uint64_t b = x.q_head;
        for (uint64_t i = 1; i < x.q_n; ++i) {
            const uint64_t j = (x.q_head + i) % x.ws;
            if (x.q_sc[j] < x.q_sc[b] || (x.q_sc[j] == x.q_sc[b] && x.q_el[j] < x.q_el[b])) b = j;
        }
        if (x.n < x.cap) x.out[x.n] = x.q_el[b];
        ++x.n;
    }
    free(x.q_el); free(x.q_sc);
    return x.n;
}

/* ------------------------------------------------------------------ Encoder::for_each_hash (encoder.h:355-394)
 * PARITY UNPINNED (un-vendored ntHash): NTC64 restated from the published definition, ntHash 1.0.x:
 *   NTC64(kmerSeq, k, fhVal, rhVal):  fhVal = XOR_i rol(seedTab[s_i], k-1-i);  rhVal = XOR_i rol(seedTab[s_i & cpOff], i);
 *                                     returns min(rhVal, fhVal)                              (cpOff = 0x07)
 *   NTC64(charOut, charIn, k, fhVal, rhVal):  fhVal = rol(fhVal,1) ^ rol(seedTab[charOut],k) ^ seedTab[charIn];
 *                                     rhVal = ror(rhVal ^ rol(seedTab[charIn & cpOff],k) ^ seedTab[charOut & cpOff], 1)
 * The table geometry is in-tree: make_nthash_lut, encoder.h:93-103. */
static inline uint64_t nt_rol(uint64_t v, unsigned s) { s &= 63u; return s ? (v << s) | (v >> (64u - s)) : v; }
static inline uint64_t nt_ror(uint64_t v, unsigned s) { s &= 63u; return s ? (v >> s) | (v << (64u - s)) : v; }

void bo_nthash_tables(uint64_t a, uint64_t c, uint64_t g, uint64_t t, uint64_t *ret)
{
    memset(ret, 0, 256 * sizeof(uint64_t));
    ret[4] = ret['a'] = ret['A'] = a;            /* encoder.h:98 */
    ret[7] = ret['c'] = ret['C'] = c;            /* :99 */
    ret[3] = ret['g'] = ret['G'] = g;            /* :100 */
    ret[1] = ret['t'] = ret['T'] = t;            /* :101 */
}

static uint64_t ntc64_init(const unsigned char *kmer, unsigned k, const uint64_t *T, uint64_t *fh, uint64_t *rh)
{
    uint64_t f = 0, r = 0;
    for (unsigned i = 0; i < k; ++i) {
        f ^= nt_rol(T[kmer[i]], k - 1u - i);
        r ^= nt_rol(T[kmer[i] & 7u], i);
    }
    *fh = f; *rh = r;
    return r < f ? r : f;
}
static uint64_t ntc64_roll(unsigned char cout, unsigned char cin, unsigned k, const uint64_t *T, uint64_t *fh, uint64_t *rh)
{
    *fh = nt_rol(*fh, 1) ^ nt_rol(T[cout], k) ^ T[cin];
    *rh = nt_ror(*rh ^ nt_rol(T[cin & 7u], k) ^ T[cout & 7u], 1);
    return *rh < *fh ? *rh : *fh;
}
/* cstr_lut (kmerutil.h:36-48): 128 entries, A C G T in either case are >= 0.  DEFINED-BEHAVIOUR: bytes >= 128 index the
 * reference's table out of bounds (negative char); here they are invalid. */
static inline int nt_valid(unsigned char c) { return c < 128 && bo_dna4(c) >= 0; }

/* The loop as written (encoder.h:366-393), labels and all.  s must be readable at s[l] as the reference reads its NUL; the
 * callers here pass buffers of l bytes, so the terminator is simulated: position l reads as 0. */
uint64_t bo_for_each_hash(const char *str, uint64_t l_, unsigned k, int canon, const uint64_t *T, uint64_t *out, uint64_t cap)
{
    const unsigned char *s_ = (const unsigned char *)str;
#define CH(idx) ((uint64_t)(idx) < l_ ? s_[idx] : (unsigned char)0)
    uint64_t n = 0;
    if (l_ < k) return 0;                                                   /* :365 */
    uint64_t i = 0, fhv = 0, rhv = 0, hv;
    uint64_t p, p2;
start:
    p = i;                                                                  /* :369 */
    while (CH(p) && !nt_valid(CH(p))) ++p;                                  /* :370 */
    for (;;) {
        p2 = p;
        if (CH(p2) == 0) return n;                                          /* :373 */
        while (CH(p2) && nt_valid(CH(p2)) && p2 - p < k) ++p2;              /* :374 */
        if (CH(p2) == 0) return n;                                          /* :375 -- before the length test: a first window that
                                                                               ends at the end of the string is never emitted */
        if (p2 - p == k) break;                                             /* :376 */
        p = p2 + 1;                                                         /* :377 */
    }
    i = p;                                                                  /* :379 */
    hv = ntc64_init(s_ + i, k, T, &fhv, &rhv);                              /* :380 */
    if (n < cap) out[n] = canon ? hv : fhv;                                 /* :381 */
    ++n;
    for (; i < l_ - k; ++i) {                                               /* :382 */
        const unsigned char newc = CH(i + k);
        if (!nt_valid(newc)) {                                              /* :384 */
            i += k;
            fhv = rhv = 0;
            goto start;
        }
        hv = ntc64_roll(s_[i], newc, k, T, &fhv, &rhv);                     /* :389 */
        if (n < cap) out[n] = canon ? hv : fhv;
        ++n;
    }
    return n;
#undef CH
}
