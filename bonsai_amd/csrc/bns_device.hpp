// bns_device.hpp -- device-side building blocks of the classify hot path (gfx950 / CDNA4, wave64).
//
// Reference semantics restated per function (file:line into dnbaker/bonsai):
//   wang64            khash64.h:202-211        hash of khash_t(c) keys
//   revcomp/canonical kmerutil.h:83-90,137-140
//   probe_khash       khash64.h:250-263        kh_get on the on-disk SoA arrays
//   probe_bucket      same key->value map, re-hashed into 64-byte buckets (one HBM sector per lookup)
//   lca_dev           util.h:634-663
//   resolve (in bns_kernels.hip) util.h:831-869
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace bns {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

constexpr u32 TAX_ABSENT = 0xFFFFFFFFu;

// One slot of the bucket layout: 16 bytes, 4 slots per 64-byte bucket.
struct alignas(16) Slot {
    u64 key;
    u32 val;
    u32 occ;   // 0 = empty, 1 = occupied (key 0 is a legal key: poly-A, SURVEY 7 "Key 0 is a real key")
};

// Flattened taxonomy node: parent[] plus an Euler-tour interval, so "a is an ancestor of b" is
// tin[a] < tin[b] < tout[a] with no pointer chasing.  tin == tout == 0 for ids that are not in the forest.
struct alignas(16) TaxNode {
    u32 parent;   // TAX_ABSENT when the id is not a key of the parent map
    u32 tin, tout;
    u32 flags;    // bit0: id is a key; bit1: every node on the chain id -> root is a key (lca() cannot hit "Missing taxid")
};
constexpr u32 NODE_PRESENT = 1u, NODE_CHAIN_OK = 2u;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ u64 wang64(u64 key)
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

// Reverse the order of the 2-bit symbols and complement: bit-reverse both halves (2x v_bfrev_b32, swapped), then swap
// the two bits inside every symbol back and complement -- per 32-bit half two shifts and one three-input bit op (the
// bits a 64-bit shift would carry across the halves are masked out anyway).
// (the swap-and-complement as ONE v_bitop3_b32: D = ~((x >> 1 & C) | (x << 1 & ~C)), C = 0x5555...; truth table with
// a = 0xF0, b = 0xCC, c = 0xAA: ~((a & c) | (b & ~c)) = 0x1B -- left to itself the compiler spends an AND or two per half beside it)
__device__ __forceinline__ u32 pairswap_not(u32 x)
{
    u32 r;
    const u32 c = 0x55555555u;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x1b" : "=v"(r) : "v"(x >> 1), "v"(x << 1), "s"(c));
    return r;
}
__device__ __forceinline__ u64 revcomp(u64 kmer, u32 k)
{
    const u32 a = __builtin_bitreverse32((u32)(kmer >> 32)), b = __builtin_bitreverse32((u32)kmer);     // b:a = brev64(kmer)
    const u32 lo = pairswap_not(a), hi = pairswap_not(b);
    return (((u64)hi << 32) | lo) >> (64u - (k << 1));
}

// Same for a k-mer sitting in the TOP 2k bits of `win` (whatever is below): the bit reversal leaves the reversed k-mer in the
// low 2k bits, so a mask replaces the final shift.
__device__ __forceinline__ u64 revcomp_top(u64 win, u32 k)
{
    const u32 a = __builtin_bitreverse32((u32)(win >> 32)), b = __builtin_bitreverse32((u32)win);
    const u32 lo = pairswap_not(a), hi = pairswap_not(b);
    return (((u64)hi << 32) | lo) & (~0ULL >> (64u - (k << 1)));
}

__device__ __forceinline__ u64 canonical(u64 kmer, u32 k)
{
    const u64 rc = revcomp(kmer, k);
    return kmer < rc ? kmer : rc;
}

// ---- DPP helpers (quad = 4 adjacent lanes) ---------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ u32 dpp(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v)
{
    const u32 lo = dpp<CTRL>((u32)v), hi = dpp<CTRL>((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
constexpr int QP_XOR1 = 0xB1;   // quad_perm(1,0,3,2)
constexpr int QP_XOR2 = 0x4E;   // quad_perm(2,3,0,1)
template <int S> struct QBcast { static constexpr int ctrl = S * 0x55; };   // quad_perm(S,S,S,S)

__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ u32 readlane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    return ((u64)readlane((u32)(v >> 32), l) << 32) | readlane((u32)v, l);
}
__device__ __forceinline__ u64 lanemask_lt() { return (1ULL << lane_id()) - 1ULL; }

struct ProbeResult { u32 val; bool found; };
#ifdef BNS_COUNT_FETCHES                        // measurement builds only (tools/measure.sh)
__device__ unsigned long long g_fetch_count[8];      // buckets fetched, probe passes, lanes sent to the overflow table, rounds with such lanes, quad-probe iterations
#endif

// kh_get on the untouched khash arrays (khash64.h:250-263): triangular probing, 2-bit flags
// (bit1 empty, bit0 deleted, khash64.h:169-177), abort when the probe returns to its start.
__device__ __forceinline__ ProbeResult probe_khash(const u32 *__restrict__ flags, const u64 *__restrict__ keys,
                                                   const u32 *__restrict__ vals, u64 n_buckets, u64 key, bool active)
{
    ProbeResult r{0u, false};
    if (!active || n_buckets == 0) return r;
    const u64 mask = n_buckets - 1;
    u64 i = wang64(key) & mask, step = 0;
    const u64 last = i;
    for (;;) {
        const u32 f = (flags[i >> 4] >> ((i & 0xfu) << 1)) & 3u;
        if (f & 2u) return r;                                  // empty slot ends the search: miss
        if (!(f & 1u) && keys[i] == key) { r.val = vals[i]; r.found = true; return r; }
        i = (i + (++step)) & mask;
        if (i == last) return r;
    }
}

// Bucket layout probe: the 4 lanes of a quad fetch one 64-byte bucket together (lane s of the quad
// reads slot s: one fully coalesced 64-byte request per lookup), four lookups per quad per pass.
// Must be called from wave-uniform control flow (uses DPP); `active` may differ per lane.
template <bool LINEAR>
__device__ __forceinline__ ProbeResult probe_bucket_from(const Slot *__restrict__ slots, u64 bucket_mask, u64 key, u64 b, bool active)
{
    ProbeResult r{0u, false};
    const int sub = lane_id() & 3;
    u64 step = 0;
    u32 pending = active ? 1u : 0u;
    while (ballot64(pending != 0)) {
#ifdef BNS_COUNT_FETCHES
        if (lane_id() == 0) atomicAdd(&g_fetch_count[4], 1ULL);
#endif
        uint4 s0, s1, s2, s3;
        const u64 b0 = dpp64<QBcast<0>::ctrl>(b), b1 = dpp64<QBcast<1>::ctrl>(b);
        const u64 b2 = dpp64<QBcast<2>::ctrl>(b), b3 = dpp64<QBcast<3>::ctrl>(b);
        const u32 p0 = dpp<QBcast<0>::ctrl>(pending), p1 = dpp<QBcast<1>::ctrl>(pending);
        const u32 p2 = dpp<QBcast<2>::ctrl>(pending), p3 = dpp<QBcast<3>::ctrl>(pending);
        const uint4 zero = make_uint4(0, 0, 0, 0);
        const uint4 *base = reinterpret_cast<const uint4 *>(slots);
        s0 = p0 ? base[b0 * 4 + sub] : zero;
        s1 = p1 ? base[b1 * 4 + sub] : zero;
        s2 = p2 ? base[b2 * 4 + sub] : zero;
        s3 = p3 ? base[b3 * 4 + sub] : zero;
#define BNS_PROBE_STEP(S, sl, pS)                                                                  \
        {                                                                                          \
            const u64 kk = dpp64<QBcast<S>::ctrl>(key);                                            \
            const u64 skey = ((u64)sl.y << 32) | sl.x;                                             \
            const u32 occ = (pS && sl.w) ? 1u : 0u;                                                \
            const u32 match = (occ && skey == kk) ? 1u : 0u;                                       \
            u32 mv = match ? sl.z : 0u;                                                            \
            u32 fl = match | (occ << 1);      /* bit0: any match, bit1: all occupied */            \
            mv |= dpp<QP_XOR1>(mv); mv |= dpp<QP_XOR2>(mv);                                        \
            { const u32 o1 = dpp<QP_XOR1>(fl); fl = ((fl | o1) & 1u) | (fl & o1 & 2u); }           \
            { const u32 o2 = dpp<QP_XOR2>(fl); fl = ((fl | o2) & 1u) | (fl & o2 & 2u); }           \
            if (sub == S && pending) {                                                             \
                if (fl & 1u) { r.found = true; r.val = mv; pending = 0; }                          \
                else if (!(fl & 2u)) pending = 0;            /* bucket has a free slot: miss */    \
                else b = (b + (LINEAR ? 1 : (++step))) & bucket_mask;   /* bucket full: next bucket */ \
            }                                                                                      \
        }
        BNS_PROBE_STEP(0, s0, p0)
        BNS_PROBE_STEP(1, s1, p1)
        BNS_PROBE_STEP(2, s2, p2)
        BNS_PROBE_STEP(3, s3, p3)
#undef BNS_PROBE_STEP
    }
    return r;
}

__device__ __forceinline__ ProbeResult probe_bucket(const Slot *__restrict__ slots, u64 bucket_mask, u64 key, bool active)
{
    return probe_bucket_from<false>(slots, bucket_mask, key, wang64(key) & bucket_mask, active);
}
// Bucket of a key in the MINBUCKET layout's overflow table: the HIGH half of the Wang hash.  The low bits are the key's position
// in the khash arrays, and the fill kernel walks those arrays with a power-of-two stride -- the keys ONE thread handles share
// their low hash bits, and which keys end up in the overflow table depends on when their thread ran: hashed by the low bits,
// overflow keys pile into the buckets of the late threads (measured: 29 % of the buckets full at 40 % load, 160 probe steps per
// lookup, the kernel 8x slower; with the high bits 8 %, as a uniform hash gives).
__device__ __forceinline__ u64 ovf_bucket(u64 key, u64 ovf_mask) { return (wang64(key) >> 32) & ovf_mask; }

// ---- minimizer-clustered bucket layout (BNS_LAYOUT_MINBUCKET) --------------------------------------------
// The home bucket of a key is chosen by the smallest hash among the canonical m-mers INSIDE the key -- a pure
// function of the key -- and a full bucket spills to the NEXT bucket.  Consecutive k-mers of a read share their
// minimizer for ~(k-m+2)/2 positions, so their lookups land in the same 128-byte bucket: one DRAM fetch serves
// several lookups instead of one.  The key->value map is unchanged.
//   m = k for k <= 19 (no clustering: 4^m must dwarf the db or groups outgrow buckets), else k - 8, k - 11 or k - 15 (minimizer_len below);
//   m = k for spaced seeds too (consecutive spaced keys share no m-mers, so clustering buys nothing).
//   bucket = 128 B: u64 keys[10] | u32 vals[10] | u32 n (header word, bits below) | u32 pad (the bucket's perfect-hash
//   multiplier, see mph_slot below).  A minimizer group has up to k-m+1 keys (9 / 12 / 16 for the windows 8 / 11 / 15), a
//   bucket holds 10: a window whose groups outgrow their buckets is rejected when the table is loaded (spill count).
struct alignas(16) MinBucket {
    u64 keys[10];
    u32 vals[10];
    u32 n;
    u32 pad;
};
static_assert(sizeof(MinBucket) == 128, "MinBucket must be one 128-byte line");
constexpr u32 MINB_CAP = 10;
// Inside a bucket a key sits at slot mph_slot(mph_fold(key), S), S being a per-bucket odd multiplier found when the table
// is loaded such that the bucket's keys land on distinct slots (a minimal perfect hash of <= 10 keys: about 2755 candidates
// for a full bucket, tried 64 at a time).  The header word `n` holds count | occupancy << 8, `pad` holds S.  A bucket without
// such an S -- two keys with the same fold, about one bucket in 10^8 -- has its keys moved to the overflow table and its
// count set to MINB_N_IN_OVF, which reads as "full, no hit, go on" (and "look in the overflow table if the walk finds
// nothing").  Unused slots hold ~0.
constexpr u32 MINB_N_IN_OVF = 0xFFu;
// Header word `n`:  bits 0-7 count (MINB_N_IN_OVF: keys moved to the overflow table) | bits 8-17 slot occupancy |
// bits 18-21 "where do keys whose HOME bucket this is live": bit 18+d (d = 0..2) = some such key sits d+1 or more buckets down
// the chain, bit 21 = some such key lives in the overflow table (its chain of MINB_MAX_CHAIN buckets was full when the table was
// filled; setting it sets bits 18-20 too).  A lookup that misses in the c-th bucket of its walk goes on only when its HOME
// bucket's bit 18+c is set: a miss in a full bucket that never spilled anything ends there (most lookups of a read are misses
// on a db of window minimizers, and at high table loads most buckets are full), and the overflow table -- the slowest thing a
// lane can do -- is consulted only on the say-so of the home bucket.
constexpr u32 MINB_HOME_SHIFT = 18u;
constexpr u32 MINB_HOME_OVF = 1u << 21;
constexpr u32 MINB_HOME_MASK = 0xFu << MINB_HOME_SHIFT;
// bits 22-29 "WHICH of the minimizer groups whose home this is have keys elsewhere": one bit per group tag (three bits of a second
// hash of the minimizer value, minb_tagbit below), set when a key with that tag is placed outside this (its home) bucket.  A
// lookup that misses here goes on only when its own tag's bit is set -- a group that lives here whole, and any k-mer that is not
// in the db at all (a read with a sequencing error: a minimizer nobody put here), ends at its first fetch even when the bucket is
// full and other groups spilled.  The group-aware fill (minbucket_tagcount / _decide / _fill_kernel) keeps whole groups at home,
// largest first, so that the groups that do spill are few.
constexpr u32 MINB_TAG_SHIFT = 22u;
constexpr u32 MINB_TAG_MASK = 0xFFu << MINB_TAG_SHIFT;
__device__ __forceinline__ u32 minb_tag(u32 minh) { return (minh * 0x85EBCA6Bu) >> 29; }          // (bucket_of mixes with another multiplier)
__device__ __forceinline__ u32 minb_tagbit(u32 minh) { return (1u << MINB_TAG_SHIFT) << minb_tag(minh); }
__device__ __forceinline__ u32 mph_fold(u64 key) { return (u32)key ^ __builtin_rotateleft32((u32)(key >> 32), 15); }
__device__ __forceinline__ u32 mph_slot(u32 x, u32 S) { return __umulhi(x * S, MINB_CAP); }
__device__ __forceinline__ u32 mph_candidate(u64 bucket, u32 t)
{
    u32 z = (u32)bucket * 0x9E3779B1u + t * 0x85EBCA6Bu + (u32)(bucket >> 32);
    z ^= z >> 15; z *= 0x2C1B3C6Du; z ^= z >> 12;
    return z | 1u;
}

// Minimizer length m (contiguous seeds): the window of a k-mer is its k - m + 1 m-mers.  A wider window means fewer minimizer runs
// per read (density 2 / (k - m + 2): 25 bucket fetches per 150-bp read at k - m = 8, 16 at 15) but groups of up to k - m + 1
// keys -- of which a bucket holds 10.  Which one a table uses is decided when it is loaded (bns_load_table_device): a db of every
// k-mer fills its groups and needs the narrow window; a db of window minimizers (bonsai build -w 50: one k-mer in ten) leaves
// them nearly empty and takes the wide one.  m never goes below `floor`: 4^m must dwarf the number of minimizer groups.  The
// wide candidate is k - 15 rather than k - 14 because for k = 31 that is m = 16: an m-mer that fits ONE word, whose canonical
// form is a v_min_u32 (round_minhash) -- 2 % faster than m = 17, while m = 15 (k - 16) puts unrelated k-mers into one group.
struct MinCand { u32 span, floor; };
constexpr MinCand MIN_CANDS[3] = {{15u, 16u}, {11u, 19u}, {8u, 19u}};        // widest first; the last one always fits (groups <= 9)
constexpr int BNS_MAX_SPAN = 15;                                              // round_minhash unrolls windows of up to this + 1
__device__ __host__ constexpr u32 minimizer_len(u32 k, MinCand c) { return k <= c.floor ? k : (k - c.span > c.floor ? k - c.span : c.floor); }
__device__ __host__ constexpr u32 minimizer_len(u32 k) { return minimizer_len(k, MIN_CANDS[2]); }     // the narrow window
// 32-bit mix of a folded m-mer: ONE multiply.  Only the ORDER of the values matters here (the smallest wins, and minhash_bucket
// re-mixes the winner before it is masked), and the order is decided by the product's high bits, which every input bit
// reaches.  (The murmur3 finaliser shape this replaced -- xorshift, multiply, xorshift -- cost four more VALU instructions per
// hash and bought nothing: same spill counts, kernel 1.3 % slower.)
__device__ __forceinline__ u32 mmer_mix(u32 h) { return h * 0x7FEB352Du; }
__device__ __forceinline__ u32 mmer_hash(u64 x)                 // 32-bit hash of a <= 64-bit m-mer
{
    return mmer_mix((u32)x ^ __builtin_rotateleft32((u32)(x >> 32), 13));
}
__device__ __forceinline__ u64 canon_mmer(u64 fw, u32 m)
{
    const u64 r = revcomp(fw, m);
    return fw < r ? fw : r;
}
// Which part of a key the minimizer is taken from: m-mers of `len` consecutive key bases whose last base sits `shift` bits above
// the key's low end.  Contiguous seeds: the whole key (len = k, shift = 0), canonical m-mers -- the m-mer sets of a k-mer and of
// its reverse complement coincide.  Spaced seeds: the k sampled bases are not consecutive in the read except inside a run of
// adjacent sampled positions; the LONGEST such run (16 bases for 1x15,0x15) is the only thing neighbouring spaced k-mers share
// position by position (k-mers j and j+1 overlap in len-1 of its bases), so the minimizer is taken inside it, over plain m-mers
// (spaced k-mers are never canonicalised, encoder.h:148-150).  len == m == k: no clustering, the one m-mer.
struct MinSpec { u32 m, len, shift, canon, wide; };
// 32-bit minimizer value -> bucket in [0, n_mb): multiply, xorshift (a minimum of hashes is biased towards small values: re-mix
// first; the product's top bits alone spread the groups worse -- 50 % more keys outside their home bucket), then a
// multiply-high range reduction of the bit-reversed word, so that the bucket count need not be a power of two (a table can take
// exactly the memory there is: 8e9 keys in 230 GB instead of a choice between 137 and 275).  For n_mb = 2^b this is the
// reversed low b bits of the mixed word -- the masked index of rounds 1-2, permuted.
__device__ __forceinline__ u32 bucket_of(u32 minh, u32 n_mb)
{
    u32 x = minh * 0x9E3779B1u; x ^= x >> 15;
    return __umulhi(__builtin_bitreverse32(x), n_mb);
}
// WIDE minimizer identity (MinSpec::wide; contiguous seeds).  The narrow form orders the m-mers of a window by a 32-bit hash and
// derives the bucket from the winning HASH VALUE: a minimum over w draws concentrates near 0 (effective space 2^32 / (w/2)), so
// beyond a few 1e8 minimizer groups distinct groups share hash values -- and buckets -- whatever the table size (4e9 keys: nearly
// every bucket shared).  The wide form carries 52 bits through the window minimum: a positive double with a fixed exponent
// whose mantissa is  hash20 : m-mer  (m <= 16: the canonical m-mer itself rides along, the bucket comes from IT, no collisions
// at all)  or the top 52 bits of a 64-bit product (longer m-mers).  Doubles of one exponent order like their bit patterns, so
// the window minimum is one v_min_f64 per entry instead of a 64-bit compare and two selects.
__device__ __forceinline__ u64 wide_ident32(u32 x) { return ((u64)__builtin_amdgcn_alignbit(0x3FFu, x * 0x7FEB352Du, 12) << 32) | x; }
__device__ __forceinline__ u64 wide_ident64(u64 x) { return 0x3FF0000000000000ULL | ((x * 0xD6E8FEB86659FD93ULL) >> 12); }
__device__ __forceinline__ u64 wide_ident(u64 x, u32 m) { return m <= 16u ? wide_ident32((u32)x) : wide_ident64(x); }
__device__ __forceinline__ u32 wide_bucket_in(u64 ident, u32 m)
{
    return m <= 16u ? (u32)ident : (u32)ident ^ (((u32)(ident >> 32) & 0xFFFFFu) * 0x85EBCA6Bu);
}
__device__ __forceinline__ u64 wide_min(u64 a, u64 b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, a)), "v"(__builtin_bit_cast(double, b)));
    return __builtin_bit_cast(u64, r);
}
// generic form: the minimizer value of a key (what bucket_of takes), from the 2k-bit key alone
__device__ __forceinline__ u32 key_minhash(u64 key, u32 k, MinSpec sp)
{
    const u32 m = sp.m;
    const u64 mmask = ~0ULL >> (64u - 2u * m);
    if (sp.wide) {                                                   // (contiguous seeds, whole key, canonical m-mers)
        const u64 rc = revcomp(key, k);
        u64 best = ~0ULL;
        for (u32 i = 0; i + m <= k; ++i) {
            const u64 a = (key >> (2u * (k - m - i))) & mmask, b = (rc >> (2u * i)) & mmask;
            const u64 id = wide_ident(a < b ? a : b, m);
            best = id < best ? id : best;
        }
        return wide_bucket_in(best, m);
    }
    if (m == k) return mmer_hash(canon_mmer(key & mmask, m));     // no clustering (k <= 19, masks without a long run): the one m-mer, no loop
    u32 best = 0xFFFFFFFFu;
    if (!sp.canon && sp.shift + 2u * sp.len <= 32u) {               // (wave-uniform) the region lies in the key's low word: 32-bit
        const u32 r32 = (u32)key >> sp.shift, mm = 0xFFFFFFFFu >> (32u - 2u * m);   // arithmetic, same values (mmer_hash of a
        for (u32 i = 0; i + m <= sp.len; ++i) {                                     // value below 2^32 mixes its low word only)
            const u32 h = mmer_mix((r32 >> (2u * (sp.len - m - i))) & mm);
            best = h < best ? h : best;
        }
        return best;
    }
    const u64 region = key >> sp.shift;
    for (u32 i = 0; i + m <= sp.len; ++i) {
        const u64 x = (region >> (2u * (sp.len - m - i))) & mmask;
        const u32 h = mmer_hash(sp.canon ? canon_mmer(x, m) : x);
        best = h < best ? h : best;
    }
    return best;
}
// contiguous seeds (whole key, canonical m-mers): the loop with nothing but k and m in it.  The standalone probe kernel is
// instantiated with one form or the other: with both inlined behind a run-time test it needed 84 SGPRs instead of 70, which is
// 7 waves per SIMD instead of 8 and cost it 16 %.
template <bool WIDE = false>
__device__ __forceinline__ u32 key_minhash(u64 key, u32 k, u32 m)
{
    const u64 mmask = ~0ULL >> (64u - 2u * m);
    // the reverse complement of the key's i-th m-mer is the (k - m - i)-th m-mer of the key's reverse complement: ONE bit reversal
    // for the whole window instead of one per m-mer
    const u64 rc = revcomp(key, k);
    if (WIDE) {
        u64 best = ~0ULL;
        if (m <= 16u) {
            const u32 mm = 0xFFFFFFFFu >> (32u - 2u * m);
            for (u32 i = 0; i + m <= k; ++i) {
                const u64 id = wide_ident32(min((u32)(key >> (2u * (k - m - i))) & mm, (u32)(rc >> (2u * i)) & mm));
                best = id < best ? id : best;
            }
        } else {
            for (u32 i = 0; i + m <= k; ++i) {
                const u64 a = (key >> (2u * (k - m - i))) & mmask, b = (rc >> (2u * i)) & mmask;
                const u64 id = wide_ident64(a < b ? a : b);
                best = id < best ? id : best;
            }
        }
        return wide_bucket_in(best, m);
    }
    if (m == k) return mmer_hash(canon_mmer(key & mmask, m));
    u32 best = 0xFFFFFFFFu;
    if (m <= 16u) {                                                  // m-mers that fit a word (as round_minhash)
        const u32 mm = 0xFFFFFFFFu >> (32u - 2u * m);
        for (u32 i = 0; i + m <= k; ++i) {
            const u32 h = mmer_mix(min((u32)(key >> (2u * (k - m - i))) & mm, (u32)(rc >> (2u * i)) & mm));
            best = h < best ? h : best;
        }
        return best;
    }
    for (u32 i = 0; i + m <= k; ++i) {
        const u64 a = (key >> (2u * (k - m - i))) & mmask, b = (rc >> (2u * i)) & mmask;
        const u32 h = mmer_hash(a < b ? a : b);
        best = h < best ? h : best;
    }
    return best;
}

// Probe: wave-cooperative.  Lanes whose neighbour wants the same bucket share ONE fetch: run leaders are ranked
// with a ballot, up to 16 distinct buckets per pass are fetched by two fully coalesced 1 KiB loads (lane l reads 16-byte
// chunk l&7 of bucket l>>3) and staged in LDS; every lane then looks at the ONE slot its key can occupy in its bucket
// (a per-bucket perfect hash, see mph_slot): header, key, value -- three LDS reads, one compare.
// Runs ranked 16 and above simply stay pending for the next pass.
// aux = per-wave LDS (u32 units): [0,64) bucket list, [64, 64 + 16*32) stage (16-byte aligned).
constexpr u32 MINB_MAX_CHAIN = 4;              // a key lives in one of its first 4 buckets (40 keys) or in the overflow table
constexpr int MINB_STRIDE = 8;                  // uint4 per staged bucket (the loads write LDS directly, lane-linear: no padding possible)
constexpr int MINB_LIST_U32 = 64;               // bucket list in front of the stage
constexpr int MINB_AUX_U32 = MINB_LIST_U32 + 16 * MINB_STRIDE * 4;
// aux size for a stage of NB buckets per pass (NB = 16: contiguous seeds, whose 64 lookups touch ~13 buckets; 32 / 64: spaced
// seeds, whose 64 lookups touch 64 -- fewer passes, more fetches in flight per wavefront)
constexpr int minb_aux_u32(int nb) { return MINB_LIST_U32 + nb * MINB_STRIDE * 4; }
constexpr int DPP_WAVE_SHR1 = 0x138;            // lane i <- lane i-1 across the whole wavefront (gfx9 DPP)
constexpr u32 MINB_NONE = 0xFFFFFFFFu;          // "no bucket wanted" (bucket indices are < 2^31)
// Oversized minimizer groups (conserved sequence shared by many genomes) would make spill chains arbitrarily long, so
// a chain is capped at MINB_MAX_CHAIN buckets: keys that find them all full go to a small plain-hashed overflow table
// (64-byte buckets of 4 slots), and a lookup that walks MINB_MAX_CHAIN full buckets without a hit continues there.
// NB = buckets staged per pass (16 for contiguous seeds; the spaced instantiations use a wider stage).
// PEEL: the first pass as code of its own (classify_kernel: its lanes are all at home then, and instructions are what it is short
// of); the standalone probe kernel is short of registers instead and runs every pass through the general form.
// TAGS: the caller hands in tbit = minb_tagbit(minimizer value) and a lane leaves its home bucket only when that bit is set in the
// bucket's header (the crowded-table instantiations; without it every miss in a full home that spilled anything walks on).
template <bool KEY_MAY_BE_ONES = true, int NB = 16, bool OVF_COOP = false, bool PEEL = true, bool TAGS = false>
__device__ __forceinline__ ProbeResult probe_minbucket(const MinBucket *__restrict__ buckets, u64 key, u32 b, bool active, u32 *aux,
                                                       const Slot *__restrict__ ovf_slots, u64 ovf_mask, u32 tbit = 0u)
{
    // Per-lane state is kept as integers in VGPRs and updated with selects: `bool`s updated under divergent control
    // flow live in SGPR lane masks and cost three scalar mask merges per variable per join.  A lane is pending while
    // bkt != MINB_NONE.
    const int lane = lane_id();
    u32 *list = aux;
    uint4 *stage = reinterpret_cast<uint4 *>(aux + MINB_LIST_U32);
    const uint4 *base = reinterpret_cast<const uint4 *>(buckets);
    u32 bkt = active ? b : MINB_NONE;
    // found: bit 0 = hit, bit 1 = "look in the overflow table".  home: 0 while the lane has not left its HOME bucket; from its
    // first step down the chain on, that bucket's header bits 18-21 under a marker bit (22), shifted right once per bucket
    // walked -- bit 18 is always "go on from here", and nothing above bit 19 means the lane stands in the last bucket of its chain.
    u32 found = 0u, val = 0u, home = 0u;
    const u32 xfold = mph_fold(key);
    // One pass: fetch the buckets of up to NB run leaders, look every pending lane's key up in its bucket.  FIRST (the first pass
    // of a call): every pending lane stands at its home bucket, so "full, no hit, and a key of this home lives further down" is ONE
    // masked compare of the header word it has just read; later passes mix lanes at home (runs ranked beyond NB) with lanes down
    // their chain, whose verdict comes from the saved word.  Returns false when no lane was pending.
    bool more = true;
    auto pass = [&](auto first_tag) -> bool {
        constexpr bool FIRST = decltype(first_tag)::value;
        // run leader = pending lane whose left neighbour wants another bucket (lane 0 sees ~bkt, which always differs)
        const u32 prev = (u32)__builtin_amdgcn_update_dpp((int)~bkt, (int)bkt, DPP_WAVE_SHR1, 0xf, 0xf, false);
        const bool pend = bkt != MINB_NONE, chg = bkt != prev, leader = pend & chg;
        const u64 lead = ballot64(pend) & ballot64(chg);                       // (two plain compare masks and'ed in SALU)
        if (!lead) return false;                                               // every pending lane has a leader at or before it
        const int n_lead = __popcll(lead);
#ifdef BNS_COUNT_FETCHES
        if (lane == 0) { atomicAdd(&g_fetch_count[0], (unsigned long long)(n_lead < NB ? n_lead : NB)); atomicAdd(&g_fetch_count[1], 1ULL); }
#endif
        // rank of my run's leader = popc(lead & lanes <= me) - 1, as two v_mbcnt over lead >> 1
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(lead >> 33), __builtin_amdgcn_mbcnt_lo((u32)(lead >> 1), (u32)(lead & 1ULL) - 1u));
        if (leader) list[rank] = bkt;
        __builtin_amdgcn_wave_barrier();
        {
            // no predication: slots past the last leader re-read the last bucket (same lines, no extra HBM traffic)
            const u32 last = (u32)n_lead - 1u, slot = (u32)lane >> 3;
            typedef const void __attribute__((address_space(1))) *gptr_t;
            typedef void __attribute__((address_space(3))) *lptr_t;
            if (NB > 16) {
                // wide stage: NB / 8 loads of 8 buckets each, the later ones only when there are leaders for them (wave-uniform)
#pragma unroll
                for (int h = 0; h < NB / 8; ++h) {
                    if (h == 0 || (u32)(8 * h) <= last) {
                        const u32 sl = slot + 8u * (u32)h;
                        const u32 bh = list[sl < last ? sl : last];
                        __builtin_amdgcn_global_load_lds((gptr_t)(base + ((u64)bh * 8 + (u64)(lane & 7))), (lptr_t)(stage + 64 * h), 16, 0, 2);
                    }
                }
            } else {
                const u32 b0 = list[slot < last ? slot : last];
                // global_load_lds_dwordx4: lane i's 16 bytes go straight to stage + 16 i (bucket l>>3, chunk l&7) -- no VGPRs
                // in flight, no ds_write; `nt`: a bucket line is not touched again, keep it from displacing the reads and the taxonomy in L2
                // (structured buffer addressing -- buffer_load ... idxen with stride 128, which would form base + 128 * bucket in the
                // memory unit and drop five 64-bit VALU instructions per pass -- cannot be used: on gfx950 it reaches only the first
                // 4 GiB behind the base whatever NUM_RECORDS says, tools/micro/bufaddr.hip)
                __builtin_amdgcn_global_load_lds((gptr_t)(base + ((u64)b0 * 8 + (u64)(lane & 7))), (lptr_t)stage, 16, 0, 2);
                if (last >= 8u) {                  // (wave-uniform) the second load only when there are leaders for it: with the wide minimizer
                                                   // window a round has 8 leaders on average, and a continuation pass has one or two
                    const u32 b1 = list[slot + 8u < last ? slot + 8u : last];
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + ((u64)b1 * 8 + (u64)(lane & 7))), (lptr_t)(stage + 64), 16, 0, 2);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the compiler's own LDS-DMA tracking missed it in one instantiation)
        }
        __builtin_amdgcn_wave_barrier();
        const bool mine = bkt != MINB_NONE && rank < (u32)NB;
        const char *B = reinterpret_cast<const char *>(stage) + (mine ? rank : 0u) * (16u * MINB_STRIDE);
        const uint2 hdr = *reinterpret_cast<const uint2 *>(B + 120);             // {count | occupancy << 8 | home bits << 18, S}
        const u32 slot = mph_slot(xfold, hdr.y);                               // the one slot the key can be in
        const bool eq = *reinterpret_cast<const u64 *>(B + 8u * slot) == key;
        // (the occupancy bit guards a ~0 key against the padding; keys of k <= 31 never look like it)
        const bool hit = mine & eq & (!KEY_MAY_BE_ONES || ((hdr.x >> (8u + slot)) & 1u));
        const u32 v = *reinterpret_cast<const u32 *>(B + 80 + 4u * slot);
        found = hit ? 1u : found;
        val = hit ? v : val;
        // A miss goes on to the next bucket of the chain only when the bucket is full (a bucket a key spilled past is full; one
        // whose keys were moved to the overflow table reads MINB_N_IN_OVF, full as well) AND the lane's HOME bucket says that one
        // of its keys lives that far down (header bits 18+c); after the last bucket of the chain that is bit 21, "in the overflow table".
        constexpr u32 GO = 1u << MINB_HOME_SHIFT;
        bool cont;
        // (TAGS: a tag bit is set only when a key of that group was placed outside this bucket -- which was full then, and stays so)
        if (FIRST) cont = TAGS ? (mine & !hit & ((hdr.x & tbit) != 0u)) : (mine & !hit & ((hdr.x & (GO | 0xFFu)) >= (GO | MINB_CAP)));
        else if (TAGS) cont = mine & !hit & (home ? (((hdr.x & 0xFFu) >= MINB_CAP) & ((home & GO) != 0u)) : ((hdr.x & tbit) != 0u));
        else       cont = mine & !hit & ((hdr.x & 0xFFu) >= MINB_CAP) & ((((home ? home : hdr.x) & GO)) != 0u);
        bkt = (mine && !cont) ? MINB_NONE : bkt;                               // resolved lanes leave
        const u64 cont_m = ballot64(cont);
        // is there anything left for another pass?  Known here in scalar terms -- runs ranked beyond the stage, or lanes that walk on --
        // so the usual round (one pass, nobody left) does not pay a DPP shift, two compares and a ballot to find that out
        more = n_lead > NB || cont_m != 0ULL;
        if (cont_m) {                                                          // uncommon: walk on to the next bucket of the chain
            const u32 cur = (FIRST || home == 0u) ? ((hdr.x & MINB_HOME_MASK) | (GO << 4)) : home;   // (marker above the four bits)
            const bool exhausted = cont && (cur >> (MINB_HOME_SHIFT + 2u)) == 0u;   // last bucket of the chain (and bit 21 was set): the overflow table
            // (a bucket whose keys were moved to the overflow table -- no perfect hash -- reads MINB_N_IN_OVF; the homes of those keys
            // carry all four bits, so a lookup of one of them is still walking when it gets here: it need look no further)
            found |= (exhausted || (cont && (hdr.x & 0xFFu) == MINB_N_IN_OVF)) ? 2u : 0u;
            // (no wrap-around: the table has MINB_MAX_CHAIN - 1 buckets behind the last one a key can call home)
            const u32 next = exhausted ? MINB_NONE : bkt + 1u;
            bkt = cont ? next : bkt;
            home = cont ? cur >> 1 : home;
        }
        __builtin_amdgcn_wave_barrier();
        return true;
    };
    if (!PEEL || pass(std::true_type{}))
        while (more && pass(std::false_type{})) {}
    // Lanes whose chain was exhausted look their key up in the overflow table -- which IS a plain bucket table (64-byte buckets of 4
    // slots, triangular spill).  Two forms, chosen per table by the host (ClassifyParams comes with the instantiation): OVF_COOP,
    // for tables with more than 1 key in 1000 there (a table filled to a third or more): all such lanes together, quad-cooperatively
    // (probe_bucket) -- 14 % / 11 % faster on such tables, but its sixteen staging registers cost the kernel scratch and 1-5 % on
    // tables that hardly ever get here; for those, the scalar walk below.
#ifdef BNS_COUNT_FETCHES
    { const u64 g2 = ballot64(found == 2u); if (g2 && lane == 0) { atomicAdd(&g_fetch_count[2], (unsigned long long)__popcll(g2)); atomicAdd(&g_fetch_count[3], 1ULL); } }
#endif
    if (OVF_COOP) {
        // The lanes that must look there (a few per round even on a crowded table) are compacted by rank, and QUADS of lanes do
        // the lookups: lane 4 q + s reads slot s of lookup q's bucket -- one 16-byte load per lane and sixteen lookups per step,
        // against four loads per lane (sixteen staging registers: 32-40 bytes of scratch at 8 waves) in the all-lanes form.
        // Keys, buckets, step counts and results travel through the wave's LDS lines (free here: every fetch has landed).
        bool go = found == 2u;
        u64 todo = ballot64(go);
        if (todo) {
            u32 *lb = aux, *lstep = aux + 16, *lres = aux + 32, *lflag = aux + 48;
            u64 *lkey = reinterpret_cast<u64 *>(aux + MINB_LIST_U32);
            const uint4 *ob = reinterpret_cast<const uint4 *>(ovf_slots);
            u32 l3 = (u32)lane;
            asm volatile("" : "+v"(l3));                         // (quad and slot are formed here, not carried in registers across the hot loop)
            const u32 q = (l3 >> 2) & 15u, sub = l3 & 3u;
            do {
                const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(todo >> 32), __builtin_amdgcn_mbcnt_lo((u32)todo, 0u));
                const bool mine = go && rank < 16u;
                const u32 n = (u32)__popcll(todo);
                if (mine) { lkey[rank] = key; lb[rank] = (u32)ovf_bucket(key, ovf_mask); lstep[rank] = 0u; lflag[rank] = 0u; }
                __builtin_amdgcn_wave_barrier();
                bool walking = q < n;
                while (ballot64(walking)) {
                    u32 b = 0;
                    uint4 sl = make_uint4(0u, 0u, 0u, 0u);
                    if (walking) { b = lb[q]; sl = ob[(u64)b * 4u + sub]; }
                    const bool occ = sl.w != 0u;
                    const bool match = walking && occ && ((((u64)sl.y << 32) | sl.x) == lkey[q]);
                    u32 mv = match ? sl.z : 0u, fl = (match ? 1u : 0u) | (occ ? 2u : 0u);      // bit 0: any match, bit 1: all four occupied
                    mv |= dpp<QP_XOR1>(mv); mv |= dpp<QP_XOR2>(mv);
                    { const u32 o1 = dpp<QP_XOR1>(fl); fl = ((fl | o1) & 1u) | (fl & o1 & 2u); }
                    { const u32 o2 = dpp<QP_XOR2>(fl); fl = ((fl | o2) & 1u) | (fl & o2 & 2u); }
                    if (walking && sub == 0u) {
                        if (fl & 1u) { lres[q] = mv; lflag[q] = 1u; }
                        else if (fl & 2u) { const u32 st = lstep[q] + 1u; lstep[q] = st; lb[q] = (u32)(((u64)b + st) & ovf_mask); }   // full, no hit: triangular step
                    }
                    walking = walking && fl == 2u;
                    __builtin_amdgcn_wave_barrier();
                }
                if (mine && lflag[rank]) { found = 1u; val = lres[rank]; }
                go = go && !mine;
                todo = ballot64(go);
                __builtin_amdgcn_wave_barrier();
            } while (todo);
        }
        return ProbeResult{val, (found & 1u) != 0u};
    }
    // Rare: lanes whose chain was exhausted look their key up in the overflow table, one lane at a time with wave-uniform
    // (scalar) control flow -- a divergent per-lane walk here costs the hot loop ~20 SGPRs of lane masks.
    u64 todo = ballot64(found == 2u);
    while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const u64 skey = readlane64(key, l);
        const uint4 *ob = reinterpret_cast<const uint4 *>(ovf_slots);
        u64 b2 = ovf_bucket(skey, ovf_mask), step = 0;
        bool hit = false, open = false;
        u32 hv = 0;
        while (!hit && !open) {
            for (int s = 0; s < 4 && !hit && !open; ++s) {
                const uint4 sl = ob[b2 * 4 + (u64)s];
                if (!sl.w) open = true;
                else if ((((u64)sl.y << 32) | sl.x) == skey) { hit = true; hv = sl.z; }
            }
            b2 = (b2 + (++step)) & ovf_mask;
        }
        if (hit && lane == l) { found = 1u; val = hv; }
    }
    ProbeResult r{val, (found & 1u) != 0u};
    return r;
}

// ---- taxonomy ----------------------------------------------------------------------------------------
__device__ __forceinline__ TaxNode load_node(const TaxNode *__restrict__ nodes, u32 n_nodes, u32 id)
{
    if (id < n_nodes) return nodes[id];
    return TaxNode{TAX_ABSENT, 0u, 0u, 0u};
}

// is x an ancestor-or-self of a, for ids in the forest (self handled by equality)
__device__ __forceinline__ bool anc_or_self(u32 x, const TaxNode &nx, u32 a, const TaxNode &na)
{
    return x == a || (nx.tin < na.tin && na.tin < nx.tout);
}

// util.h:634-663.  `nodes` (the chain of a) is tested through the Euler interval instead of a stored
// set; the "Missing taxid" exits are reproduced through NODE_CHAIN_OK / NODE_PRESENT.
__device__ inline u32 lca_dev(const TaxNode *__restrict__ nodes, u32 n_nodes, u32 a, u32 b)
{
    if (a == b) return a;
    if (b == 0) return a;
    if (a == 0) return b;
    const TaxNode na = load_node(nodes, n_nodes, a);
    if (!(na.flags & NODE_CHAIN_OK)) return 0xFFFFFFFFu;      // some node on a's chain is not a key
    while (b) {
        const TaxNode nb = load_node(nodes, n_nodes, b);
        if (anc_or_self(b, nb, a, na)) return b;
        if (!(nb.flags & NODE_PRESENT)) return 0xFFFFFFFFu;
        b = nb.parent;
    }
    return 1;
}

}  // namespace bns
