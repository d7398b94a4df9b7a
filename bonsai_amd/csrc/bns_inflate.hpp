// bns_inflate.hpp -- raw DEFLATE (RFC 1951) decoder for BGZF members, ONE MEMBER PER LANE.
//
// What it replaces: the reference reads its input through zlib's gzFile (kseq_declare.h:112-145, klib/kseq.h:177-225: ks_getuntil
// over gzread) -- one inflate stream on one core.  A BGZF file (bgzip / htslib) is thousands of independent <= 64 KiB members, which
// the host reader already inflates side by side on CPU threads; on a 16-CPU quota that tops out at ~6 GB/s of text (bns_host.cpp,
// docs/HOST_PATH_NOTES.md).  Here a wavefront takes 64 members, one per lane, so a batch of a few thousand members is inflated by a
// few hundred wavefronts at once; the text goes back to the host parser.
//
// Shape of the decoder (branch-poor on purpose: 64 lanes decode 64 different streams in lockstep, every divergent branch is paid by all):
//   * canonical Huffman decoding WITHOUT a bit-serial walk or a big table: the next 15 stream bits, bit-reversed, are the candidate code
//     left-aligned (w); a code of length l is the one read iff limit[l-1] <= w < limit[l] with limit[l] = (first code of length l +
//     count[l]) << (15 - l), which is non-decreasing in l, so  l = 1 + #{ j : w >= limit[j] }  -- fifteen independent compares, no
//     dependent chain, no divergence; symbol = sym[base[l] + (w >> (15 - l))].
//     Both codes also have a direct table for their short codes (literal/length: up to 10 bits, distance: up to 9 -- symbol and
//     length by the next stream bits: one LDS read); the chain serves the longer ones.
//   * per-lane tables (limits, bases, symbols of both codes, the direct tables: 3872 bytes) live in LDS, interleaved by lane
//     (element pair j of lane i at dword j * MPW + i): same-index accesses of a wavefront's busy lanes hit different banks.
//   * length / distance bases and extra-bit counts are arithmetic, not tables.
//   * code lengths of a dynamic block are staged in a per-member scratch row in global memory (352 bytes, touched once per block).
//
// The same source compiles for the host (STRIDE = 1, plain arrays) -- that is how tests/test_inflate.py checks it against zlib in the
// CPU tier; on the GPU box the kernel's output is compared with zlib's byte for byte.
#pragma once
#include <stdint.h>

#ifndef BNS_INF_FN
#define BNS_INF_FN __device__ __forceinline__
#endif

namespace bns_inf {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// status of one member
enum : u32 {
    INF_OK = 0,
    INF_BAD_BLOCK = 1,        // block type 3
    INF_BAD_STORED = 2,       // LEN != ~NLEN
    INF_BAD_LENGTHS = 3,      // over-subscribed code, repeat without a previous length, too many lengths, no end-of-block code
    INF_BAD_CODE = 4,         // bits that are no code of the current table / a length or distance symbol that does not exist
    INF_BAD_DISTANCE = 5,     // a match reaching in front of the member's output
    INF_OUT_OVERFLOW = 6,     // more output than the member's ISIZE says
    INF_IN_OVERRUN = 7,       // the stream runs past the member's payload
    INF_OUT_SHORT = 8,        // final block ended before ISIZE bytes
};

// per-lane table area, in u16 elements
constexpr int T_LIT_LIMIT = 0;     // 16: limit[l] at l - 1 (l = 1..15), entry 15 = 0xFFFF
constexpr int T_LIT_BASE = 16;     // 16: base[l] at l - 1
constexpr int T_DST_LIMIT = 32;
constexpr int T_DST_BASE = 48;
constexpr int T_NEXT = 64;         // 16: counts, then running offsets, while a table is built
constexpr int T_LIT_SYM = 80;      // 288
constexpr int T_DST_SYM = 368;     // 32 (the code-length code, 19 symbols, is built here too)
constexpr int T_LUT = 400;         // 1024: literal/length code, looked up by the next LUT_BITS stream bits: symbol << 4 | length, 0 = a longer code
constexpr int LUT_BITS = 10;
constexpr int T_DLUT = T_LUT + (1 << LUT_BITS);      // 512: the same for the distance code (one symbol in seven of FASTQ text is a match)
constexpr int DLUT_BITS = 9;
constexpr int T_U16_NOLUT = T_LUT;                   // 800 bytes per lane without the direct tables
constexpr int T_U16 = T_DLUT + (1 << DLUT_BITS);     // 3872 bytes per lane with them
constexpr int SCRATCH_BYTES = 352; // per member: [0, 32) code-length code lengths, [32, 352) literal/length + distance code lengths

BNS_INF_FN u32 brev32(u32 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

BNS_INF_FN u32 load32u(const u8 *p)            // (gfx9 global loads need no alignment)
{
    u32 w;
    __builtin_memcpy(&w, p, 4);
    return w;
}

// S = distance in dwords between consecutive element pairs of one lane (64 on the GPU, 1 on the host); tb = the lane's first element
template <int S>
struct Tables {
    u16 *tb;
    BNS_INF_FN u16 *at(int j) const { return tb + ((j >> 1) * S) * 2 + (j & 1); }
    BNS_INF_FN u32 pair(int jpair) const { return reinterpret_cast<const u32 *>(tb)[jpair * S]; }
    BNS_INF_FN u32 ld(int j) const { return *at(j); }
    BNS_INF_FN void st(int j, u32 v) const { *at(j) = (u16)v; }
};

struct Q128 { u32 x, y, z, w; };
BNS_INF_FN Q128 load128u(const u8 *p)
{
    Q128 q;
    __builtin_memcpy(&q, p, 16);
    return q;
}
BNS_INF_FN u64 load64u(const u8 *p)
{
    u64 w;
    __builtin_memcpy(&w, p, 8);
    return w;
}
BNS_INF_FN void store64u(u8 *p, u64 w) { __builtin_memcpy(p, &w, 8); }

// The bit reader: 64 bits of stream in `bits`, fed 32 at a time from a queue of four words in registers, behind which the NEXT sixteen
// bytes are already on their way -- a lane touches global memory for input once per 128 bits (~20 symbols), and the load it then
// needs was issued 128 bits earlier.  Reads up to 32 bytes past the payload (the caller pads).
struct BitIn {
    const u8 *p;
    u32 fed, n;        // bytes moved into `bits` so far; valid bits in `bits`
    u64 bits;
    Q128 q, qn;        // the words being fed; the sixteen bytes behind them (in flight)
    u32 left;          // words of q not yet fed
    BNS_INF_FN void start(const u8 *base, u32 at)
    {
        p = base; fed = at; n = 0u; bits = 0ULL; left = 4u;
        q = load128u(p + at);
        qn = load128u(p + at + 16u);
    }
    BNS_INF_FN void refill()                       // at least 33 valid bits afterwards
    {
        if (n <= 32u) {
            if (left == 0u) { q = qn; qn = load128u(p + fed + 16u); left = 4u; }
            bits |= (u64)q.x << n;
            q.x = q.y; q.y = q.z; q.z = q.w;
            --left; fed += 4u; n += 32u;
        }
    }
    BNS_INF_FN u32 peek(u32 k) const { return (u32)bits & ((1u << k) - 1u); }
    BNS_INF_FN void drop(u32 k) { bits >>= k; n -= k; }
    BNS_INF_FN u32 take(u32 k) { const u32 v = peek(k); drop(k); return v; }
    BNS_INF_FN u32 consumed() const { return fed - (n >> 3); }       // bytes used up (a byte partly used counts)
};

// Build one canonical code from n code lengths (u8, 0 = unused).  false: over-subscribed.
// WITH_LUT (the literal/length code): also the direct table -- every code of at most LUT_BITS bits at all the indices whose low bits
// are the code as it arrives (LSB first, i.e. bit-reversed); a symbol's code is its rank among the codes of its length plus that
// length's first code, which is what `next` and `base` already hold.
template <int S, bool WITH_LUT = false, int LUT_AT = T_LUT, int LUT_B = LUT_BITS>
BNS_INF_FN bool build_code(const Tables<S> &t, int LIM, int BAS, int SYM, const u8 *lens, u32 n)
{
    for (int l = 0; l < 16; ++l) t.st(T_NEXT + l, 0);
    for (u32 i = 0; i < n; ++i) { const int a = T_NEXT + (lens[i] & 15); t.st(a, t.ld(a) + 1u); }
    u32 code = 0, offs = 0, prev = 0;
    int left = 1;
    bool ok = true;
    for (int l = 1; l <= 15; ++l) {
        const u32 c = t.ld(T_NEXT + l);
        left = (left << 1) - (int)c;
        ok = ok && left >= 0;
        code = (code + prev) << 1;
        prev = c;
        t.st(LIM + l - 1, (code + c) << (15 - l));        // <= 0x8000 for a code that is not over-subscribed
        t.st(BAS + l - 1, offs - code);                   // (mod 2^16)
        t.st(T_NEXT + l, offs);
        offs += c;
    }
    t.st(LIM + 15, 0xFFFFu);
    t.st(BAS + 15, 0);
    if (!ok) return false;
    if (WITH_LUT)
        for (int j = 0; j < (1 << LUT_B); j += 2) reinterpret_cast<u32 *>(t.at(LUT_AT + j))[0] = 0u;
    for (u32 i = 0; i < n; ++i) {
        const u32 l = lens[i] & 15u;
        if (l) {
            const u32 o = t.ld(T_NEXT + (int)l);
            t.st(SYM + (int)o, i);
            t.st(T_NEXT + (int)l, o + 1u);
            if (WITH_LUT && l <= (u32)LUT_B) {
                const u32 c = (o - t.ld(BAS + (int)l - 1)) & 0xFFFFu;
                const u32 e = (i << 4) | l;
                for (u32 j = brev32(c) >> (32u - l); j < (1u << LUT_B); j += 1u << l) t.st(LUT_AT + (int)j, e);
            }
        }
    }
    return true;
}

// The sixteen limits of one code, two per register (loaded once per block: the compare chain then touches no memory)
struct Lim { u32 p[8]; };
template <int S>
BNS_INF_FN Lim load_limits(const Tables<S> &t, int LIM)
{
    Lim L;
#pragma unroll
    for (int q = 0; q < 8; ++q) L.p[q] = t.pair((LIM >> 1) + q);
    return L;
}

// One symbol of the code whose limits are in L and whose bases / symbols are at (BAS, SYM); SYMMASK bounds the symbol index.
// A length of 16 means the bits are no code.
template <int S, int BAS, int SYM, int SYMMASK>
BNS_INF_FN u32 decode_sym(const Tables<S> &t, const Lim &L, BitIn &in, bool &bad)
{
    const u32 w = brev32((u32)in.bits) >> 17;
    u32 len = 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        len += (w >= (L.p[q] & 0xFFFFu)) ? 1u : 0u;
        len += (w >= (L.p[q] >> 16)) ? 1u : 0u;
    }
    bad = bad || len > 15u;
    len = len > 15u ? 15u : len;
    const u32 idx = (t.ld(BAS + (int)len - 1) + (w >> (15u - len))) & 0xFFFFu;
    in.drop(len);
    return t.ld(SYM + (int)(idx & (u32)SYMMASK));
}

// len (3..16) bytes of the sixteen in (a, b) to dst, exactly: two stores that may overlap each other but touch nothing else
BNS_INF_FN void store_exact(u8 *dst, u64 a, u64 b, u32 len)
{
    if (len >= 8u) {
        const u32 sh = 8u * (len - 8u);                     // 0..64
        const u64 second = sh == 0u ? a : (sh == 64u ? b : ((a >> sh) | (b << (64u - sh))));
        store64u(dst, a);
        store64u(dst + len - 8u, second);
    } else if (len >= 4u) {
        const u32 lo = (u32)a, hi = (u32)(a >> (8u * (len - 4u)));
        __builtin_memcpy(dst, &lo, 4);
        __builtin_memcpy(dst + len - 4u, &hi, 4);
    } else {
        const u16 lo = (u16)a;
        __builtin_memcpy(dst, &lo, 2);
        dst[2] = (u8)(a >> 16);
    }
}

// Inflate one member: `in` (in_len bytes of raw DEFLATE; readable up to in_len + 40) -> out (exactly out_len bytes expected; matches
// never reach in front of out).  scratch: SCRATCH_BYTES of this member's own.  Returns the status; *out_n = bytes written.
// LUT: the literal/length code's direct table is there (T_U16 elements per lane) and used; without it (T_U16_NOLUT elements) every
// symbol goes through the compare chain -- a third of the LDS, so three times the wavefronts per CU for batches that can fill them.
template <int S, bool LUT = true>
BNS_INF_FN u32 inflate_member(const u8 *in_p, u32 in_len, u8 *out, u32 out_len, const Tables<S> &t, u8 *scratch, u32 *out_n)
{
    BitIn in;
    in.start(in_p, 0u);
    u32 o = 0, status = INF_OK;
    u32 pend_len = 0u, pend_dst = 0u;             // a short match whose source bytes have been asked for and not yet stored
    u64 pa = 0ULL, pb = 0ULL;
    bool last = false;
    while (!last && status == INF_OK) {
        if (in.consumed() > in_len) { status = INF_IN_OVERRUN; break; }
        in.refill();
        last = in.peek(1) != 0u;
        const u32 type = (in.peek(3) >> 1);
        in.drop(3);
        if (type == 0u) {
            in.drop(in.n & 7u);
            in.refill();
            const u32 len = in.peek(16);
            in.drop(16);
            const u32 nlen = in.peek(16);
            in.drop(16);
            if ((len ^ 0xFFFFu) != nlen) { status = INF_BAD_STORED; break; }
            const u32 src = in.consumed();                 // (a whole number of bytes is buffered: the reader starts again behind the block)
            if (src + len > in_len) { status = INF_IN_OVERRUN; break; }
            if (o + len > out_len) { status = INF_OUT_OVERFLOW; break; }
            for (u32 i = 0; i < len; ++i) out[o + i] = in_p[src + i];
            o += len;
            in.start(in_p, src + len);
            continue;
        }
        if (type == 3u) { status = INF_BAD_BLOCK; break; }
        u8 *L = scratch + 32;
        u32 hlit, hdist;
        if (type == 1u) {
            hlit = 288u; hdist = 30u;
            for (u32 i = 0; i < 288u; ++i) L[i] = (u8)(i < 144u ? 8u : i < 256u ? 9u : i < 280u ? 7u : 8u);
            for (u32 i = 0; i < 30u; ++i) L[288u + i] = 5u;
        } else {
            in.refill();
            hlit = in.take(5) + 257u;
            hdist = in.take(5) + 1u;
            const u32 hclen = in.take(4) + 4u;
            if (hlit > 286u || hdist > 30u) { status = INF_BAD_LENGTHS; break; }
            for (u32 i = 0; i < 19u; ++i) scratch[i] = 0u;
            for (u32 i = 0; i < hclen; ++i) {
                // 16,17,18,0, then 8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 (RFC 1951 3.2.7): up from 8 and down from 7 in turns
                const u32 j = i - 4u;
                const u32 ord = i < 3u ? 16u + i : (i == 3u ? 0u : ((j & 1u) ? 7u - (j >> 1) : 8u + (j >> 1)));
                in.refill();
                scratch[ord] = (u8)in.take(3);
            }
            if (!build_code<S>(t, T_DST_LIMIT, T_DST_BASE, T_DST_SYM, scratch, 19u)) { status = INF_BAD_LENGTHS; break; }
            const u32 total = hlit + hdist;
            u32 i = 0;
            bool bad = false;
            const Lim lim_cl = load_limits<S>(t, T_DST_LIMIT);
            while (i < total && !bad) {
                if (in.consumed() > in_len + 4u) { bad = true; break; }    // (a damaged member must not read far behind its payload)
                in.refill();
                const u32 s = decode_sym<S, T_DST_BASE, T_DST_SYM, 31>(t, lim_cl, in, bad);
                if (bad) break;
                if (s < 16u) { L[i++] = (u8)s; continue; }
                u32 rep, val = 0u;
                if (s == 16u) { if (i == 0u) { bad = true; break; } val = L[i - 1u]; rep = 3u + in.take(2); }
                else if (s == 17u) rep = 3u + in.take(3);
                else rep = 11u + in.take(7);
                if (i + rep > total) { bad = true; break; }
                for (u32 r = 0; r < rep; ++r) L[i + r] = (u8)val;
                i += rep;
            }
            if (bad || L[256] == 0u) { status = INF_BAD_LENGTHS; break; }
            if (in.consumed() > in_len) { status = INF_IN_OVERRUN; break; }
        }
        if (!build_code<S, LUT>(t, T_LIT_LIMIT, T_LIT_BASE, T_LIT_SYM, L, hlit) ||
            !build_code<S, LUT, T_DLUT, DLUT_BITS>(t, T_DST_LIMIT, T_DST_BASE, T_DST_SYM, L + hlit, hdist)) { status = INF_BAD_LENGTHS; break; }
        // the block's symbols
        const Lim lim_lit = load_limits<S>(t, T_LIT_LIMIT), lim_dst = load_limits<S>(t, T_DST_LIMIT);
        for (;;) {
            if (in.consumed() > in_len + 4u) { status = INF_IN_OVERRUN; break; }
            in.refill();
            bool bad = false;
            // (the usual symbol -- a literal with a short code -- is one table read; the compare chain only for codes beyond LUT_BITS)
            const u32 e = LUT ? t.ld(T_LUT + (int)((u32)in.bits & ((1u << LUT_BITS) - 1u))) : 0u;
            u32 s;
            if (LUT && e != 0u) { s = e >> 4; in.drop(e & 15u); }
            else s = decode_sym<S, T_LIT_BASE, T_LIT_SYM, 511>(t, lim_lit, in, bad);
            if (bad) { status = INF_BAD_CODE; break; }
            if (s < 256u) {
                if (o >= out_len) { status = INF_OUT_OVERFLOW; break; }
#ifdef BNS_INF_ABLATE_LITERAL_STORES                        // measurement builds only (wrong output): what do the literal stores cost?
                ++o;
#else
                out[o++] = (u8)s;
#endif
                continue;
            }
            if (s == 256u) break;
            if (s > 285u) { status = INF_BAD_CODE; break; }
            // length: 257..264 -> 3..10; 265..284 -> e = (s - 261) >> 2 extra bits, base 3 + ((4 + ((s - 265) & 3)) << e); 285 -> 258
            u32 len;
            if (s < 265u) len = s - 254u;
            else if (s == 285u) len = 258u;
            else { const u32 e = (s - 261u) >> 2; len = 3u + ((4u + ((s - 265u) & 3u)) << e) + in.take(e); }
            in.refill();
            const u32 de = LUT ? t.ld(T_DLUT + (int)((u32)in.bits & ((1u << DLUT_BITS) - 1u))) : 0u;
            u32 ds;
            if (LUT && de != 0u) { ds = de >> 4; in.drop(de & 15u); }
            else ds = decode_sym<S, T_DST_BASE, T_DST_SYM, 31>(t, lim_dst, in, bad);
            if (bad || ds > 29u) { status = INF_BAD_CODE; break; }
            // distance: 0..3 -> 1..4; else e = (ds >> 1) - 1 extra bits, base 1 + ((2 + (ds & 1)) << e)
            u32 dist;
            if (ds < 4u) dist = ds + 1u;
            else { const u32 e = (ds >> 1) - 1u; dist = 1u + ((2u + (ds & 1u)) << e) + in.take(e); }
            if (dist > o) { status = INF_BAD_DISTANCE; break; }
            if (o + len > out_len) { status = INF_OUT_OVERFLOW; break; }
            // The copy.  A lane's load of bytes it stored itself needs no wait for the store (one wave's memory operations on an
            // address stay in order), but every load is a round trip to L2, and in a wavefront of 64 streams some lane has a match in
            // nearly every step.  The usual match (<= 16 bytes from >= 16 back) therefore only ASKS for its sixteen source bytes
            // here; they are stored -- exactly `len` of them -- when the lane next has a match (whose source may be those bytes) or
            // the block ends, i.e. after at least one more symbol has been decoded under the load's latency.  Literals in between go
            // to their own addresses behind the match and need not wait.
            if (pend_len) { store_exact(out + pend_dst, pa, pb, pend_len); pend_len = 0u; }
            if (dist >= 16u && len <= 16u) {
                const Q128 v = load128u(out + o - dist);
                pa = (u64)v.x | ((u64)v.y << 32); pb = (u64)v.z | ((u64)v.w << 32);
                pend_dst = o; pend_len = len;
            } else if (dist >= 16u && o + len + 16u <= out_len) {
                // (long: eight bytes per trip, two trips in flight; the up to fifteen bytes written past the match are inside the
                // member's own text and overwritten by what follows)
                for (u32 i = 0; i < len; i += 16u) {
                    const u64 a = load64u(out + o + i - dist), b = load64u(out + o + i + 8u - dist);
                    store64u(out + o + i, a); store64u(out + o + i + 8u, b);
                }
            } else if (dist >= 8u && o + len + 8u <= out_len) {
                for (u32 i = 0; i < len; i += 8u) store64u(out + o + i, load64u(out + o + i - dist));
            } else {
                for (u32 i = 0; i < len; ++i) out[o + i] = out[o + i - dist];
            }
            o += len;
        }
        if (pend_len) { store_exact(out + pend_dst, pa, pb, pend_len); pend_len = 0u; }
    }
    if (status == INF_OK && in.consumed() > in_len) status = INF_IN_OVERRUN;
    if (status == INF_OK && o != out_len) status = INF_OUT_SHORT;
    *out_n = o;
    return status;
}

// CRC-32 (the gzip one: reflected 0xEDB88320), byte at a time over a 256-entry table the caller provides
BNS_INF_FN u32 crc32_entry(u32 i)
{
    u32 c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    return c;
}
BNS_INF_FN u32 crc32_bytes(const u32 *tbl, const u8 *p, u32 n)
{
    u32 c = 0xFFFFFFFFu;
    for (u32 i = 0; i < n; ++i) c = tbl[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return ~c;
}

}  // namespace bns_inf
