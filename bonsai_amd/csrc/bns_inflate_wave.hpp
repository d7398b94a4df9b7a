// bns_inflate_wave.hpp -- raw DEFLATE (RFC 1951) decoder for BGZF members, ONE MEMBER PER WAVEFRONT (device only).
//
// The other form (bns_inflate.hpp) gives every member a lane: a member's symbols are a serial chain of ~124 wavefront instructions each
// (docs/KERNEL_NOTES.md), 20 ms for 64 KiB whatever runs beside it, so a batch of a few thousand members -- what a reader thread holds at
// a time -- inflates at 8-13 GB/s.  Here a wavefront decodes ONE member, and the 64 lanes decode it SPECULATIVELY:
//   * a round looks at the next 64 bit positions of the stream.  Lane i assumes a symbol starts at position i and decodes it whole:
//     the literal/length code by the direct table, and for a length its extra bits, the distance code (direct table) and its extra bits
//     -- two LDS look-ups, all lanes at once.  It ends up with the position of the symbol behind its own.
//   * which lanes guessed right is a walk from lane 0 along those positions: one v_readlane per symbol on the scalar unit -- the only
//     serial part of a round (~7 scalar instructions per symbol; the first form of this kernel decoded on the scalar unit outright:
//     480 cycles per literal, 8.4 ms per member).  The walk stops at the round's end or at a symbol the lanes do not decode (a code
//     longer than the direct tables, end of block), which the scalar path then takes.
//   * the lanes on the walk are the round's symbols: a prefix sum of their output lengths (DPP scan) gives every literal its byte in
//     the staged batch and every match its queue slot.
//   * everything else is by all 64 lanes as well: a block's tables (counts and canonical ranks by ballots, the direct tables filled one
//     symbol per lane), the match copies (each lane fetches the source bytes of one queued match that lie in text already written;
//     matches that reach into the batch itself are resolved in LDS, 64 bytes per step), the write of a finished batch (~1 KB, 16 bytes
//     per lane, coalesced), the CRC (64 slices, combined with zlib's x^n mod p arithmetic).  Input comes through a ring in LDS that the
//     lanes top up 256 bytes at a time, one load ahead; output is staged in LDS per batch: the decoder never waits for global memory.
// Per wavefront: 4 KB + 2 KB direct tables (literal/length 10 bits, distance 9), 1.3 KB of entries by canonical rank for longer
// codes, 320 bytes of code lengths, 1.3 KB of staging, a 512-byte input ring, 512 bytes of match queue: 9.9 KB, sixteen wavefronts
// per CU = 4096 members resident on the chip.
//
// Same contract as bns_inf::inflate_member (status codes, bytes written, CRC-32 of them); tests/test_inflate.py runs both forms
// against zlib on the GPU.  What it replaces in the reference: gzread under kseq (kseq_declare.h:112-145, klib/kseq.h:177-225).
#pragma once
#include "bns_inflate.hpp"

namespace bns_infw {

using bns_inf::u8;
using bns_inf::u16;
using bns_inf::u32;
using bns_inf::u64;

constexpr int LB = 10;              // direct table of the literal/length code: codes of at most LB bits
constexpr int DB = 9;               // ... of the distance code
constexpr u32 FLUSH_AT = 960;         // a batch of output is flushed when it has grown beyond this
constexpr u32 STAGE_CAP = FLUSH_AT + 258;        // ... so a match always fits

// a table entry (direct table or by-rank table): bits 0-3 code length, 4-7 extra bits, 8-9 kind, 16-31 value
constexpr u32 F_LIT = 1u << 15;                      // a literal (kind bits 0)
constexpr u32 K_LIT = F_LIT, K_LEN = 1u << 8, K_EOB = 2u << 8, K_BAD = 3u << 8;      // (for a distance entry: kind 0 = fine, 3 = no such symbol)

constexpr u32 RING = 128;             // words of input in LDS

// T: what a decoded byte is written as -- u8 (a BGZF member: its own text is all a match can reach), or u16 for a stream entered in the
// MIDDLE (bns_inflate.hip, "one gzip stream"): the 32 KiB in front of the entry point are not known yet, so the output is SYMBOLS,
// a byte (< 256) or a marker 0x8000 | j = "byte j of those 32 KiB", and matches copy symbols -- the same code, elements twice as wide.
template <class T>
struct alignas(16) WaveLdsT {
    u32 lut[1 << LB];
    u32 dlut[1 << DB];               // (the code-length code's direct table, 7 bits, lives here while a dynamic block's lengths are read)
    u32 lit_rank[288];
    u32 dst_rank[32];
    T stage[(STAGE_CAP + 15) & ~15u];
    u32 ring[RING];                  // input word w at [w % RING]
    u32 q[128];                      // queued match j: [2j] staging position | length << 16, [2j + 1] distance
    u8 lens[320];
};
using WaveLds = WaveLdsT<u8>;

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 rdlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
// (this clang has no __builtin_amdgcn_writelane; the intrinsic by its name: s_mov_b32 m0, lane; v_writelane_b32 v, val, m0)
extern "C" __device__ int bns_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ u32 wrlane(u32 val, u32 l, u32 old) { return (u32)bns_llvm_writelane((int)val, (int)l, (int)old); }
__device__ __forceinline__ u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ u32 below(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }   // set bits of m under this lane
__device__ __forceinline__ void wave_sync() { __syncthreads(); }               // (one wavefront per block: a fence, not a wait for anybody)

// literal/length symbol -> entry without its code length
__device__ __forceinline__ u32 litlen_entry(u32 s)
{
    if (s < 256u) return K_LIT | (s << 16);
    if (s == 256u) return K_EOB;
    if (s > 285u) return K_BAD;
    if (s < 265u) return K_LEN | ((s - 254u) << 16);
    if (s == 285u) return K_LEN | (258u << 16);
    const u32 e = (s - 261u) >> 2;
    return K_LEN | (e << 4) | ((3u + ((4u + ((s - 265u) & 3u)) << e)) << 16);
}
__device__ __forceinline__ u32 dist_entry(u32 s)
{
    if (s > 29u) return K_BAD;
    if (s < 4u) return (s + 1u) << 16;
    const u32 e = (s >> 1) - 1u;
    return (e << 4) | ((1u + ((2u + (s & 1u)) << e)) << 16);
}

// zlib's CRC-32 combination arithmetic: polynomials mod p, reflected (crc32.c multmodp / x2nmodp)
constexpr u32 multmodp(u32 a, u32 b)
{
    u32 m = 1u << 31, p = 0u;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0u) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
struct X2N {
    u32 v[32];
    constexpr X2N() : v{}
    {
        u32 p = 1u << 30;                      // x^1
        for (int k = 0; k < 32; ++k) { v[k] = p; p = multmodp(p, p); }
    }
};
__device__ const X2N x2n_table{};
__device__ __forceinline__ u32 x2nmodp(u32 n, u32 k)               // x^(n * 2^k) mod p
{
    u32 p = 1u << 31;
    while (n) {
        if (n & 1u) p = multmodp(x2n_table.v[k & 31u], p);
        n >>= 1;
        ++k;
    }
    return p;
}

// inclusive prefix sum over the wavefront (DPP: shifts inside the rows of 16, then the rows' totals broadcast on)
__device__ __forceinline__ u32 wave_scan(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}

// The member's state.  Everything named here is wave-uniform except `pre`.
template <class T>
struct Dec {
    WaveLdsT<T> *S;
    const u8 *in_p, *comp_end, *wbase;   // wbase: in_p rounded down to a word
    T *out;
    u32 in_len, out_len, mis;         // mis = in_p - wbase
    u32 lane;
    u32 bp;                           // the stream's position in bits from wbase
    u32 filled;                       // input words [0, filled) have been put into the ring (the last RING of them are there)
    u32 pre;                          // VGPR: lane l = input word filled + l, asked for a top-up ago
    u32 ob, so, nq;                   // bytes written to global; bytes staged; matches queued
    u32 status;

    __device__ __forceinline__ u32 ldw(u32 w) const
    {
        const u8 *a = wbase + 4ULL * w;
        u32 x = 0u;
        if (a + 4 <= comp_end) x = *reinterpret_cast<const u32 *>(a);
        return x;
    }
    // the stream from byte `at` of the member's payload
    __device__ __forceinline__ void start(u32 at, u32 bit = 0u)
    {
        bp = 8u * (mis + at) + bit;
        filled = (bp >> 5) & ~63u;
        pre = ldw(filled + lane);
    }
    // the ring holds the words up to bit `upto`
    __device__ __forceinline__ void need(u32 upto)
    {
        while (filled * 32u < upto) {
            S->ring[(filled + lane) & (RING - 1u)] = pre;
            filled += 64u;
            pre = ldw(filled + lane);
            wave_sync();
        }
    }
    // 64 bits of the stream at bit position b, lane by lane
    __device__ __forceinline__ u64 bits_at(u32 b) const
    {
        const u32 w = b >> 5, s = b & 31u;
        const u32 x0 = S->ring[w & (RING - 1u)], x1 = S->ring[(w + 1u) & (RING - 1u)], x2 = S->ring[(w + 2u) & (RING - 1u)];
        const u64 lo = ((u64)x1 << 32) | x0, hi = x2;
        return (lo >> s) | ((hi << 1) << (63u - s));            // (= hi << (64 - s), and nothing for s = 0, without a branch)
    }
    // ... at the stream's position, for the scalar paths
    __device__ __forceinline__ u64 peek64()
    {
        need(bp + 96u);
        const u64 v = bits_at(bp);
        return ((u64)uni((u32)(v >> 32)) << 32) | uni((u32)v);
    }
    __device__ __forceinline__ u32 consumed() const { return ((bp + 7u) >> 3) - mis; }      // bytes used up (a byte partly used counts)

    // A finished batch: the staged bytes [0, so) become out[ob, ob + so).
    __device__ __forceinline__ void flush()
    {
        wave_sync();
        const bool mine = lane < nq;
        u32 qa = 0u, dist = 0u;
        if (mine) { qa = S->q[2u * lane]; dist = S->q[2u * lane + 1u]; }
        const u32 sp = qa & 0xFFFFu, len = qa >> 16;
        // 1. the source bytes that lie in text already written (in front of this batch): one match per lane
        if (mine && dist > sp) {
            const u32 g = min(len, dist - sp);
            const T *src = out + ((u64)ob + sp - dist);
            T *dst = S->stage + sp;
            constexpr u32 E = 8u / (u32)sizeof(T);              // elements per 64-bit load
            u32 k = 0u;
            for (; k + E <= g; k += E) {
                const u64 v = bns_inf::load64u(reinterpret_cast<const u8 *>(src + k));
#pragma unroll
                for (u32 b = 0; b < E; ++b) dst[k + b] = (T)(v >> (8u * (u32)sizeof(T) * b));
            }
            for (; k < g; ++k) dst[k] = src[k];
        }
        wave_sync();
        // 2. the source bytes inside the batch: match by match in stream order (one may copy what an earlier one produced), 64 lanes a step
        u64 todo = ballot(mine && dist < sp + len);
        while (todo) {
            const u32 j = (u32)__builtin_ctzll(todo);
            todo &= todo - 1ULL;
            const u32 a = rdlane(qa, j), d = rdlane(dist, j);
            const u32 msp = a & 0xFFFFu, mlen = a >> 16;
            const u32 k0 = d > msp ? d - msp : 0u;            // bytes [0, k0) came from global memory in step 1
            const u32 rem = mlen - k0, dst0 = msp + k0, src0 = dst0 - d;
            if (d >= 64u) {
                // (a wavefront's LDS operations execute in order: a step's reads are in front of its writes, the next step's behind them)
                for (u32 r = 0u; r < rem; r += 64u) {
                    const u32 k = r + lane;
                    if (k < rem) S->stage[dst0 + k] = S->stage[src0 + k];
                }
            } else {
                // the match repeats the d bytes in front of it: every byte's source is among them
                for (u32 r = 0u; r < rem; r += 64u) {
                    const u32 k = r + lane;
                    if (k < rem) S->stage[dst0 + k] = S->stage[src0 + k % d];
                }
            }
            wave_sync();
        }
        // 3. out it goes
        if (ob + so > out_len) {
            if (status == bns_inf::INF_OK) status = bns_inf::INF_OUT_OVERFLOW;
            so = out_len - ob;
        }
        u8 *o = reinterpret_cast<u8 *>(out + ob);
        const u8 *sb = reinterpret_cast<const u8 *>(S->stage);
        const u32 nb = so * (u32)sizeof(T);
        for (u32 j = lane * 16u; j < nb; j += 1024u) {
            if (j + 16u <= nb) {
                const uint4 v = *reinterpret_cast<const uint4 *>(sb + j);
                __builtin_memcpy(o + j, &v, 16);
            } else {
                for (u32 t = j; t < nb; ++t) o[t] = sb[t];
            }
        }
        ob += so; so = 0u; nq = 0u;
        wave_sync();
    }
};

// One canonical code from n code lengths, built by the whole wavefront.  len_of(i): the length of symbol i (0 = unused), called by
// every lane with its own i; ent_of(s): a symbol's entry.  Fills the direct table `lut` (codes of at most BITS bits; 0 = a longer
// code or no code), `rank_tab` (entry of the symbol at canonical rank r, for the longer ones) and, in lanes 0..14 of lim_v / base_v,
// limit[l] = (first code of length l + count[l]) << (15 - l) and base[l] = rank of the first code of length l - that code, for
// l = lane + 1 (what bns_inflate.hpp keeps in LDS: the same compare chain, done here by one ballot).  false: over-subscribed.
template <int BITS, u32 RANKS, class LenOf, class EntOf>
__device__ __forceinline__ bool build_code(u32 lane, u32 n, LenOf len_of, EntOf ent_of, u32 *lut, u32 *rank_tab, u32 &lim_v, u32 &base_v)
{
    for (u32 j = lane; j < (1u << BITS); j += 64u) lut[j] = 0u;
    for (u32 j = lane; j < RANKS; j += 64u) rank_tab[j] = 0u;          // (ranks no symbol has: entry 0 = no code)
    u32 cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0u;
    for (u32 c = 0u; c < n; c += 64u) {
        const u32 i = c + lane;
        const u32 l = i < n ? len_of(i) : 0u;
#pragma unroll
        for (int k = 1; k < 16; ++k) cnt[k] += (u32)__builtin_popcountll(ballot(l == (u32)k));
    }
    // (scalar, fifteen steps)
    u32 first_v = 0u, off_v = 0u;
    lim_v = 0xFFFFu; base_v = 0u;
    u32 code = 0u, offs = 0u, prev = 0u;
    int left = 1;
    bool ok = true;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        const u32 c = cnt[l];
        left = (left << 1) - (int)c;
        ok = ok && left >= 0;
        code = (code + prev) << 1;
        prev = c;
        lim_v = wrlane(((code + c) << (15 - l)) & 0x1FFFFu, (u32)(l - 1), lim_v);
        base_v = wrlane((offs - code) & 0xFFFFu, (u32)(l - 1), base_v);
        first_v = wrlane(code, (u32)l, first_v);
        off_v = wrlane(offs, (u32)l, off_v);
        offs += c;
    }
    if (!ok) return false;
    wave_sync();                                                // (the zeroed table before the fills)
    u32 run[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) run[l] = 0u;
    for (u32 c = 0u; c < n; c += 64u) {
        const u32 i = c + lane;
        const u32 l = i < n ? len_of(i) : 0u;
        u32 rank = 0u;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const u64 b = ballot(l == (u32)k);
            if (l == (u32)k) rank = run[k] + below(b);
            run[k] += (u32)__builtin_popcountll(b);
        }
        const u32 first = (u32)__shfl((int)first_v, (int)l, 64), off = (u32)__shfl((int)off_v, (int)l, 64);
        if (l) {
            const u32 cd = first + rank;
            const u32 e = ent_of(i) | l;
            rank_tab[off + rank] = e;
            if (l <= (u32)BITS)
                for (u32 j = bns_inf::brev32(cd) >> (32u - l); j < (1u << BITS); j += 1u << l) lut[j] = e;
        }
    }
    wave_sync();
    return true;
}

// a code of more than the direct table's bits (or bits that are no code): the compare chain of bns_inflate.hpp as one ballot
__device__ __forceinline__ u32 slow_entry(u32 lane, u64 bits, u32 lim_v, u32 base_v, const u32 *rank_tab, u32 mask, u32 &len)
{
    const u32 w = bns_inf::brev32((u32)bits) >> 17;
    len = 1u + (u32)__builtin_popcountll(ballot(lane < 15u && w >= lim_v));
    if (len > 15u) return K_BAD;
    const u32 idx = (rdlane(base_v, len - 1u) + (w >> (15u - len))) & 0xFFFFu;
    if (idx > mask) return K_BAD;
    const u32 e = uni(rank_tab[idx]);
    return e ? e : K_BAD;
}

// A stream entered in the middle (STREAM; all of it wave-uniform).  The text goes behind out[0, prefix): what lies in front of the entry
// point, as far as a match can reach -- markers, or the bytes themselves where they are known.  Only WHOLE blocks count: whatever stops
// the decoder inside a block (the input ends there and more will come; room; an error) takes the result back to the last block boundary.
struct StreamIO {
    u32 bit0;            // in: the entry point is bit `bit0` (0-7) of in_p[0]: a block header
    u32 stop_bit;        // in: stop at the first block boundary at or behind this many bits from in_p[0]
    u32 prefix;          // in: elements in front of the text
    u32 end_bit;         // out: bits from in_p[0] behind the last whole block decoded
    u32 member_end;      // out: that block was the member's last
};

// Inflate one member with the whole wavefront.  Returns the status; *out_n = bytes written to out (STREAM: elements behind the prefix
// that belong to whole blocks; the status is INF_OK when there is at least one, whatever stopped the decoder behind it).
template <class T, bool STREAM>
__device__ __forceinline__ u32 inflate_member_wave(WaveLdsT<T> *S, const u8 *in_p, u32 in_len, const u8 *comp_end, T *out, u32 out_len, u32 *out_n, StreamIO *io = nullptr)
{
    using namespace bns_inf;
    Dec<T> D;
    D.S = S; D.in_p = in_p; D.comp_end = comp_end; D.out = out; D.in_len = in_len; D.out_len = out_len; D.lane = lane_id();
    D.mis = (u32)((uintptr_t)in_p & 3u);
    D.wbase = in_p - D.mis;
    D.ob = STREAM ? io->prefix : 0u; D.so = 0u; D.nq = 0u; D.status = INF_OK;
    const u32 lane = D.lane;
    D.start(0u, STREAM ? io->bit0 : 0u);
    bool last = false;
    u32 blk_bp = D.bp, blk_out = D.ob;                            // (STREAM) the last block boundary: stream position, elements in front of it
    while (!last && D.status == INF_OK) {
        if (STREAM) {
            if (D.consumed() > in_len) { D.status = INF_IN_OVERRUN; break; }         // (the block in front "ended" in the zeros behind the bytes: no boundary)
            if (D.ob + D.so > out_len) { D.status = INF_OUT_OVERFLOW; break; }      // (staged text that will not fit: the block in front is not whole in memory)
            blk_bp = D.bp; blk_out = D.ob + D.so;
            if (D.bp - 8u * D.mis >= io->stop_bit) break;
        }
        if (D.consumed() > in_len) { D.status = INF_IN_OVERRUN; break; }
        u64 hb = D.peek64();
        last = (hb & 1ULL) != 0ULL;
        const u32 type = (u32)(hb >> 1) & 3u;
        D.bp += 3u;
        if (type == 3u) { D.status = INF_BAD_BLOCK; break; }
        if (type == 0u) {
            D.bp = (D.bp + 7u) & ~7u;
            hb = D.peek64();
            const u32 len = (u32)hb & 0xFFFFu, nlen = (u32)(hb >> 16) & 0xFFFFu;
            D.bp += 32u;
            if ((len ^ 0xFFFFu) != nlen) { D.status = INF_BAD_STORED; break; }
            const u32 src = D.consumed();
            if (src + len > in_len) { D.status = INF_IN_OVERRUN; break; }
            D.flush();
            if (D.status != INF_OK) break;
            if (D.ob + len > out_len) { D.status = INF_OUT_OVERFLOW; break; }
            for (u32 i = lane; i < len; i += 64u) out[D.ob + i] = in_p[src + i];
            D.ob += len;
            wave_sync();
            D.start(src + len);
            continue;
        }
        u32 hlit = 288u, hdist = 30u;
        if (type == 1u) {
            for (u32 i = lane; i < 320u; i += 64u) S->lens[i] = (u8)(i < 144u ? 8u : i < 256u ? 9u : i < 280u ? 7u : i < 288u ? 8u : 5u);
            wave_sync();
        } else {
            hb = D.peek64();
            hlit = ((u32)hb & 31u) + 257u;
            hdist = ((u32)(hb >> 5) & 31u) + 1u;
            const u32 hclen = ((u32)(hb >> 10) & 15u) + 4u;
            D.bp += 14u;
            if (hlit > 286u || hdist > 30u) { D.status = INF_BAD_LENGTHS; break; }
            hb = D.peek64();                                     // (19 x 3 bits: one look)
            u32 cl_v = 0u;                                       // VGPR: lane s = length of code-length symbol s
            for (u32 i = 0u; i < hclen; ++i) {
                const u32 j = i - 4u;
                const u32 ord = i < 3u ? 16u + i : (i == 3u ? 0u : ((j & 1u) ? 7u - (j >> 1) : 8u + (j >> 1)));
                cl_v = wrlane((u32)(hb >> (3u * i)) & 7u, ord, cl_v);
            }
            D.bp += 3u * hclen;
            u32 lim_c, base_c;
            // (its entries are the symbol itself << 16; the rank table of the code-length code borrows the distance code's)
            if (!build_code<7, 32u>(lane, 19u, [&](u32) { return cl_v; }, [](u32 s) { return s << 16; }, S->dlut, S->dst_rank, lim_c, base_c)) { D.status = INF_BAD_LENGTHS; break; }
            const u32 total = hlit + hdist;
            u32 i = 0u;
            bool bad = false;
            u32 prev_len = 0u;
            u64 cb = 0ULL;                                        // the stream at D.bp, cn bits of it good
            u32 cn = 0u;
            while (i < total) {
                if (cn < 15u) {
                    if (D.consumed() > in_len + 4u) { bad = true; break; }
                    cb = D.peek64();
                    cn = 64u;
                }
                u32 e = uni(S->dlut[(u32)cb & 127u]), cl = e & 15u;
                if (e == 0u) { e = slow_entry(lane, cb, lim_c, base_c, S->dst_rank, 31u, cl); if (e == K_BAD) { bad = true; break; } }
                cb >>= cl; cn -= cl; D.bp += cl;
                const u32 s = e >> 16;
                if (s < 16u) {
                    if (lane == 0u) S->lens[i] = (u8)s;
                    prev_len = s; ++i;
                    continue;
                }
                u32 rep, val = 0u, xb;
                if (s == 16u) { if (i == 0u) { bad = true; break; } val = prev_len; rep = 3u + ((u32)cb & 3u); xb = 2u; }
                else if (s == 17u) { rep = 3u + ((u32)cb & 7u); xb = 3u; }
                else { rep = 11u + ((u32)cb & 127u); xb = 7u; }
                cb >>= xb; cn -= xb; D.bp += xb;
                if (i + rep > total) { bad = true; break; }
                for (u32 r = lane; r < rep; r += 64u) S->lens[i + r] = (u8)val;
                i += rep;
                prev_len = val;
            }
            wave_sync();
            if (bad || uni(S->lens[256]) == 0u) { D.status = INF_BAD_LENGTHS; break; }
            if (D.consumed() > in_len) { D.status = INF_IN_OVERRUN; break; }
        }
        u32 lim_l, base_l, lim_d, base_d;
        const u8 *lens = S->lens;
        if (!build_code<LB, 288u>(lane, hlit, [&](u32 i) { return (u32)lens[i]; }, [](u32 s) { return litlen_entry(s); }, S->lut, S->lit_rank, lim_l, base_l) ||
            !build_code<DB, 32u>(lane, hdist, [&](u32 i) { return (u32)lens[hlit + i]; }, [](u32 s) { return dist_entry(s); }, S->dlut, S->dst_rank, lim_d, base_d)) {
            D.status = INF_BAD_LENGTHS;
            break;
        }
        // The block's symbols, a round of 64 bit positions at a time.
        for (;;) {
            if (D.so > FLUSH_AT || D.nq == 64u) {
                D.flush();
                if (D.status != INF_OK) break;
                if (D.consumed() > in_len + 4u) { D.status = INF_IN_OVERRUN; break; }
            }
            D.need(D.bp + 64u + 96u);
            // every lane: the symbol that starts at its bit position, if one does
            const u64 v = D.bits_at(D.bp + lane);
            const u32 e1 = S->lut[(u32)v & ((1u << LB) - 1u)];
            const bool is_lit = (e1 & F_LIT) != 0u, is_len = (e1 & (F_LIT | K_BAD)) == K_LEN;
            // (no branches in here: both readings of the second look-up are computed and selected)
            const u32 l1 = e1 & 15u;
            const u32 eb = is_len ? (e1 >> 4) & 15u : 0u;
            const u32 used1 = l1 + eb;
            const u32 lenval = (e1 >> 16) + ((u32)(v >> l1) & ((1u << eb) - 1u));
            // the second look-up: a length's distance code -- or, behind a literal, the next symbol: two literals are one lane's work
            const u64 dv = v >> used1;
            const u32 *tab2 = is_len ? S->dlut + ((u32)dv & ((1u << DB) - 1u)) : S->lut + ((u32)dv & ((1u << LB) - 1u));
            const u32 e2 = *tab2;
            const u32 l2 = e2 & 15u, deb = (e2 >> 4) & 15u;
            const u32 distval = (e2 >> 16) + ((u32)(dv >> l2) & ((1u << deb) - 1u));
            const bool dist_ok = e2 != 0u && (e2 & K_BAD) != K_BAD;
            const bool pair = is_lit && (e2 & F_LIT) != 0u;
            const u32 used = is_len ? used1 + l2 + deb : (pair ? l1 + l2 : l1);
            const u32 olen = is_len ? lenval : (pair ? 2u : 1u);
            const u32 dist = is_len ? distval : 0u;
            const u32 lit2 = e2 >> 16;
            const bool stop = is_len ? !dist_ok : !is_lit;
            // Which lanes are symbols: the walk from lane 0.  A lane that is no symbol, or whose symbol ends behind the round, points at
            // itself, so the walk needs no test per step: v_readlane + a bit set, six steps between looks at whether it has arrived.
            const u32 nxt_v = lane + used;
            const bool term = stop || nxt_v > 63u;
            const u32 hop_v = term ? lane : nxt_v;
            const u64 stops = ballot(stop);
            u64 valid = 0ULL;
            u32 pos = 0u;
            for (;;) {
#pragma unroll
                for (int k = 0; k < 6; ++k) { valid |= 1ULL << pos; pos = rdlane(hop_v, pos); }
                if (rdlane(hop_v, pos) == pos) break;
            }
            valid |= 1ULL << pos;
            if ((stops >> pos) & 1ULL) valid &= ~(1ULL << pos);          // arrived at a lane that is no symbol: the scalar path's
            else pos = rdlane(nxt_v, pos);                               // ... at the round's last symbol: the next round starts behind it
            // their output: a byte of the batch for each literal, a queue slot for each match
            bool on = ((valid >> lane) & 1ULL) != 0ULL;
            u32 incl = wave_scan(on ? olen : 0u);
            const u64 mvalid = ballot(on && is_len);
            const u32 mrank = below(mvalid);
            const u64 cut = ballot(on && (D.so + incl > STAGE_CAP || (is_len && (D.nq + mrank >= 64u || dist > D.ob + D.so + incl - olen))));
            u32 produced;
            bool bad_dist = false;
            if (cut) {
                const u32 c = (u32)__builtin_ctzll(cut);             // the round ends in front of lane c's symbol
                bad_dist = rdlane((u32)(is_len && dist > D.ob + D.so + incl - olen), c) != 0u && rdlane((u32)(D.so + incl <= STAGE_CAP && D.nq + mrank < 64u), c) != 0u;
                valid &= (1ULL << c) - 1ULL;
                on = ((valid >> lane) & 1ULL) != 0ULL;
                produced = rdlane(incl - olen, c);
                pos = c;
            } else {
                produced = rdlane(incl, 63u);
            }
            if (on) {
                const u32 at = D.so + incl - olen;
                if (is_lit) {
                    S->stage[at] = (u8)(e1 >> 16);
                    if (olen == 2u) S->stage[at + 1u] = (u8)lit2;
                } else {
                    const u32 slot = D.nq + mrank;
                    S->q[2u * slot] = at | (olen << 16);
                    S->q[2u * slot + 1u] = dist;
                }
            }
            D.so += produced;
            D.nq += (u32)__builtin_popcountll(mvalid & valid);
            D.bp += pos;
            if (bad_dist) { D.status = INF_BAD_DISTANCE; break; }
            if (cut || pos >= 64u) continue;
            // the walk stopped at a symbol the lanes leave alone: a long code, a distance beyond its direct table, the end of the block
            if (D.so > FLUSH_AT || D.nq == 64u) continue;            // (room first: the loop's top)
            const u64 sb = ((u64)rdlane((u32)(v >> 32), pos) << 32) | rdlane((u32)v, pos);
            u64 b = sb;
            u32 e = rdlane(e1, pos), cl = e & 15u;
            if (e == 0u) { e = slow_entry(lane, b, lim_l, base_l, S->lit_rank, 287u, cl); if ((e & K_BAD) == K_BAD) { D.status = INF_BAD_CODE; break; } }
            b >>= cl;
            u32 took = cl;
            if (e & F_LIT) {                                      // (a literal with a code longer than the direct table's)
                if (lane == 0u) S->stage[D.so] = (u8)(e >> 16);
                ++D.so;
            } else {
                const u32 kind = e & K_BAD;
                if (kind == K_EOB) { D.bp += took; break; }
                if (kind != K_LEN) { D.status = INF_BAD_CODE; break; }
                const u32 eb = (e >> 4) & 15u;
                const u32 len = (e >> 16) + ((u32)b & ((1u << eb) - 1u));
                b >>= eb; took += eb;
                u32 d = uni(S->dlut[(u32)b & ((1u << DB) - 1u)]), dcl = d & 15u;
                if (d == 0u) d = slow_entry(lane, b, lim_d, base_d, S->dst_rank, 31u, dcl);
                if ((d & K_BAD) == K_BAD) { D.status = INF_BAD_CODE; break; }
                b >>= dcl; took += dcl;
                const u32 deb = (d >> 4) & 15u;
                const u32 mdist = (d >> 16) + ((u32)b & ((1u << deb) - 1u));
                took += deb;
                if (mdist > D.ob + D.so) { D.status = INF_BAD_DISTANCE; break; }
                if (lane == 0u) { S->q[2u * D.nq] = D.so | (len << 16); S->q[2u * D.nq + 1u] = mdist; }
                ++D.nq;
                D.so += len;
            }
            D.bp += took;
        }
    }
    if (STREAM && D.status == INF_OK && D.consumed() > in_len) D.status = INF_IN_OVERRUN;     // (the last block's end lies behind the input: not a whole block)
    if (STREAM && D.status == INF_OK && D.ob + D.so > out_len) D.status = INF_OUT_OVERFLOW;
    const bool whole = D.status == INF_OK;                       // (STREAM) the loop ended at a block boundary: the stop, or the member's end
    if (STREAM && whole) { blk_bp = D.bp; blk_out = D.ob + D.so; }
    {
        const u32 st = D.status;
        D.status = INF_OK;
        D.flush();                                               // (what was decoded before an error is still written: the CRC is of the bytes there)
        if (st != INF_OK) D.status = st;
    }
    if (STREAM) {
        io->end_bit = blk_bp - 8u * D.mis;
        io->member_end = (whole && last) ? 1u : 0u;
        *out_n = blk_out - io->prefix;
        return blk_out > io->prefix || whole ? INF_OK : D.status;
    }
    if (D.status == INF_OK && D.consumed() > in_len) D.status = INF_IN_OVERRUN;
    if (D.status == INF_OK && D.ob != out_len) D.status = INF_OUT_SHORT;
    *out_n = D.ob;
    return D.status;
}

// CRC-32 of out[0, got) by the whole wavefront: 64 slices, then zlib's combination.  tbl: 256 words of LDS (anything the decoder is done with).
__device__ __forceinline__ u32 crc32_wave(u32 *tbl, const u8 *out, u32 got)
{
    const u32 lane = lane_id();
    for (u32 i = lane; i < 256u; i += 64u) tbl[i] = bns_inf::crc32_entry(i);
    wave_sync();
    const u32 slice = (((got + 63u) >> 6) + 3u) & ~3u;
    const u32 a = min(got, lane * slice), b = min(got, (lane + 1u) * slice);
    u32 c = 0xFFFFFFFFu, i = a;
    for (; i + 4u <= b; i += 4u) {
        c ^= bns_inf::load32u(out + i);
        c = tbl[c & 0xFFu] ^ (c >> 8);
        c = tbl[c & 0xFFu] ^ (c >> 8);
        c = tbl[c & 0xFFu] ^ (c >> 8);
        c = tbl[c & 0xFFu] ^ (c >> 8);
    }
    for (; i < b; ++i) c = tbl[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
    c = b > a ? ~c : 0u;
    u32 v = c ? multmodp(x2nmodp(got - b, 3u), c) : 0u;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v ^= (u32)__shfl_xor((int)v, s, 64);
    return v;
}

}  // namespace bns_infw
