// bns_kernels.hip -- hand-written HIP kernels of the classify hot path for gfx950 (CDNA4, wave64).
//
//   pack_kernel       ASCII reads -> 2-bit words (MSB-first, 32 bases / u64) + N-mask (1 bit / base)
//   classify_kernel   per read (or mate pair), one wavefront: k-mer extract -> canonicalise -> table
//                     probe -> insertion-ordered vote -> resolve_tree            (classifier.h:212-251)
//   encode_kernel     Encoder::for_each over a batch                             (encoder.h:415-442)
//   probe_kernel      kh_get over a batch of keys                                (khash64.h:250-263)
//   rebucket_kernel   khash arrays -> 64-byte bucket layout (same key->value map)
//   build_*           update_lca_map on device                                   (feature_min.h:205-228)
//
// HBM-bound integer work: no MFMA.  The levers are one 64-byte sector per lookup (bucket layout, quad-
// cooperative loads), coalesced 2-bit read words, wave ballots for the vote, and an Euler-interval
// taxonomy so resolve_tree needs one 16-byte node fetch per distinct taxon instead of a parent walk.
#include "bns_device.hpp"
#include "bns_kernels.hpp"

namespace bns {

// Cold kernel arguments (output pointers, taxonomy, overflow list) are re-read from the kernarg segment at their point
// of use through a volatile constant-address-space view, so they do not occupy SGPRs across the hot loop: at 8 waves
// per SIMD the budget is 80 SGPRs and every spilled scalar costs v_writelane / v_readlane VALU slots.
typedef const volatile __attribute__((address_space(4))) ClassifyParams ColdParams;
__device__ __forceinline__ ColdParams *cold_params() { return (ColdParams *)__builtin_amdgcn_kernarg_segment_ptr(); }

// =====================================================================================================
// pack: one wavefront per read; each lane converts 4 ASCII bytes per pass (256 bases / pass).
// Word layout: base i of the read sits in word i>>5 at bits [62-2(i&31), 64-2(i&31)); the N-mask word has
// bit (31-(i&31)) set when base i is not A/C/G/T (alphabet.h:128: case-insensitive ACGT, everything else -1).
// Read r's words start at ((offsets[r] >> 5) + r): monotone, non-overlapping, needs no prefix scan.
// =====================================================================================================
__device__ __forceinline__ u32 base_code(u32 b, u32 &bad)
{
    const u32 c = b & 0xDFu;                                   // fold case (bit 5)
    const bool ok = (c == 0x41u) | (c == 0x43u) | (c == 0x47u) | (c == 0x54u);
    const u32 x = (c >> 1) & 3u;                               // A0 C1 G3 T2
    bad = ok ? 0u : 1u;
    return ok ? (x ^ (x >> 1)) : 0u;                           // A0 C1 G2 T3
}

__global__ __launch_bounds__(256) void pack_kernel(const u8 *__restrict__ bases, const u64 *__restrict__ offsets,
                                                   u64 n_reads, u64 *__restrict__ words, u32 *__restrict__ nmask)
{
    const int lane = lane_id();
    const u64 wave = (u64)blockIdx.x * (blockDim.x >> 6) + (u64)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 n_waves = (u64)gridDim.x * (blockDim.x >> 6);
    for (u64 r = wave; r < n_reads; r += n_waves) {
        const u64 o = offsets[r];
        const u64 L = offsets[r + 1] - o;
        const u64 wb = (o >> 5) + r;
        const u64 n_words = (L + 31) >> 5;
        for (u64 p = 0; p < (n_words << 5); p += 256) {
            const u64 bi = p + (u64)lane * 4;                  // first base of this lane
            u32 w = 0;
            if (bi < L) {
                const u64 addr = o + bi;
                const u32 mis = (u32)(addr & 3u);
                const u32 *ap = reinterpret_cast<const u32 *>(bases + (addr - mis));
                const u32 nbytes = (L - bi) < 4 ? (u32)(L - bi) : 4u;
                const u32 lo = ap[0];
                const u32 hi = (mis + nbytes > 4u) ? ap[1] : 0u;
                w = (u32)((((u64)hi << 32) | lo) >> (8u * mis));
            }
            u32 codes = 0, bads = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32 bad;
                const u32 cd = base_code((w >> (8 * i)) & 0xFFu, bad);
                if (bi + i >= L) bad = 1u;
                codes = (codes << 2) | (bad ? 0u : cd);
                bads = (bads << 1) | bad;
            }
            const int g = lane & 7;                            // position of this lane's 4 bases inside the word
            u32 hi32 = g < 4 ? codes << (24 - 8 * g) : 0u;
            u32 lo32 = g >= 4 ? codes << (24 - 8 * (g - 4)) : 0u;
            u32 nm = bads << (28 - 4 * g);
            hi32 |= dpp<QP_XOR1>(hi32); lo32 |= dpp<QP_XOR1>(lo32); nm |= dpp<QP_XOR1>(nm);
            hi32 |= dpp<QP_XOR2>(hi32); lo32 |= dpp<QP_XOR2>(lo32); nm |= dpp<QP_XOR2>(nm);
            hi32 |= (u32)__shfl_xor((int)hi32, 4); lo32 |= (u32)__shfl_xor((int)lo32, 4); nm |= (u32)__shfl_xor((int)nm, 4);
            const u64 wi = (p >> 5) + (u64)(lane >> 3);
            if (g == 0 && wi < n_words) {
                words[wb + wi] = ((u64)hi32 << 32) | lo32;
                nmask[wb + wi] = nm;
            }
        }
    }
}

// In-kernel pack (classify_kernel): the same conversion as pack_kernel, fused so a read's ASCII is the only thing
// fetched.  A chunk = 64 words = 2048 bases = up to 8 passes of 256 bases (4 per lane); pass 0 of a unit's first
// chunk is prefetched one unit ahead by the caller (raw_load) so its HBM latency hides behind the previous unit.
// Requirement on the caller's buffer: `bases` 4-byte aligned and readable up to the next 4-byte boundary past its end.
__device__ __forceinline__ void raw_load(const u8 *__restrict__ bases, u64 o, u32 L, u32 first_base, u32 &lo, u32 &hi)
{
    const u32 bi = first_base + (u32)lane_id() * 4u;
    lo = 0; hi = 0;
    if (bi < L) {
        const u64 addr = o + bi;
        const u32 mis = (u32)(addr & 3u);
        const u32 *ap = reinterpret_cast<const u32 *>(bases + (addr - mis));
        const u32 nbytes = (L - bi) < 4u ? (L - bi) : 4u;
        lo = ap[0];
        if (mis + nbytes > 4u) hi = ap[1];
    }
}

// Every lane drops its byte of 2-bit codes (and a byte of 2-bit N fields) straight into a per-wave LDS image of the
// chunk -- no cross-lane combine at all -- and k-mers are funnel-shifted out of two adjacent u64 words read back with
// one ds_read2_b64.
//   pk[0..64)   code words (MSB-first, 32 bases each);  pk[64..128) N words in the same geometry (11 = not A/C/G/T / past the end)
// Four ASCII bytes (little-endian dword, byte 0 = first base) -> 8 code bits + 8 N-field bits, all four at once: fold
// case, pick the expected letter for each byte's (b>>1)&3 with one v_perm_b32 (A,C,T,G sit at 0,1,2,3 of that hash),
// compare, and gather the per-byte 2-bit fields with one multiply each.  nvalid (0..4) = bytes that belong to the read.
__device__ __forceinline__ void swar_codes2(u32 w, u32 nvalid, u32 &codes8, u32 &mask8, u32 &bad8)
{
    const u32 x = w & 0xDFDFDFDFu;
    const u32 sel = (x >> 1) & 0x03030303u;
    const u32 expect = __builtin_amdgcn_perm(0u, 0x47544341u, sel);
    const u32 diff = expect ^ x;
    const u32 nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
    const u32 inv3 = (nz >> 7) * 3u;                                  // 0x03 per invalid byte
    const u32 c = (sel ^ ((sel >> 1) & 0x01010101u)) & ~inv3;
    const u32 tail = 0xFFu >> (2u * nvalid);
    codes8 = ((c * 0x40100401u) >> 24) & ~tail;
    const u32 inv8 = (inv3 * 0x40100401u) >> 24;
    mask8 = inv8 | tail;
    bad8 = inv8 & ~tail;                                              // non-A/C/G/T bytes INSIDE the read
}

// Returns true when every base of the read inside this chunk is A/C/G/T (wave-uniform).
// offv = the unit's offsets, one per lane (lanes 0..2); mate = which read of the unit.  The 64-bit offset is pulled out of
// offv where it is needed (passes after the first) rather than carried in SGPRs across the whole unit.
// ORD32 (contiguous seeds): the image is kept as 32-bit groups of 16 bases IN BASE ORDER behind one lead dword -- group q at dword
// 1 + q, its N fields at dword IMG_N32 + 1 + q -- so that a lane's 64-base window is two v_alignbit_b32 of three consecutive dwords
// (extract_lds32) instead of three 64-bit shifts (half rate) and two ORs.  Byte of the 4 bases q4: 4 + (q4 ^ 3).
constexpr u32 IMG_N32 = 130;                          // dword index of the N image's lead dword (code image: 1 + 128 dwords)
constexpr u32 IMG_U64 = 130;                          // u64 words per wave: 2 x (1 + 128) dwords, rounded up
template <bool ORD32 = false>
__device__ __forceinline__ bool pack_chunk_lds(const u8 *__restrict__ bases, u64 offv, int mate, u32 L, u32 j0, bool have0, u32 r_lo, u32 r_hi, u64 *pk)
{
    const int lane = lane_id();
    const u32 rem = L - j0;
    const u32 n_pass = rem >= 2048u ? 8u : (rem + 255u) >> 8;
    const u32 mis8 = 8u * ((readlane((u32)offv, mate) + j0) & 3u);
    u8 *pc = reinterpret_cast<u8 *>(pk) + (ORD32 ? 4 : 0), *pm = ORD32 ? reinterpret_cast<u8 *>(pk) + 4u * IMG_N32 + 4 : reinterpret_cast<u8 *>(pk + 64);
    u64 dirty = 0;
    auto convert = [&](u32 pass, u32 lo, u32 hi) {
        const u32 w = (u32)((((u64)hi << 32) | lo) >> mis8);
        const u32 bi = j0 + pass * 256u + (u32)lane * 4u;
        u32 codes, mask, bad;
        swar_codes2(w, bi < L ? (L - bi < 4u ? L - bi : 4u) : 0u, codes, mask, bad);
        dirty |= ballot64(bad != 0u);
        const u32 at = pass * 64u + ((u32)lane ^ (ORD32 ? 3u : 7u));  // byte 7 of a little-endian u64 (byte 3 of a dword) holds its first 4 bases
        pc[at] = (u8)codes;
        pm[at] = (u8)mask;
    };
    // pass 0 stands apart from the loop: with the prefetched dwords it needs no load, and sharing a join with the passes
    // that do would put a full vmcnt(0) -- a wait for the NEXT unit's prefetch -- in front of it
    if (n_pass) {
        u32 lo = r_lo, hi = r_hi;
        if (!have0) { raw_load(bases, readlane64(offv, mate), L, j0, lo, hi); asm volatile("" : "+v"(lo), "+v"(hi)); }   // (the wait for this load stays inside the branch)
        convert(0u, lo, hi);
    }
    for (u32 pass = 1; pass < n_pass; ++pass) {
        u32 lo, hi;
        raw_load(bases, readlane64(offv, mate), L, j0 + pass * 256u, lo, hi);
        convert(pass, lo, hi);
    }
    __builtin_amdgcn_wave_barrier();
    return dirty == 0;
}

// ---- packed input (bns_classify_batch_packed*): the reads arrive as the image itself -- 2-bit words, MSB-first, 32 bases per u64,
// read r's words starting at word (offsets[r] >> 5) + r, and (optionally) one u32 of invalid-base flags per word (bit 31 - i = base
// i of the word is not A/C/G/T).  A chunk of 2048 bases is then ONE coalesced 8-byte load per lane (64 lanes = 64 words) instead
// of eight passes of SWAR conversion: 40 bytes per 150-bp read over HBM (and PCIe) instead of 150.
// What is in flight for the unit after the current one: ASCII: the first 256 bases (4 per lane); packed: the first 64 words and
// their flag words.
struct Prefetch { u32 lo, hi, m; };
template <bool PACKED>
__device__ __forceinline__ void prefetch_read(const ClassifyParams &p, u64 r, u64 o, u32 L, Prefetch &f)
{
    f.lo = 0; f.hi = 0; f.m = 0;
    if (!PACKED) { raw_load(p.bases, o, L, 0u, f.lo, f.hi); return; }
    const u32 lane = (u32)lane_id();
    if (lane < ((L + 31u) >> 5)) {
        const u64 wi = (o >> 5) + r + lane;
        const u64 w = p.words[wi];
        f.lo = (u32)w; f.hi = (u32)(w >> 32);
        if (p.nmask) f.m = p.nmask[wi];
    }
}
// 32 one-bit flags -> 32 two-bit fields (11 = invalid), the geometry of the image's N words
__device__ __forceinline__ u64 mask1_to_mask2(u32 m)
{
    u64 x = m;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x * 3ULL;
}
// chunk starting at base j0 of read r (j0 is a multiple of 64) -> pk; true when no base of the read inside it is flagged
template <bool ORD32 = false>
__device__ __forceinline__ bool load_chunk_packed(const ClassifyParams &p, u64 r, u64 o, u32 L, u32 j0, bool have, const Prefetch &f, u64 *pk)
{
    const u32 lane = (u32)lane_id();
    u64 w = ((u64)f.hi << 32) | f.lo;
    u32 m = f.m;
    if (!have) {
        w = 0; m = 0;
        const u32 wl = (j0 >> 5) + lane;
        if (wl < ((L + 31u) >> 5)) {
            const u64 wi = (o >> 5) + r + wl;
            w = p.words[wi];
            if (p.nmask) m = p.nmask[wi];
        }
    }
    const bool dirty = ballot64(m != 0u) != 0ULL;
    if (ORD32) {
        u32 *img = reinterpret_cast<u32 *>(pk);
        img[1u + 2u * lane] = (u32)(w >> 32); img[2u + 2u * lane] = (u32)w;
        if (dirty) { const u64 m2 = mask1_to_mask2(m); img[IMG_N32 + 1u + 2u * lane] = (u32)(m2 >> 32); img[IMG_N32 + 2u + 2u * lane] = (u32)m2; }
    } else {
        pk[lane] = w;
        if (dirty) pk[64 + lane] = mask1_to_mask2(m);
    }
    __builtin_amdgcn_wave_barrier();
    return !dirty;
}

// win = the 64 bits (32 bases) of the chunk image starting at base rd*64 + lane, MSB-first.
// The same from the base-ordered 32-bit image (pack_chunk_lds<true> / load_chunk_packed<true>): the window that starts at base
// b = 64 rd + lane is taken from the three dwords that begin with the one holding base b - 1 (the lead dword for b = 0), so the
// shift n = 30 - 2 ((b + 15) & 15) is never 32: hi:lo = two v_alignbit_b32, full rate.  Dword index and shift are lane constants
// (64 rd is a multiple of 16).
__device__ __forceinline__ void extract_lds32(const u64 *pk, u32 rd, u32 k, bool clean, u64 &win, bool &valid)
{
    const u32 lane = (u32)lane_id();
    const u32 *img = reinterpret_cast<const u32 *>(pk);
    const u32 g = 4u * rd + ((lane + 15u) >> 4), n = 30u - 2u * ((lane + 15u) & 15u);
    const u32 w0 = img[g], w1 = img[g + 1u], w2 = img[g + 2u];
    const u32 hi = __builtin_amdgcn_alignbit(w0, w1, n), lo = __builtin_amdgcn_alignbit(w1, w2, n);
    win = ((u64)hi << 32) | lo;
    valid = true;
    if (!clean) {
        const u32 m0 = img[IMG_N32 + g], m1 = img[IMG_N32 + g + 1u], m2 = img[IMG_N32 + g + 2u];
        const u64 mw = ((u64)__builtin_amdgcn_alignbit(m0, m1, n) << 32) | __builtin_amdgcn_alignbit(m1, m2, n);
        valid = (mw >> (64u - 2u * k)) == 0;
    }
}

// 2-bit N fields -> 1 bit per base (only the spaced paths still want the compact form)
__device__ __forceinline__ u32 mask2_to_mask1(u64 m2)
{
    u64 x = m2 & 0x5555555555555555ULL;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
    return (u32)x;
}

// =====================================================================================================
// k-mer extraction from the wave-resident chunk: lane l holds word (chunk_word0 + l) in W and its N-mask in M.
// =====================================================================================================
// Contiguous seed: k-mer jl (relative to the chunk's first base) = 2k bits starting at bit 2*jl.
__device__ __forceinline__ bool extract_unspaced(u64 W, u32 M, u32 rd, u32 k, u64 &kmer)
{
    const int lane = lane_id();
    const int w0 = (int)(2 * rd);
    const u64 wa = readlane64(W, w0), wb = readlane64(W, w0 + 1), wc = readlane64(W, w0 + 2);
    const u32 ma = readlane(M, w0), mb = readlane(M, w0 + 1), mc = readlane(M, w0 + 2);
    const bool up = lane >= 32;
    const u64 hi = up ? wb : wa, lo = up ? wc : wb;
    const u64 m64 = up ? (((u64)mb << 32) | mc) : (((u64)ma << 32) | mb);
    const u32 o = (u32)lane & 31u;
    const u64 win = o ? ((hi << (2 * o)) | (lo >> (64 - 2 * o))) : hi;
    kmer = win >> (64u - 2u * k);
    return ((m64 << o) >> (64u - k)) == 0;                     // no non-ACGT base inside [jl, jl+k)
}

// The 16-entry window (k - m = 15: the wide window of a db of window minimizers, configs[1]) WITHOUT reading sixteen ring entries
// per lane: LDS instructions are what a round is shortest of after the 4-cycle VALU ones (tools/micro/valu_rate.hip: a
// ds_read2_b32 occupies the LDS pipe for 16 cycles, eight of them per round).  van Herk / Gil-Werman over blocks of 16 = the DPP
// rows: P = prefix minimum of the lane's own row up to the lane, S = suffix minimum from the lane to the end of its row (four
// v_min_u32 row_shr / row_shl steps each, no memory); the window of k-mer j is m-mer positions j .. j + 15 = the tail of one
// block and the head of the next = min(S of position j, P of position j + 15).  Position j + 15 IS lane j's own m-mer, so P is
// the lane's own value; S of position j sits fifteen lanes down -- the one thing that goes through the ring (one write, one read;
// positions 0 .. 14 are the previous round's last fifteen lanes, carried as before, or round 0's head m-mers).
template <int CTRL>
__device__ __forceinline__ u32 dpp_keep(u32 v)        // lanes without a source lane keep 0xFFFFFFFF (the identity of min)
{
    return (u32)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 row_prefix_min(u32 x)
{
    x = min(x, dpp_keep<0x111>(x)); x = min(x, dpp_keep<0x112>(x)); x = min(x, dpp_keep<0x114>(x)); x = min(x, dpp_keep<0x118>(x));   // row_shr:1,2,4,8
    return x;
}
__device__ __forceinline__ u32 row_suffix_min(u32 x)
{
    x = min(x, dpp_keep<0x101>(x)); x = min(x, dpp_keep<0x102>(x)); x = min(x, dpp_keep<0x104>(x)); x = min(x, dpp_keep<0x108>(x));   // row_shl:1,2,4,8
    return x;
}
// head = round 0 only, lanes 0 .. 14: the hash of the FIRST m-mer of k-mers 0 .. 14 (positions 0 .. 14); mine = the hash of the
// lane's own m-mer (position 15 + lane).  ring: 15 + 64 entries.
__device__ __forceinline__ u32 window16_min(u32 mine, u32 head, u32 rd, u32 *ring)
{
    const u32 lane = (u32)lane_id();
    if (rd == 0) {
        asm volatile("");                                     // (keeps this a scalar branch: see round_minhash)
        const u32 sh = row_suffix_min(lane < 15u ? head : 0xFFFFFFFFu);
        if (lane < 15u) ring[lane] = sh;
    }
    const u32 P = row_prefix_min(mine), S = row_suffix_min(mine);
    ring[15u + lane] = S;
    __builtin_amdgcn_wave_barrier();
    const u32 sprev = ring[lane];
    __builtin_amdgcn_wave_barrier();
    if (lane >= 49u) ring[lane - 49u] = S;                   // the round's last fifteen positions = the next round's first
    return min(P, sprev);
}

// Minimizer hash of every k-mer of round rd (contiguous seeds).  The m-mers of k-mer j sit at positions j..j+span
// (span = k-m).  Each lane hashes only the LAST m-mer of its own forward k-mer (position lane+span; its reverse
// complement is the top of the k-mer's reverse complement, so no extra bit reversal), the first `span` positions of a
// round are the previous round's tail (stored there by the lanes that hashed them), and the minimum over the (span+1)-wide window is read back
// from a per-wave LDS line.  Equals key_minhash(key): the canonical m-mer set of a k-mer and of its reverse
// complement coincide.  Garbage from N / past-the-end positions only reaches k-mers that are invalid anyway.
// W = entries of the unrolled window: span + 1 when the span is a compile-time constant, BNS_MAX_SPAN + 1 (tail masked) otherwise.
template <int W, bool VH = false>
__device__ __forceinline__ u32 round_minhash(u64 kf, u64 rc, u32 rd, u32 k, u32 m, u32 *ring)
{
    const int lane = lane_id();
    const u32 span = k - m;
    const u64 mmask = ~0ULL >> (64u - 2u * m);
    if (span == 0) { const u64 a = kf & mmask, b = rc & mmask; return mmer_hash(a < b ? a : b); }
    u32 mine;
    if (VH) {                                                // (the fixed-k instantiations with the wide window: W == 16, span == 15)
        u32 head = 0;
        if (m <= 16u) {
            const u32 mm = 0xFFFFFFFFu >> (32u - 2u * m);
            if (rd == 0) head = mmer_mix(min((u32)(kf >> 30) & mm, (u32)rc & mm));
            mine = mmer_mix(min((u32)kf & mm, (u32)(rc >> 30)));
        } else {
            if (rd == 0) { const u64 a = kf >> 30, b = rc & mmask; head = mmer_hash(a < b ? a : b); }
            const u64 a = kf & mmask, b = rc >> 30;
            mine = mmer_hash(a < b ? a : b);
        }
        return window16_min(mine, head, rd, ring);
    }
    if (m <= 16u) {
        // m-mers that fit a word: the canonical m-mer is a v_min_u32 of two words (the low word of the k-mer, the top of its
        // reverse complement brought down by one v_alignbit), and mmer_hash of a value below 2^32 is mmer_mix of it
        const u32 mm = 0xFFFFFFFFu >> (32u - 2u * m);
        // (a scalar branch around the lane test: merged into one lane mask the compiler issues the three instructions, exec = 0, in
        // every later round too)
        if (rd == 0) { asm volatile(""); if ((u32)lane < span) ring[lane] = mmer_mix(min((u32)(kf >> (2u * span)) & mm, (u32)rc & mm)); }
        mine = mmer_mix(min((u32)kf & mm, (u32)(rc >> (2u * span))));
        ring[span + (u32)lane] = mine;
    } else {
        if (rd == 0) {                                       // positions 0..span-1: FIRST m-mer of k-mers 0..span-1
            if ((u32)lane < span) {
                const u64 a = kf >> (2u * span), b = rc & mmask;
                ring[lane] = mmer_hash(a < b ? a : b);
            }
        }                                                    // (later rounds: carried over at the end of the previous one)
        const u64 a = kf & mmask, b = rc >> (2u * span);
        mine = mmer_hash(a < b ? a : b);
        ring[span + (u32)lane] = mine;
    }
    __builtin_amdgcn_wave_barrier();
    // span <= W - 1: read the window back to back (no loop, no waits in between) -- a wide one in two halves, so that it does not
    // hold fifteen registers at once -- and mask the tail with the wave-uniform span where that is not a constant
    constexpr u32 G = W <= 9 ? (u32)W : ((u32)W + 1u) / 2u;
    u32 best = 0xFFFFFFFFu;
#pragma unroll
    for (u32 g = 0; g < (u32)W; g += G) {
        u32 h[G + 1];
#pragma unroll
        for (u32 i = 0; i <= G; ++i) h[i] = (i < G && g + i < (u32)W) ? ring[(u32)lane + g + i] : 0xFFFFFFFFu;
        if (span == (u32)W - 1u) {                               // the full window: v_min3_u32 pairs
#pragma unroll
            for (u32 i = 0; i < G; i += 2) if (g + i < (u32)W) best = min(min(best, h[i]), h[i + 1]);
        } else {
#pragma unroll
            for (u32 i = 0; i < G; ++i) { const u32 x = g + i <= span ? h[i] : 0xFFFFFFFFu; best = x < best ? x : best; }
        }
        if (g + G < (u32)W) { asm volatile("" : "+v"(best)); __builtin_amdgcn_sched_barrier(0); }     // (the first half's minimum is formed before the second half is read)
    }
    __builtin_amdgcn_wave_barrier();
    if ((u32)lane >= 64u - span) ring[(u32)lane + span - 64u] = mine;     // the round's tail = the next round's first `span` positions
    return best;
}

// The same with the WIDE minimizer identity (bns_device.hpp, wide_ident*): the ring holds 64-bit entries -- positive doubles of one
// exponent, mantissa = hash : m-mer -- and the window minimum is one v_min_f64 per entry.  Returns the winning identity; the
// bucket comes from wide_bucket_in() of it.  Equals the sp.wide branch of key_minhash(key, k, MinSpec).
template <int W>
__device__ __forceinline__ u64 round_minhash_wide(u64 kf, u64 rc, u32 rd, u32 k, u32 m, u64 *ring)
{
    const int lane = lane_id();
    const u32 span = k - m;
    const u64 mmask = ~0ULL >> (64u - 2u * m);
    u64 mine;
    if (m <= 16u) {
        const u32 mm = 0xFFFFFFFFu >> (32u - 2u * m);
        if (span && rd == 0 && (u32)lane < span) ring[lane] = wide_ident32(min((u32)(kf >> (2u * span)) & mm, (u32)rc & mm));
        mine = wide_ident32(min((u32)kf & mm, (u32)(rc >> (2u * span)) & mm));
    } else {
        if (span && rd == 0 && (u32)lane < span) { const u64 a = kf >> (2u * span), b = rc & mmask; ring[lane] = wide_ident64(a < b ? a : b); }
        const u64 a = kf & mmask, b = rc >> (2u * span);
        mine = wide_ident64(a < b ? a : b);
    }
    if (span == 0) return mine;
    ring[span + (u32)lane] = mine;
    __builtin_amdgcn_wave_barrier();
    constexpr u32 G = 4;                                         // entries in flight: 8 registers (a whole window of 64-bit entries is 18-32)
    u64 best = mine;                                             // (the window's last entry is the lane's own)
#pragma unroll
    for (u32 g = 0; g < (u32)W; g += G) {
        u64 h[G];
#pragma unroll
        for (u32 i = 0; i < G; ++i) h[i] = (g + i < (u32)W) ? ring[(u32)lane + g + i] : mine;
#pragma unroll
        for (u32 i = 0; i < G; ++i) if (g + i < (u32)W) best = wide_min(best, (span == (u32)W - 1u || g + i <= span) ? h[i] : mine);
        if (g + G < (u32)W) { asm volatile("" : "+v"(best)); __builtin_amdgcn_sched_barrier(0); }
    }
    __builtin_amdgcn_wave_barrier();
    if ((u32)lane >= 64u - span) ring[(u32)lane + span - 64u] = mine;
    return best;
}

// Spaced seed: gather k bases at cumulative offsets pos[i] (encoder.h:547-592 kmer()); only the sampled
// positions must be A/C/G/T.
__device__ __forceinline__ bool extract_spaced(u64 W, u32 M, u32 rd, u32 k, u32 rdesc, u64 &kmer)
{
    const u32 jl = rd * 64u + (u32)lane_id();
    u64 km = 0;
    u32 bad = 0;
    for (u32 i = 0; i < k; ++i) {
        const u32 p = jl + (readlane(rdesc, (int)i) >> 16);                  // pos[i], see run_desc()
        const int wi = (int)(p >> 5) & 63;
        const u32 o = p & 31u;
        const u64 w = ((u64)(u32)__shfl((int)(W >> 32), wi) << 32) | (u32)__shfl((int)(u32)W, wi);
        const u32 m = (u32)__shfl((int)M, wi);
        km = (km << 2) | ((w >> (62u - 2u * o)) & 3u);
        bad |= (m >> (31u - o)) & 1u;
    }
    kmer = km;
    // the reference's spaced loop drops a k-mer that EQUALS its overflow marker ~0 (encoder.h:236-238): with k = 32 that is
    // the genuine all-T k-mer
    return bad == 0 && km != ~0ULL;
}

// Spaced seed, comb <= 64: build the 64-base window aligned at the k-mer's first base (two funnel shifts of the
// wave-resident words, exactly as the contiguous path) and gather the sampled bases run by run with UNIFORM shifts.
// rdesc: lane r holds run r as start | len << 8, and pos[r] << 16 for the generic gather (run_desc(), read once per
// kernel: the parameter block's byte arrays would cost a memory round trip per run per round).
__device__ __forceinline__ u32 run_desc(const ClassifyParams &p)
{
    const u32 l = (u32)lane_id() & 31u;
    return (u32)p.run_start[l] | ((u32)p.run_len[l] << 8) | ((u32)p.pos[l] << 16);
}
// clean (wave-uniform) = no base of the read inside this chunk is invalid: the N window is neither built nor tested (k-mers that
// reach past the end of the read are excluded by the caller's position test).
__device__ __forceinline__ bool spaced_gather(u64 A0, u64 A1, u64 mwin, const ClassifyParams &p, u32 rdesc, u64 &kmer);
__device__ __forceinline__ bool extract_spaced_runs(u64 W, u32 M, u32 rd, const ClassifyParams &p, u32 rdesc, u64 &kmer, bool clean = false)
{
    const int lane = lane_id();
    const int w0 = (int)(2 * rd);
    const u64 wa = readlane64(W, w0), wb = readlane64(W, w0 + 1), wc = readlane64(W, w0 + 2), wd = readlane64(W, (w0 + 3) & 63);
    const bool up = lane >= 32;
    const u64 x0 = up ? wb : wa, x1 = up ? wc : wb, x2 = up ? wd : wc;
    const u32 o = (u32)lane & 31u;
    const u64 A0 = o ? ((x0 << (2 * o)) | (x1 >> (64 - 2 * o))) : x0;      // bases j .. j+31
    const u64 A1 = o ? ((x1 << (2 * o)) | (x2 >> (64 - 2 * o))) : x1;      // bases j+32 .. j+63
    u64 mwin = 0;
    if (!clean) {
        const u32 ma = readlane(M, w0), mb = readlane(M, w0 + 1), mc = readlane(M, w0 + 2), md = readlane(M, (w0 + 3) & 63);
        const u64 m01 = up ? (((u64)mb << 32) | mc) : (((u64)ma << 32) | mb);
        const u32 m2 = up ? md : mc;
        mwin = o ? ((m01 << o) | ((u64)m2 >> (32 - o))) : m01;            // N flags of bases j .. j+63 (bit 63 = base j)
    }
    return spaced_gather(A0, A1, mwin, p, rdesc, kmer);
}
// The same from the per-wave LDS image of the chunk (classify): every lane reads its three words itself -- no scalar
// broadcasts, no selects -- and a clean read needs no N window at all.  pk[0..64) code words, pk[64..128) 2-bit N fields.
__device__ __forceinline__ bool extract_spaced_lds(const u64 *pk, u32 rd, const ClassifyParams &p, u32 rdesc, u64 &kmer, bool clean)
{
    const int lane = lane_id();
    const u32 wi = 2u * rd + ((u32)lane >> 5);
    const u32 s = 2u * ((u32)lane & 31u);
    const u64 x0 = pk[wi], x1 = pk[(wi + 1u) & 63u], x2 = pk[(wi + 2u) & 63u];
    const u64 A0 = (x0 << s) | ((x1 >> 1) >> (63u - s));
    const u64 A1 = (x1 << s) | ((x2 >> 1) >> (63u - s));
    u64 mwin = 0;
    if (!clean) {
        const u64 m0 = pk[64u + wi], m1 = pk[64u + ((wi + 1u) & 63u)], m2 = pk[64u + ((wi + 2u) & 63u)];
        const u64 n0 = (m0 << s) | ((m1 >> 1) >> (63u - s)), n1 = (m1 << s) | ((m2 >> 1) >> (63u - s));
        mwin = ((u64)mask2_to_mask1(n0) << 32) | mask2_to_mask1(n1);      // 2-bit N fields -> one flag per base, bit 63 = base j
    }
    return spaced_gather(A0, A1, mwin, p, rdesc, kmer);
}
__device__ __forceinline__ bool spaced_gather(u64 A0, u64 A1, u64 mwin, const ClassifyParams &p, u32 rdesc, u64 &kmer)
{
    if (p.pext_on) {
        // compress the sampled 2-bit fields of each half towards its low end (order preserved), then concatenate; a half whose
        // sampled bases are its first n (pext_top) is a plain shift
        u64 x0 = A0 & p.pext_mask[0], x1 = A1 & p.pext_mask[1];
        if (p.pext_top[0] != 0xFFu) x0 = p.pext_top[0] ? x0 >> (64u - 2u * p.pext_top[0]) : 0ULL;
        if (p.pext_top[1] != 0xFFu) x1 = p.pext_top[1] ? x1 >> (64u - 2u * p.pext_top[1]) : 0ULL;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (p.pext_steps[0] & (1u << i)) { const u64 t = x0 & p.pext_mv[0][i]; x0 = (x0 ^ t) | (t >> (1u << i)); }
            if (p.pext_steps[1] & (1u << i)) { const u64 t = x1 & p.pext_mv[1][i]; x1 = (x1 ^ t) | (t >> (1u << i)); }
        }
        const u64 kmp = p.pext_n1 >= 32u ? x1 : ((x0 << (2u * p.pext_n1)) | x1);
        kmer = kmp;
        return (mwin & p.sample_mask) == 0 && kmp != ~0ULL;
    }
    u64 km = 0;
    for (u32 r = 0; r < p.n_runs; ++r) {
        const u32 d = readlane(rdesc, (int)r);
        const u32 s = d & 0xFFu, len = (d >> 8) & 0xFFu;                   // uniform
        u64 x;                                                             // bases s.. left-aligned
        if (s == 0) x = A0;
        else if (s < 32) x = (A0 << (2 * s)) | (A1 >> (64 - 2 * s));
        else if (s == 32) x = A1;
        else x = A1 << (2 * (s - 32));
        km = (len == 32 ? 0ULL : (km << (2 * len))) | (x >> (64 - 2 * len));
    }
    kmer = km;
    return (mwin & p.sample_mask) == 0 && km != ~0ULL;          // (same rule as extract_spaced)
}

// =====================================================================================================
// Windowed minimizers (Encoder::for_each_canon_windowed, encoder.h:211-217 + qmap.h:79-87): window i holds the
// ws = w-c+1 canonical k-mers i..i+ws-1; its value is the k-mer with the smallest (score, kmer).
// =====================================================================================================
__device__ __forceinline__ u64 kmer_score(u64 el, int kind)
{
    if (kind == 1) {                                             // score::Entropy through the path overloads
        const double x = (double)el / (-1.0 + 1e-4);            // CircusEnt::value() is NOT_FULL = -1 (entropy.h:44-48)
        // x86 cvttsd2si path of the reference's double -> u64; k = 32 k-mers above 2^63 / 0.9999 fall below -2^63, where
        // cvttsd2si returns the "integer indefinite" 0x8000000000000000 (a GPU conversion would saturate or wrap differently)
        if (!(x > -9223372036854775808.0)) return 0x8000000000000000ULL;
        return (u64)(long long)x;
    }
    u64 h = el ^ 0x533f8c2151b20f97ULL;                          // score::Lex = FRev64 (encoder.h:47), restated; parity unpinned (F9)
    h *= 0x9a98567ed20c127dULL;
    h = (h << 31) | (h >> 33);
    return h ^ 0x691a9d706391077aULL;
}

// score::Entropy through the STRING overload (encoder.h:307-346): double(fwd k-mer) / (entropy of its k bases + .001), with
// the entropy terms (n/k) ln(n/k) taken from a host-computed table (the host's libm, so GPU and CPU agree to the bit) and
// added in the order A, C, G, T (the reference's order is a hash map's: parity unpinned to the last ulp), and the
// double -> u64 conversion of gcc's x86-64 code (cvttsd2si below 2^63, cvttsd2si(x - 2^63) ^ 2^63 above).
__device__ __forceinline__ u64 dbl_to_u64_x86(double x)
{
    const double two63 = 9223372036854775808.0;
    if (x != x) return 0x8000000000000000ULL;
    if (x < two63) return x >= -two63 ? (u64)(long long)x : 0x8000000000000000ULL;
    const double y = x - two63;
    return (y < two63 ? (u64)(long long)y : 0x8000000000000000ULL) ^ 0x8000000000000000ULL;
}
__device__ __forceinline__ u64 kmer_score_entropy(u64 km, u32 k, const double *tbl)
{
    const u64 lo = km & 0x5555555555555555ULL, hi = (km >> 1) & 0x5555555555555555ULL;
    const u32 nT = (u32)__popcll(lo & hi), nG = (u32)__popcll(hi & ~lo), nC = (u32)__popcll(lo & ~hi), nA = k - nT - nG - nC;
    double v = 0.;
    if (nA) v = v + tbl[nA];
    if (nC) v = v + tbl[nC];
    if (nG) v = v + tbl[nG];
    if (nT) v = v + tbl[nT];
    return dbl_to_u64_x86((double)km / (v + .001));
}

// One round of 64 window starts (64*rd + lane, relative to the chunk).  lds = 256 u64 per wave.
// SPACED: the path overloads' for_each_uncanon_spaced -> next_minimizer (encoder.h:233-239,615-620): the raw spaced k-mer,
// ENCODE_OVERFLOW (~0) for one with a non-ACGT sampled base, no canonicalisation; the caller drops a window whose
// minimum is ~0.
template <bool SPACED>
__device__ __forceinline__ u64 windowed_round(u64 W, u32 M, u32 rd, const ClassifyParams &p, u32 rdesc, u64 *lds)
{
    const int lane = lane_id();
    const u32 ws = p.w - p.c + 1u;
    u64 *l_el = lds, *l_sc = lds + 128;
    auto entry = [&](u32 round) -> u64 {                         // the queue entry of position 64*round + lane
        u64 km;
        if (SPACED) {
            // (extract_spaced* already report the all-T 32-mer as invalid: it IS the overflow value)
            const bool ok = p.n_runs ? extract_spaced_runs(W, M, round, p, rdesc, km) : extract_spaced(W, M, round, p.k, rdesc, km);
            return ok ? km : ~0ULL;
        }
        const bool ok = extract_unspaced(W, M, round, p.k, km);
        // a k-mer with a non-ACGT base is ENCODE_OVERFLOW, which canonical_representation() maps to 0 (encoder.h:624-625)
        return ok ? canonical(km, p.k) : 0ULL;
    };
    if (ws > 64u) {
        // wide windows (w - c + 1 up to 1024): 64 entries at a time through LDS, every lane folding the ones inside its own
        // window [lane, lane + ws) into a running minimum -- O(ws + 64) per round instead of the one-shot 128-entry image
        u64 bs = ~0ULL, be = ~0ULL;
        bool have = false;
        const u32 n_seg = (ws + 126u) >> 6;                       // segments holding entries 0 .. 63 + ws - 1
        for (u32 h = 0; h < n_seg; ++h) {
            const u64 el = entry(rd + h);
            l_el[lane] = el;
            l_sc[lane] = kmer_score(el, p.score);
            __builtin_amdgcn_wave_barrier();
            for (u32 t = 0; t < 64u; ++t) {
                const u32 e = h * 64u + t;
                const u64 sc = l_sc[t], x = l_el[t];
                const bool in = e >= (u32)lane && e < (u32)lane + ws;
                const bool lt = in && (!have || sc < bs || (sc == bs && x < be));
                bs = lt ? sc : bs; be = lt ? x : be; have = have || in;
            }
            __builtin_amdgcn_wave_barrier();
        }
        return be;
    }
#pragma unroll
    for (u32 half = 0; half < 2; ++half) {
        const u64 el = entry(rd + half);
        l_el[half * 64 + (u32)lane] = el;
        l_sc[half * 64 + (u32)lane] = kmer_score(el, p.score);
    }
    __builtin_amdgcn_wave_barrier();
    u64 bs = l_sc[lane], be = l_el[lane];
    for (u32 i = 1; i < ws; ++i) {
        const u64 s = l_sc[(u32)lane + i], e = l_el[(u32)lane + i];
        const bool lt = s < bs || (s == bs && e < be);           // ElScore::operator< (qmap.h:22-24)
        bs = lt ? s : bs; be = lt ? e : be;
    }
    __builtin_amdgcn_wave_barrier();
    return be;
}

// Encoder::for_each_uncanon_unspaced_windowed (encoder.h:273-306) over sequence r by one wavefront: the contiguous,
// NOT canonicalised windowed stream (`bonsai build -C -w`, python from_str(canon=False, w > k)).  Unlike the canonical
// variant its windows run over the stream of EMITTED k-mers -- a non-ACGT base restarts the k-mer but not the queue, so a
// window reaches across an N gap -- and a sequence that never fills a window flushes the minimum of what it has (:304-305).
// Per round of 64 positions the valid forward k-mers are compacted (ballot rank) behind the ws-1 entries carried from the
// round before in `lds` (2 x 128 u64: element, score), every full window takes its (score, k-mer) minimum, and the last
// ws-1 entries move to the front.
// The reference ORs the base into the k-mer before testing for ENCODE_OVERFLOW (:286-287), so with k >= 31 a valid T that
// completes 64 one-bits restarts the k-mer like an invalid base: the 32nd, 64th, ... T of a T run.  Those positions are
// found per chunk from the words' T masks (a run's phase enters a word from the nearest preceding word that is not all T,
// or from the chunk before) and are simply added to the invalid-base mask.
// emit(kmer, on) is called wave-wide with `on` set in the lanes that carry an output, in stream order.
// WIDE (ws > 64): the queue image does not fit the 128-entry LDS buffer; `lds` is then this wavefront's slice of a global
// scratch buffer (2 x (ws + 64) u64, L2-resident), the code is the same with fences instead of LDS ordering.
template <bool WIDE, class Emit>
__device__ __forceinline__ void uncanon_windowed_seq(const ClassifyParams &p, u64 r, u64 *lds, Emit emit)
{
    const int lane = lane_id();
    const u32 k = p.k, ws = p.w - k + 1u;
    auto sync = [] { if (WIDE) __threadfence_block(); __builtin_amdgcn_wave_barrier(); };
    const u64 o = p.offsets[r];
    const u64 Lg = p.offsets[r + 1] - o;
    if (Lg < k) return;
    const u64 wb = (o >> 5) + r, n_words = (Lg + 31u) >> 5, nk = Lg - k + 1u;
    const u32 rounds_per_chunk = (2048u - (k - 1u)) / 64u;
    const u64 chunk_k = (u64)rounds_per_chunk * 64u;
    u64 *l_el = lds, *l_sc = lds + (WIDE ? ws + 64u : 128u);
    u32 carry = 0, tcarry = 0;
    bool filled_once = false;
    for (u64 j0 = 0; j0 < nk; j0 += chunk_k) {
        const u64 wi = (j0 >> 5) + (u64)lane;
        const u64 W = wi < n_words ? p.words[wb + wi] : 0ULL;
        u32 M = wi < n_words ? p.nmask[wb + wi] : 0xFFFFFFFFu;
        if (k >= 31u && p.score != 2) {                                        // (the real-entropy variant tests the base itself, encoder.h:327)
            const u32 tm = mask2_to_mask1(W & (W >> 1)) & ~M;                     // bit 31-q: base q of the word is a valid T
            const bool full = tm == 0xFFFFFFFFu;
            const u32 trail = full ? 32u : (u32)__builtin_ctz(~tm), lead = full ? 32u : (u32)__builtin_clz(~tm);
            const u64 before = ballot64(!full) & lanemask_lt();                     // preceding words that break a T run
            const int h = before ? 63 - __builtin_clzll(before) : 0;
            const u32 th = (u32)__builtin_amdgcn_ds_bpermute(h << 2, (int)trail);
            const u32 din = (before ? th : tcarry) & 31u;                           // T-run length entering this word, mod 32
            if (31u - din < lead) M |= 1u << din;                                   // base 31-din is the run's 32nd / 64th / ... T
            tcarry = readlane(din, (int)(rounds_per_chunk * 2u));                   // the next chunk's first word
        }
        const u32 chunk_nk = (nk - j0) < chunk_k ? (u32)(nk - j0) : (u32)chunk_k;
        for (u32 rd = 0; rd * 64u < chunk_nk; ++rd) {
            u64 km;
            const bool valid = extract_unspaced(W, M, rd, k, km) && rd * 64u + (u32)lane < chunk_nk;
            const u64 vm = ballot64(valid);
            if (valid) {
                const u32 idx = carry + (u32)__popcll(vm & lanemask_lt());
                l_el[idx] = km;
                l_sc[idx] = p.score == 2 ? kmer_score_entropy(km, k, p.ent_tbl) : kmer_score(km, p.score);
            }
            sync();
            const u32 nb = carry + (u32)__popcll(vm);
            const u32 n_out = nb >= ws ? nb - ws + 1u : 0u;
            u64 bs = l_sc[lane], be = l_el[lane];
            for (u32 i = 1; i < ws; ++i) {
                const u64 sc = l_sc[(u32)lane + i], e = l_el[(u32)lane + i];
                const bool lt = sc < bs || (sc == bs && e < be);                    // ElScore::operator< (qmap.h:22-24)
                bs = lt ? sc : bs; be = lt ? e : be;
            }
            const u32 nc = nb < ws - 1u ? nb : ws - 1u;                             // entries the next round still needs
            if (nb > nc)                                                            // (else they already sit at the front)
                for (u32 base = 0; base < nc; base += 64u) {                        // forward move, 64 at a time: sources stay ahead of writes
                    const u32 i = base + (u32)lane, src = nb - nc + i;
                    const u64 ce = i < nc ? l_el[src] : 0ULL, cs = i < nc ? l_sc[src] : 0ULL;
                    sync();
                    if (i < nc) { l_el[i] = ce; l_sc[i] = cs; }
                    sync();
                }
            carry = nc;
            filled_once = filled_once || n_out != 0u;
            // selection ran on forward k-mers; the real-entropy variant canonicalises what it emits (encoder.h:347-353).
            // (~0 is never selected: the all-T 32-mer is a restart above; in the entropy variant it scores 0x8000..., and the
            // reference drops it when it wins -- the `on` mask below)
            emit(p.canon ? canonical(be, k) : be, (u32)lane < n_out && be != ~0ULL);
        }
    }
    if (carry && !filled_once) {                        // qmap_.partially_full(): one value, the minimum of the queue
        u64 bs = l_sc[0], be = l_el[0];
        for (u32 i = 1; i < carry; ++i) {
            const u64 sc = l_sc[i], e = l_el[i];
            const bool lt = sc < bs || (sc == bs && e < be);
            bs = lt ? sc : bs; be = lt ? e : be;
        }
        sync();
        emit(p.canon ? canonical(be, k) : be, lane == 0);
    }
}

// =====================================================================================================
// Insertion-ordered counter (linear::counter<tax_t,u16>, linear.h:181-264) kept per wavefront in `keys`/`cnt`
// (LDS in the fast path, global scratch in the overflow path).  All arguments are wave-uniform.
// =====================================================================================================
__device__ __forceinline__ bool counter_add(u32 *keys, u32 *cnt, u32 cap, u32 &D, u32 t, u32 c)
{
    const int lane = lane_id();
    int idx = -1;
    for (u32 base = 0; base < D; base += 64) {
        const u32 i = base + (u32)lane;
        const u64 bm = ballot64(i < D && keys[i] == t);
        if (bm) { idx = (int)base + __builtin_ctzll(bm); break; }
    }
    if (idx >= 0) {
        if (lane == 0) cnt[idx] += c;
    } else {
        if (D >= cap) return false;
        if (lane == 0) { keys[D] = t; cnt[D] = c; }
        ++D;
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// The same counter with its first 64 entries in registers (lane i holds entry i: ckey, ccnt) and only entries 64.. in
// `keys`/`cnt` -- a read rarely hits more than a few taxa, and this way a vote costs a compare, a ballot and an add
// instead of an LDS round trip.  Lanes >= D hold stale keys: the match mask is cut at D (dmask), and a stale lane's count
// is overwritten when it becomes an entry.
// The vote loop tests the registers itself (ballot(ckey == t) & dmask, dmask = the lanes below min(D, 64)); this is the
// miss path: a new entry, in a register while there are fewer than 64, else through the LDS/global arrays.
__device__ __forceinline__ bool counter_insert(u32 &ckey, u32 &ccnt, u64 &dmask, u32 *keys, u32 *cnt, u32 cap, u32 &D, u32 t, u32 c)
{
    if (D < 64u) {
        const bool me = (u32)lane_id() == D;
        ckey = me ? t : ckey; ccnt = me ? c : ccnt;
        ++D; dmask = (dmask << 1) | 1ULL;
        return true;
    }
    u32 D2 = D - 64u;
    const bool ok = cap > 64u && counter_add(keys + 64, cnt + 64, cap - 64u, D2, t, c);
    D = D2 + 64u;
    return ok;
}

// score(i) = sum over ancestors-or-self a of keys[i] of count(a)  (util.h:841-844), counts are u16.
// Ancestor test through Euler intervals; taxon 0 has an empty chain.
__device__ __forceinline__ u32 score_of(const u32 *keys, const u32 *cnt, const u32 *tin, const u32 *tout, u32 D, u32 i)
{
    const u32 td = keys[i], tind = tin[i];
    u32 s = 0;
    for (u32 j = 0; j < D; ++j) {
        const u32 tj = keys[j];
        const bool anc = (tj == td) ? (td != 0u) : (tin[j] < tind && tind < tout[j]);
        s += anc ? (cnt[j] & 0xFFFFu) : 0u;
    }
    return s;
}

// resolve_tree (util.h:831-869) over the wave's counter.  Returns the taxon (wave-uniform).
// lane_ = the caller's lane id (classify_unit hands in a copy the compiler cannot trace back to threadIdx: see there)
__device__ __forceinline__ u32 resolve_wave(const u32 *keys, const u32 *cnt, u32 *tin, u32 *tout, u32 D,
                                            const TaxNode *__restrict__ nodes, u32 n_nodes, int lane_ = -1)
{
    const int lane = lane_ >= 0 ? lane_ : lane_id();
    if (D == 0) return 0u;
    if (D == 1) return keys[0];      // a lone taxon wins whatever its score (a zero score ties with the initial 0: lca(0,t)=t)
    for (u32 i = (u32)lane; i < D; i += 64) {
        const TaxNode nd = load_node(nodes, n_nodes, keys[i]);
        tin[i] = nd.tin; tout[i] = nd.tout;
    }
    __builtin_amdgcn_wave_barrier();
    // pass A: maximum score
    u32 best = 0;
    for (u32 base = 0; base < D; base += 64) {
        const u32 i = base + (u32)lane;
        const u32 s = i < D ? score_of(keys, cnt, tin, tout, D, i) : 0u;
        best = s > best ? s : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u32 o = (u32)__shfl_xor((int)best, off);
        best = o > best ? o : best;
    }
    // pass B: fold lca over the tied taxa in insertion order (util.h:846-860).  When best == 0 the
    // reference's set also holds the initial max_taxon 0, which lca() treats as neutral.
    u32 acc = 0;
    bool first = true;
    for (u32 base = 0; base < D; base += 64) {
        const u32 i = base + (u32)lane;
        const u32 s = i < D ? score_of(keys, cnt, tin, tout, D, i) : 0u;
        u64 tie = ballot64(i < D && s == best);
        while (tie) {
            const int l = __builtin_ctzll(tie);
            tie &= tie - 1;
            const u32 t = keys[base + (u32)l];
            if (first) { acc = t; first = false; }
            else acc = lca_dev(nodes, n_nodes, acc, t);
        }
    }
    return acc;
}

// resolve_tree for a counter of at most 64 entries held in registers (lane i = entry i, as the vote leaves it): no LDS, no
// barriers -- every lane fetches its own taxon's node (one 16-byte load), the D x D ancestor test walks the entries through
// v_readlane, the maximum and the tie fold run over scalar loops.  Same arithmetic as score_of / resolve_wave: u16 counts, a
// zero taxon has an empty chain, ties folded with lca() in insertion order.  With D = 2..4 (a read that touches a leaf and the
// LCAs above it) this is a few dozen instructions and ONE memory round trip instead of the LDS path's dozen round trips.
__device__ __forceinline__ u32 resolve_regs(u32 ckey, u32 ccnt, u32 D, const TaxNode *__restrict__ nodes, u32 n_nodes)
{
    const u32 lane = (u32)lane_id();
    const bool in = lane < D;
    u32 mytin = 0, mytout = 0;
    if (in) { const TaxNode nd = load_node(nodes, n_nodes, ckey); mytin = nd.tin; mytout = nd.tout; }
    u32 s = 0;
    for (u32 j = 0; j < D; ++j) {                                 // (wave-uniform)
        const u32 tj = readlane(ckey, (int)j), tinj = readlane(mytin, (int)j), toutj = readlane(mytout, (int)j);
        const u32 cj = readlane(ccnt, (int)j) & 0xFFFFu;
        const bool anc = (tj == ckey) ? (ckey != 0u) : (tinj < mytin && mytin < toutj);
        s += anc ? cj : 0u;
    }
    s = in ? s : 0u;
    u32 best = 0;
    for (u32 j = 0; j < D; ++j) { const u32 sj = readlane(s, (int)j); best = sj > best ? sj : best; }
    u64 tie = ballot64(in && s == best);
    u32 acc = 0;
    bool first = true;
    while (tie) {
        const int l = __builtin_ctzll(tie);
        tie &= tie - 1;
        const u32 t = readlane(ckey, l);
        if (first) { acc = t; first = false; }
        else acc = lca_dev(nodes, n_nodes, acc, t);
    }
    return acc;
}

// =====================================================================================================
// classify: one wavefront per unit (read or mate pair).
// =====================================================================================================
// KT > 0 fixes k at compile time (contiguous seeds only): shift counts, masks and the minimizer span become immediates,
// which also frees the SGPRs those loop-invariant values would occupy.  KT == 0 reads k from the arguments.
// NM > 0 fixes the number of mates per unit the same way (1 = single-end: no mate loop, no third offset).
// offv = offsets of the unit's reads, one per lane (lanes 0..nmates); (have0, r_lo, r_hi) = prefetched pass 0 of mate 0.
// ob = lane of offv that holds the unit's first offset (the caller keeps a whole chunk's offsets in one register pair)
template <bool SPACED, int LAYOUT, int KT, int NM, int NB = 16, int SPAN = 8, bool OVC = false, bool WIDE = false, bool PACKED = false>
__device__ __forceinline__ void classify_unit(const ClassifyParams &p, u64 u, u64 offv, u32 ob, bool have0, const Prefetch &pre0,
                                              u32 *keys, u32 *cnt, u32 *tin, u32 *tout, u32 cap, bool record_overflow, u32 *ring, u32 *aux, u64 *pk,
                                              uint4 &rec_out, bool &rec_valid)
{
    const int lane = lane_id();
    rec_valid = false;
    const u32 rdesc = SPACED ? run_desc(p) : 0u;
    const u32 k = KT ? (u32)KT : p.k, c = KT ? (u32)KT : p.c;
    const u32 mlen = KT ? (u32)(KT - SPAN) : p.m;                 // (a compile-time k comes with its window: m = k - SPAN)
    constexpr int MW = KT ? SPAN + 1 : BNS_MAX_SPAN + 1;
    const int nm = NM ? NM : p.nmates;
    u32 D = 0, n_hits = 0, missing = 0, ambig = 0;
    u32 ckey = 0, ccnt = 0;                                       // counter entries 0..63, one per lane (counter_add_reg)
    u64 dmask = 0;
    bool overflow = false;
    const bool want_hits = p.want_hits != 0;
    const u32 rounds_per_chunk = (2048u - (c - 1u)) / 64u;
    // the bucket count is wanted in the middle of every round (the multiply-high of bucket_of): fetched from the kernarg segment
    // HERE, once per unit, and pinned in an SGPR -- left to itself the compiler re-loads it right in front of its use, and the
    // scalar load's latency (and its lgkmcnt wait, shared with LDS) lands on the round's critical path
    u32 n_mb = p.n_mb;
    asm volatile("" : "+s"(n_mb));

    // the second mate's first 256 bases are asked for now and arrive while the first mate is classified (contiguous seeds: -2 %;
    // the spaced instantiations have no registers to spare for it)
    Prefetch pre1{0u, 0u, 0u};
    const bool have1 = !SPACED && NM == 2 && !OVC;            // (the k = 31 instantiations; the generic ones and the cooperative-overflow form have no registers to spare)
    if (have1) prefetch_read<PACKED>(p, u * (u64)nm + 1u, readlane64(offv, (int)ob + 1), readlane((u32)offv, (int)ob + 2) - readlane((u32)offv, (int)ob + 1), pre1);
#ifdef BNS_PAD_UNIT                                             // marginal-cost experiments: N extra instructions per unit
    { u32 pv = (u32)lane; for (int q = 0; q < BNS_PAD_UNIT; ++q) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(pv)); asm volatile("" :: "v"(pv)); }
#endif
    for (int m = 0; m < nm; ++m) {
        const u32 L = readlane((u32)offv, (int)ob + m + 1) - readlane((u32)offv, (int)ob + m);     // (reads are < 4 GiB: the low words suffice)
        const u32 nk = (L >= c && !(SPACED && p.emit_none)) ? L - c + 1u : 0u;     // (emit_none: spaced seeds only)
        for (u32 j0 = 0; j0 < nk; j0 += rounds_per_chunk * 64u) {
            // pack the chunk into the per-wave LDS image; is any base inside the read not A/C/G/T?  (wave-uniform)
            const bool have = (m == 0 ? have0 : have1) && j0 == 0;
            const bool clean = PACKED ? load_chunk_packed<!SPACED>(p, u * (u64)nm + (u64)m, readlane64(offv, (int)ob + m), L, j0, have, m == 0 ? pre0 : pre1, pk)
                                      : pack_chunk_lds<!SPACED>(p.bases, offv, (int)ob + m, L, j0, have, m == 0 ? pre0.lo : pre1.lo, m == 0 ? pre0.hi : pre1.hi, pk);
            u64 W = 0; u32 M = 0xFFFFFFFFu;                        // register image: only the spaced paths use it
            if (SPACED) {
                const u32 n_written = PACKED ? 64u : ((L - j0 >= 2048u ? 2048u : L - j0) + 255u) / 256u * 8u;    // words the passes wrote
                W = (u32)lane < n_written ? pk[lane] : 0ULL;
                if (!p.n_runs) M = (PACKED && clean) ? 0u : mask2_to_mask1((u32)lane < n_written ? pk[64 + lane] : ~0ULL);   // (the comb <= 64 path reads the image itself)
            }
            const u32 chunk_nk = (nk - j0) < rounds_per_chunk * 64u ? (nk - j0) : rounds_per_chunk * 64u;
            for (u32 rd = 0; rd * 64u < chunk_nk; ++rd) {
                const u32 jl = rd * 64u + (u32)lane;
                u64 kmer, win = 0;
                bool valid;
                if (SPACED) valid = p.n_runs ? extract_spaced_lds(pk, rd, p, rdesc, kmer, clean) : extract_spaced(W, M, rd, k, rdesc, kmer);
                else        { extract_lds32(pk, rd, k, clean, win, valid); kmer = win >> (64u - 2u * k); }
                valid = valid && jl < chunk_nk;
#ifdef BNS_PAD_VALU                                            // marginal-cost experiments (tools/pad.sh): N extra instructions per round
                { u32 pv = (u32)lane; for (int q = 0; q < BNS_PAD_VALU; ++q) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(pv)); asm volatile("" :: "v"(pv)); }
#endif
#ifdef BNS_PAD_VFAST
                { u32 pv = (u32)lane; for (int q = 0; q < BNS_PAD_VFAST; ++q) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pv)); asm volatile("" :: "v"(pv)); }
#endif
#ifdef BNS_PAD_SALU
                { for (int q = 0; q < BNS_PAD_SALU; ++q) asm volatile("s_add_u32 s20, s20, 3" ::: "s20", "scc"); }
#endif
#ifdef BNS_PAD_LDS
                { u32 pv; for (int q = 0; q < BNS_PAD_LDS; ++q) asm volatile("ds_read_b32 %0, %1" : "=v"(pv) : "v"((u32)lane * 4u)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
                const u64 kf = kmer;
                const u64 krc = SPACED ? 0ULL : revcomp_top(win, k);
                if (!SPACED && (KT != 0 || p.canon)) kmer = kf < krc ? kf : krc;      // (the fixed-k instantiations are canonical-only: launch_kt)
                ProbeResult pr;
#ifdef BNS_ABLATION                                           // profiling builds only (tools/ablate.sh): results are wrong
                if (p.dbg & 1) { pr.found = valid && (kmer & 1); pr.val = 1000u + (u32)(kmer & 3); }
                else
#endif
                if (LAYOUT == 2) {
                    u32 minh;
                    if (SPACED) minh = key_minhash(kmer, k, MinSpec{p.m, p.min_len, p.min_shift, p.min_canon, 0u});
                    else if (WIDE) minh = wide_bucket_in(round_minhash_wide<MW>(kf, krc, rd, k, mlen, reinterpret_cast<u64 *>(ring)), mlen);
                    else minh = round_minhash<MW, (KT != 0 && SPAN == 15)>(kf, krc, rd, k, mlen, ring);
#ifdef BNS_ABLATION
                    if (p.dbg & 4) minh = (u32)wang64(kmer);
#endif
                    // (crowded tables: the lane's group tag goes along, and a miss leaves its home bucket only when that group spilled)
                    pr = probe_minbucket<(KT == 0 || KT == 32), NB, OVC, true, OVC>(p.minb, kmer, bucket_of(minh, n_mb), valid, aux, p.slots, p.ovf_mask,
                                                                                     OVC ? minb_tagbit(minh) : 0u);
                } else if (LAYOUT == 1) pr = probe_bucket(p.slots, p.bucket_mask, kmer, valid);
                else                  pr = probe_khash(p.kflags, p.kkeys, p.kvals, p.kh_nb, kmer, valid);
                const u64 fm = ballot64(pr.found), vm = ballot64(valid);
                missing += (u32)__popcll(vm & ~fm);
                if (want_hits && pr.found) { u32 *hp = cold_params()->hits; hp[readlane64(offv, (int)ob) + n_hits + (u32)__popcll(fm & lanemask_lt())] = pr.val; }
                n_hits += (u32)__popcll(fm);
#ifdef BNS_ABLATION
                u64 rem = (p.dbg & 2) ? 0ULL : fm;
#else
                u64 rem = fm;
#endif
                while (rem) {
                    const int l = __builtin_ctzll(rem);
                    const u32 t = readlane(pr.val, l);
                    const u64 mm = ballot64(pr.val == t) & fm;
                    rem &= ~mm;
                    const u32 c = (u32)__popcll(mm);
                    const bool eq = ckey == t;
                    if (ballot64(eq) & dmask) { ccnt = eq ? ccnt + c : ccnt; continue; }     // the usual case: a taxon seen before
                    if (!counter_insert(ckey, ccnt, dmask, keys, cnt, cap, D, t, c)) { overflow = true; break; }
                }
            }
        }
        // classifier.h:232 / :235 -- u32 arithmetic with the cumulative totals, reproduced as written
        if (m == 0) ambig = L - c + 1u - n_hits - missing;
        else        ambig += L - (c - 1u) - n_hits - missing;
    }

    ColdParams *kp = cold_params();
    if (overflow && record_overflow) {
        if (lane == 0) {
            u32 *oc = kp->ovf_count;
            u64 *ol = kp->ovf_list;
            if (ol) { const u32 slot = atomicAdd(oc, 1u); ol[slot] = u; }
        }
        return;                                                // the overflow kernel recomputes this unit
    }
    u32 taxon;
    if (D <= 1u) taxon = D ? readlane(ckey, 0) : 0u;           // a lone taxon wins whatever its score (a zero score ties with the initial 0: lca(0,t)=t)
    else if (D <= 64u) taxon = resolve_regs(ckey, ccnt, D, kp->nodes, kp->n_nodes);       // the whole counter is in registers
    else {
        // (rare path -- more than 64 distinct taxa in one unit: its per-lane LDS addresses are formed HERE, from a lane id the
        // compiler cannot see through, instead of being hoisted to the top of the kernel and kept in registers -- or, in the
        // instantiations at the 64-register limit, in scratch -- across it)
        u32 l2 = (u32)lane;
        asm volatile("" : "+v"(l2));
        keys[l2] = ckey; cnt[l2] = ccnt;
        __builtin_amdgcn_wave_barrier();
        taxon = resolve_wave(keys, cnt, tin, tout, D, kp->nodes, kp->n_nodes, (int)(l2 & 63u));
    }
    rec_out = make_uint4(taxon, missing, ambig, n_hits);       // the caller stores it (one 16-byte record per unit)
    rec_valid = true;
}

#ifndef BNS_WAVES_PER_SIMD
#define BNS_WAVES_PER_SIMD 8
#endif
#ifdef BNS_WAVE_TIMES
__device__ unsigned long long g_wave_times[2 * 8192];
#endif
// Spaced seeds have no minimizer locality (every lookup its own bucket), so their rounds are bound by the random-gather rate
// of the memory system (41 G fetches/s reached, 44 G/s is the part's ceiling); a 32-bucket stage at 6 waves/SIMD measured
// 6 % faster than 16 buckets at 8 (two passes per round instead of four), a 64-bucket stage at 3-4 waves 9 % slower.
#ifndef BNS_SPACED_NB
#define BNS_SPACED_NB 32         // buckets staged per probe pass for spaced seeds (16 / 32 / 64)
#endif
#ifndef BNS_SPACED_WAVES
#define BNS_SPACED_WAVES 6       // waves per SIMD the spaced instantiations are compiled for
#endif
template <bool SPACED> struct ClassifyCfg { static constexpr int NB = 16, WAVES = BNS_WAVES_PER_SIMD; };
template <> struct ClassifyCfg<true> { static constexpr int NB = BNS_SPACED_NB, WAVES = BNS_SPACED_WAVES; };
template <bool SPACED, int LAYOUT, int KT, int NM, int SPAN = 8, bool OVC = false, bool WIDE = false, bool PACKED = false>
// (the 64-byte bucket layout stages four 16-byte slots per lane -- sixteen registers: 7 waves per SIMD, no scratch)
__global__ __launch_bounds__(256, (LAYOUT == 1 && !SPACED) ? 7 : ClassifyCfg<SPACED>::WAVES) void classify_kernel(ClassifyParams p)
{
    constexpr int NB = LAYOUT == 2 ? ClassifyCfg<SPACED>::NB : 16;
    constexpr int AUX_U32 = minb_aux_u32(NB);
    // per wave: counter keys/counts (1 KB), minimizer ring + bucket list + bucket stage (3.1 KB; the stage doubles as the
    // tin/tout scratch of resolve_wave, which runs when no probe is in flight), packed chunk image (1 KB): 19.8 KB / block
    __shared__ u32 s_keys[4][LDS_CAP], s_cnt[4][LDS_CAP];
    // (ring, list + stage and chunk image are separate arrays: the stage is written by the fetch itself (LDS DMA), and the compiler
    // puts a vmcnt wait in front of any LDS access it cannot tell apart from it)
    // (the ring holds 64 + window - 1 <= 79 entries: 32-bit hashes, or 64-bit identities for a table with the wide minimizer identity)
    __shared__ __attribute__((aligned(8))) u32 s_ring[4][WIDE ? 160 : 96];
    __shared__ __attribute__((aligned(16))) u32 s_mh[4][AUX_U32];
    __shared__ u64 s_pk[4][IMG_U64];
    static_assert(AUX_U32 - MINB_LIST_U32 >= 2 * (int)LDS_CAP, "stage must hold tin/tout");
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: keeps the unit loop scalar
    const int lane = lane_id();
#ifdef BNS_WAVE_TIMES                        // measurement builds only: when does each wavefront start and finish?
    const unsigned long long t_start = wall_clock64();
    struct TimeStamp { unsigned long long t0; u32 slot; __device__ ~TimeStamp() { if ((threadIdx.x & 63u) == 0) { g_wave_times[2 * slot] = t0; g_wave_times[2 * slot + 1] = wall_clock64(); } } } stamp{t_start, blockIdx.x * 4u + (u32)wv};
#endif
    // unit indices are 32-bit here (bns_classify_batch_device rejects batches of 2^32 units or more)
    const u32 n_units = (u32)p.n_units;
    const u32 nm = NM ? (u32)NM : (u32)p.nmates;
    // Work distribution: wavefronts CLAIM chunks of consecutive units (classify_chunk(): 63 reads / 31 pairs) from a counter instead of owning a fixed
    // stride of them.  With equal shares the wavefronts of one launch finished anywhere between 66 % and 100 % of its duration
    // (older waves win the issue arbitration and run ahead; measured with a -DBNS_WAVE_TIMES build, tools/wave_times.py), i.e. the
    // last third of the kernel ran on a thinning population of waves -- and this kernel needs all eight per SIMD to cover its
    // latencies.  Claiming keeps every wave busy until the batch is empty.
    // Software pipeline, per wave: the NEXT chunk's claim (one atomic) is in flight from the start of a chunk, its offsets (one
    // 64-lane load: 17 or 33 of them) from the chunk's third unit, the first 256 bases of the next unit while a unit is
    // classified.  Offsets travel through VECTOR loads so that LDS waits (lgkmcnt) never stall on them.
    u32 *const ctr = p.work_counter;
    const u32 CH = p.chunk;                                  // classify_chunk(nm), less for a batch too small to give every wave one that size
    auto claim = [&]() -> u32 { return lane == 0 ? atomicAdd(ctr, CH) : 0u; };                             // (lane 0 holds the result)
    // (the offsets pointer is taken from the kernarg segment at the point of use and advanced in SGPRs: kept as a per-lane address
    // across the kernel it cost two VGPRs -- or, in the instantiations at the 64-register limit, 8 bytes of scratch)
    auto load_offs = [&](u32 base) -> u64 {
        const u32 left = n_units - base, cnt = left < CH ? left : CH;
        const u64 *op = const_cast<const u64 *>(cold_params()->offsets) + (u64)base * nm;
        return (u32)lane <= cnt * nm ? op[lane] : 0ULL;
    };
    u32 base = (u32)__builtin_amdgcn_readfirstlane((int)claim());
    if (base >= n_units) return;
    u64 offs = load_offs(base);
    Prefetch pre;
    {
        const u64 o0 = readlane64(offs, 0);
        prefetch_read<PACKED>(p, (u64)base * nm, o0, readlane((u32)offs, 1) - (u32)o0, pre);
    }
    uint4 pend = make_uint4(0, 0, 0, 0);
    u32 pend_u = 0;
    bool pend_valid = false;
    for (;;) {
        const u32 left = n_units - base, cnt = left < CH ? left : CH;
        u32 next_v = claim();                                    // next chunk, claimed now, looked at two units from now
        u32 nbase = 0xFFFFFFFFu;
        u64 noffs = 0;
        const u32 jload = cnt > 2u ? 2u : cnt - 1u;
        for (u32 j = 0; j < cnt; ++j) {
            if (j == jload) {
                nbase = (u32)__builtin_amdgcn_readfirstlane((int)next_v);
                if (nbase < n_units) noffs = load_offs(nbase);
            }
            // first 256 bases of the unit after this one: the next of the chunk, or the first of the next chunk
            Prefetch npre{0u, 0u, 0u};
            if (j + 1u < cnt) {
                const u64 n0 = readlane64(offs, (int)((j + 1u) * nm));
                prefetch_read<PACKED>(p, (u64)(base + j + 1u) * nm, n0, readlane((u32)offs, (int)((j + 1u) * nm + 1u)) - (u32)n0, npre);
            } else if (nbase < n_units) {
                const u64 n0 = readlane64(noffs, 0);
                prefetch_read<PACKED>(p, (u64)nbase * nm, n0, readlane((u32)noffs, 1) - (u32)n0, npre);
            }
            // The previous unit's record is stored HERE, next to the prefetch loads: gfx9 has one counter for loads and stores,
            // so the first wait after a store waits for its acknowledgement too -- this way that is the first bucket fetch.
            if (pend_valid && lane == 0) cold_params()->records[pend_u] = pend;
            classify_unit<SPACED, LAYOUT, KT, NM, NB, SPAN, OVC, WIDE, PACKED>(p, base + j, offs, j * nm, true, pre, s_keys[wv], s_cnt[wv], s_mh[wv] + MINB_LIST_U32,
                                          s_mh[wv] + MINB_LIST_U32 + LDS_CAP, LDS_CAP, true, s_ring[wv], s_mh[wv], s_pk[wv], pend, pend_valid);
            pend_u = base + j;
            pre = npre;
        }
        if (nbase >= n_units) break;
        base = nbase; offs = noffs;
    }
    if (pend_valid && lane == 0) cold_params()->records[pend_u] = pend;
}

// Overflow path: units with more than LDS_CAP distinct taxa.  One wavefront per listed unit; the counter
// lives in global scratch at the unit's own base offset (a unit has at most as many k-mers as bases).
template <bool SPACED, int LAYOUT, bool WIDE = false, bool PACKED = false>
__global__ __launch_bounds__(64) void classify_overflow_kernel(ClassifyParams p, u32 *scratch, u64 total_bases)
{
    __shared__ __attribute__((aligned(8))) u32 s_ring[WIDE ? 160 : 96];
    __shared__ __attribute__((aligned(16))) u32 s_mh[MINB_AUX_U32];
    __shared__ u64 s_pk[IMG_U64];
    const u32 n = *p.ovf_count;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        const u64 u = p.ovf_list[i];
        const u64 b0 = p.offsets[u * (u64)p.nmates];
        const u64 bm = p.offsets[u * (u64)p.nmates + 1];
        const u64 b1 = p.offsets[(u + 1) * (u64)p.nmates];
        uint4 rec;
        bool ok;
        const u64 offv = (threadIdx.x & 63u) == 0 ? b0 : ((threadIdx.x & 63u) == 1 ? bm : b1);
        classify_unit<SPACED, LAYOUT, 0, 0, 16, 8, false, WIDE, PACKED>(p, u, offv, 0u, false, Prefetch{0u, 0u, 0u}, scratch + b0, scratch + total_bases + b0,
                                      scratch + 2 * total_bases + b0, scratch + 3 * total_bases + b0, (u32)(b1 - b0), false, s_ring, s_mh, s_pk, rec, ok);
        if (ok && threadIdx.x == 0) p.records[u] = rec;
    }
}

// =====================================================================================================
// Encoder::for_each over a batch (encoder.h:415-442): emits the k-mer stream of every read, in order.
// =====================================================================================================
template <bool SPACED>
__global__ __launch_bounds__(256) void encode_kernel(ClassifyParams p, u64 *__restrict__ kmers, u32 *__restrict__ n_kmers)
{
    __shared__ u64 s_win[4][256];
    const int lane = lane_id();
    const u32 rdesc = SPACED ? run_desc(p) : 0u;
    const u64 wave = (u64)blockIdx.x * 4 + (u64)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 n_waves = (u64)gridDim.x * 4;
    const u32 k = p.k, c = p.c;
    const bool windowed = p.w > c;
    const u32 span = windowed ? p.w : c;                          // bases one emitted value needs
    const u32 rounds_per_chunk = (2048u - (span - 1u)) / 64u;
    u64 *win = s_win[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    if (!SPACED && windowed && (!p.canon || p.score == 2)) {                       // for_each_uncanon_unspaced_windowed
        for (u64 r = wave; r < p.n_units; r += n_waves) {
            const u64 o = p.offsets[r];
            u32 emitted = 0;
            auto put = [&](u64 kmer, bool on) {
                const u64 vm = ballot64(on);
                if (on) kmers[o + emitted + (u32)__popcll(vm & lanemask_lt())] = kmer;
                emitted += (u32)__popcll(vm);
            };
            if (p.w - p.k + 1u > 64u) uncanon_windowed_seq<true>(p, r, p.win_scratch + wave * 2 * (u64)(p.w - p.k + 65u), put);
            else                      uncanon_windowed_seq<false>(p, r, win, put);
            if (lane == 0) n_kmers[r] = emitted;
        }
        return;
    }
    for (u64 r = wave; r < p.n_units; r += n_waves) {
        const u64 o = p.offsets[r];
        const u32 L = (u32)(p.offsets[r + 1] - o);
        const u64 wb = (o >> 5) + r;
        const u32 n_words = (L + 31u) >> 5;
        const u32 nk = (L >= span && !p.emit_none) ? L - span + 1u : 0u;
        u32 emitted = 0;
        for (u32 j0 = 0; j0 < nk; j0 += rounds_per_chunk * 64u) {
            const u32 wi = (j0 >> 5) + (u32)lane;
            const u64 W = wi < n_words ? p.words[wb + wi] : 0ULL;
            const u32 M = wi < n_words ? p.nmask[wb + wi] : 0xFFFFFFFFu;
            const u32 chunk_nk = (nk - j0) < rounds_per_chunk * 64u ? (nk - j0) : rounds_per_chunk * 64u;
            for (u32 rd = 0; rd * 64u < chunk_nk; ++rd) {
                const u32 jl = rd * 64u + (u32)lane;
                u64 kmer;
                bool valid;
                if (windowed) { kmer = windowed_round<SPACED>(W, M, rd, p, rdesc, win); valid = !SPACED || kmer != ~0ULL; }   // contiguous: every window emits (overflow -> 0)
                else if (SPACED) valid = p.n_runs ? extract_spaced_runs(W, M, rd, p, rdesc, kmer) : extract_spaced(W, M, rd, k, rdesc, kmer);
                else        valid = extract_unspaced(W, M, rd, k, kmer);
                valid = valid && jl < chunk_nk;
                if (!SPACED && !windowed && p.canon) kmer = canonical(kmer, k);
                const u64 vm = ballot64(valid);
                if (valid) kmers[o + emitted + (u32)__popcll(vm & lanemask_lt())] = kmer;
                emitted += (u32)__popcll(vm);
            }
        }
        if (lane == 0) n_kmers[r] = emitted;
    }
}

// =====================================================================================================
// kh_get over a batch of keys.
// =====================================================================================================
// MIN: 0 = the table's minimizer is taken over the whole canonical key (contiguous seeds), 1 = inside a sub-run of the key
// (spaced seeds), 2 = whole key with the wide minimizer identity
template <int LAYOUT, int MIN = 0>
__global__ __launch_bounds__(256, 8) void probe_kernel(ClassifyParams p, const u64 *__restrict__ keys, u64 n,
                                                    u32 *__restrict__ vals, u8 *__restrict__ found)
{
    __shared__ __attribute__((aligned(16))) u32 s_aux[4][MINB_AUX_U32];
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 n_round = (n + 63) & ~63ULL;                       // keep whole wavefronts in the loop (DPP)
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool active = i < n;
        const u64 key = active ? keys[i] : 0ULL;
        ProbeResult pr;
        if (LAYOUT == 2) {
            const u32 minh = MIN == 1 ? key_minhash(key, p.k, MinSpec{p.m, p.min_len, p.min_shift, p.min_canon, 0u})
                                      : (MIN == 2 ? key_minhash<true>(key, p.k, p.m) : key_minhash<false>(key, p.k, p.m));
            pr = probe_minbucket<true, 16, false, false>(p.minb, key, bucket_of(minh, p.n_mb), active, s_aux[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))], p.slots, p.ovf_mask);
        }
        else if (LAYOUT == 1) pr = probe_bucket(p.slots, p.bucket_mask, key, active);
        else                  pr = probe_khash(p.kflags, p.kkeys, p.kvals, p.kh_nb, key, active);
        if (active) { vals[i] = pr.found ? pr.val : 0u; if (found) found[i] = pr.found ? 1 : 0; }
    }
}

// =====================================================================================================
// khash arrays -> bucket layout.  One thread per khash slot; a present slot claims the first free slot
// of the first non-full bucket on its (triangular, bucket-granular) probe path.
// =====================================================================================================
__global__ __launch_bounds__(256) void rebucket_kernel(const u32 *__restrict__ flags, const u64 *__restrict__ keys,
                                                       const u32 *__restrict__ vals, u64 n_buckets, Slot *slots,
                                                       u64 bucket_mask, unsigned long long *n_present)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 local = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_buckets; i += stride) {
        const u32 f = (flags[i >> 4] >> ((i & 0xfu) << 1)) & 3u;
        if (f) continue;                                       // empty or deleted
        ++local;
        const u64 key = keys[i];
        const u32 val = vals[i];
        u64 b = wang64(key) & bucket_mask, step = 0;
        for (;;) {
            bool placed = false;
            for (int s = 0; s < 4 && !placed; ++s) {
                Slot *sl = &slots[b * 4 + (u64)s];
                if (atomicCAS(&sl->occ, 0u, 1u) == 0u) { sl->key = key; sl->val = val; placed = true; }
            }
            if (placed) break;
            b = (b + (++step)) & bucket_mask;
        }
    }
    if (local) atomicAdd(n_present, (unsigned long long)local);
}

// present keys of a khash (what kh_size would say): one thread per flag word
__global__ __launch_bounds__(256) void count_present_kernel(const u32 *__restrict__ flags, u64 n_buckets, unsigned long long *out)
{
    const u64 n_fw = n_buckets < 16 ? 1 : n_buckets >> 4, stride = (u64)gridDim.x * blockDim.x;
    u64 local = 0;
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_fw; w += stride) {
        u32 f = flags[w];
        f = (f | (f >> 1)) & 0x55555555u;                       // 1 per slot that is empty or deleted
        const u32 slots = n_buckets < 16 ? (u32)n_buckets : 16u;
        local += slots - (u32)__popc(slots == 16u ? f : (f & ((1u << (2u * slots)) - 1u)));
    }
    if (local) atomicAdd(out, (unsigned long long)local);
}

// khash arrays -> minimizer-clustered layout: claim the next index of the home bucket (CAS on its count), spill
// to the following bucket when it is full -- at most MINB_MAX_CHAIN buckets, after which the key is left for the
// overflow pass; minbucket_place_kernel then moves every bucket's keys to their perfect-hash slots.
// A key placed outside its home bucket sets, in the HOME bucket's header, the depth bits (how far down the chain lookups that
// start there must be prepared to walk) and its group's tag bit (which lookups need walk at all: bns_device.hpp).
//
// Group-aware fill (crowded tables: the loader takes this route when more than 1 key in 100 would miss its home bucket).  Which
// keys of an overfull home spill decides what lookups cost: a read's consecutive k-mers share their minimizer -- a GROUP of up to
// window + 1 keys -- and its lookups take a second pass as soon as ONE key of the group is elsewhere.  Filled in arrival order, a
// home with groups of 6 and 7 keys loses three random keys: both groups pay.  So:
//   minbucket_tagcount_kernel   every key counts itself in its home bucket, per group tag (8 counters in the still empty vals[])
//   minbucket_decide_kernel     per bucket: the tags that stay WHOLE -- largest first, while they fit -- go into `pad` as a mask
//   minbucket_fill_kernel<1>    keys of resident tags take their home slots (they fit by construction)
//   minbucket_fill_kernel<2>    the rest: what room is left at home, then down the chain as before
// (8e9-key every-k-mer db at 46 % load, simulated and measured: runs of a read that need a second pass 46 % -> 22-24 %.)
// MODE 0 = every key in arrival order (tables with room: nothing to decide).
__global__ __launch_bounds__(256) void minbucket_tagcount_kernel(const u32 *__restrict__ flags, const u64 *__restrict__ keys, u64 n_buckets,
                                                                 MinBucket *out, u32 n_mb, u32 k, MinSpec m)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_buckets; i += stride) {
        const u32 f = (flags[i >> 4] >> ((i & 0xfu) << 1)) & 3u;
        if (f) continue;
        const u32 minh = key_minhash(keys[i], k, m);
        atomicAdd(&out[bucket_of(minh, n_mb)].vals[minb_tag(minh)], 1u);
    }
}

// stats[0] += keys of resident tags, stats[1] += keys of the others
__global__ __launch_bounds__(256) void minbucket_decide_kernel(MinBucket *out, u64 n_mb, unsigned long long *stats)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 n_res = 0, n_out = 0;
    for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_mb; b += stride) {
        MinBucket *mb = &out[b];
        u32 c[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { c[t] = mb->vals[t]; mb->vals[t] = 0u; }
        u32 resident = 0, used = 0, left = 0xFFu;
#pragma unroll
        for (int round = 0; round < 8; ++round) {                   // largest remaining tag first
            u32 best = 0, bt = 8;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (((left >> t) & 1u) && c[t] > best) { best = c[t]; bt = (u32)t; }
            if (bt == 8u) break;
            left &= ~(1u << bt);
            if (used + best <= MINB_CAP) { used += best; resident |= 1u << bt; n_res += best; } else n_out += best;
        }
        mb->pad = resident;
    }
    if (n_res) atomicAdd(stats, (unsigned long long)n_res);
    if (n_out) atomicAdd(stats + 1, (unsigned long long)n_out);
}

template <int MODE>
__global__ __launch_bounds__(256) void minbucket_fill_kernel(const u32 *__restrict__ flags, const u64 *__restrict__ keys,
                                                             const u32 *__restrict__ vals, u64 n_buckets, MinBucket *out,
                                                             u32 n_mb, unsigned long long *n_present, u32 k, MinSpec m)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 local = 0, local_ovf = 0, local_spill = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_buckets; i += stride) {
        const u32 f = (flags[i >> 4] >> ((i & 0xfu) << 1)) & 3u;
        if (f) continue;
        const u64 key = keys[i];
        const u32 minh = key_minhash(key, k, m);
        const u32 home = bucket_of(minh, n_mb);
        if (MODE != 0) {
            const bool resident = (__atomic_load_n(&out[home].pad, __ATOMIC_RELAXED) >> minb_tag(minh)) & 1u;
            if (resident != (MODE == 1)) continue;
        }
        ++local;
        const u32 val = vals[i];
        u32 b = home;
        bool placed = false;
        for (u32 chain = 0; chain < MINB_MAX_CHAIN && !placed; ++chain) {
            MinBucket *mb = &out[b];
            u32 old = __atomic_load_n(&mb->n, __ATOMIC_RELAXED);
            while ((old & 0xFFu) < MINB_CAP) {                          // (the header's home bits change under other threads' atomicOr)
                const u32 seen = atomicCAS(&mb->n, old, old + 1u);
                if (seen == old) { mb->keys[old & 0xFFu] = key; mb->vals[old & 0xFFu] = val; placed = true; break; }
                old = seen;
            }
            // the HOME bucket remembers how far down its chain its keys went (probe_minbucket walks no further), and whose they were
            if (placed && chain) { ++local_spill; atomicOr(&out[home].n, (((1u << chain) - 1u) << MINB_HOME_SHIFT) | minb_tagbit(minh)); }
            ++b;                                                       // (no wrap: MINB_MAX_CHAIN - 1 buckets behind the last home)
        }
        if (!placed) ++local_ovf;
    }
    if (local) atomicAdd(n_present, (unsigned long long)local);
    if (local_ovf) atomicAdd(n_present + 1, (unsigned long long)local_ovf);
    if (local_spill) atomicAdd(n_present + 4, (unsigned long long)local_spill);      // keys that are not in their home bucket
}

// Which minimizer window suits this db?  ONE pass over the khash arrays tries every candidate at once, on a sample: the keys whose
// home bucket (under that candidate) lies in the first n_mb / div buckets are poured into a count-only image of those buckets --
// same chain rule as minbucket_fill_kernel -- and the keys that miss their home bucket are counted.  Sampling by BUCKET keeps
// whole minimizer groups (a sample of khash slots would keep one key of each group and see no crowding at all).
// cnt: n_cand images of (sample + MINB_MAX_CHAIN) u32 counters; out[3 c + {0,1,2}] = keys sampled / spilled / chain exhausted.
struct TrialCands { MinSpec c[6]; u32 n; };
__global__ __launch_bounds__(256) void minbucket_trial_kernel(const u32 *__restrict__ flags, const u64 *__restrict__ keys, u64 n_buckets,
                                                              u32 n_mb, u32 sample, u32 *cnt, unsigned long long *out, u32 k, TrialCands tc)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 seen[6] = {0, 0, 0, 0, 0, 0}, spill[6] = {0, 0, 0, 0, 0, 0}, ovf[6] = {0, 0, 0, 0, 0, 0};
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_buckets; i += stride) {
        const u32 f = (flags[i >> 4] >> ((i & 0xfu) << 1)) & 3u;
        if (f) continue;
        const u64 key = keys[i];
        for (u32 c = 0; c < tc.n; ++c) {
            const u32 home = bucket_of(key_minhash(key, k, tc.c[c]), n_mb);
            if (home >= sample) continue;
            ++seen[c];
            u32 *img = cnt + (u64)c * (sample + MINB_MAX_CHAIN);
            bool placed = false;
            for (u32 chain = 0; chain < MINB_MAX_CHAIN && !placed; ++chain) {
                u32 old = __atomic_load_n(&img[home + chain], __ATOMIC_RELAXED);
                while (old < MINB_CAP) {
                    const u32 was = atomicCAS(&img[home + chain], old, old + 1u);
                    if (was == old) { placed = true; break; }
                    old = was;
                }
                if (placed && chain) ++spill[c];
            }
            if (!placed) ++ovf[c];
        }
    }
    for (u32 c = 0; c < tc.n; ++c) {
        if (seen[c]) atomicAdd(out + 3 * c, (unsigned long long)seen[c]);
        if (spill[c]) atomicAdd(out + 3 * c + 1, (unsigned long long)spill[c]);
        if (ovf[c]) atomicAdd(out + 3 * c + 2, (unsigned long long)ovf[c]);
    }
}

// Overflow pass (before the sort): every present khash key that is not in one of its MINB_MAX_CHAIN buckets goes into the
// plain-hashed overflow table (64-byte buckets of 4 slots, triangular spill -- the BUCKET layout).
// Claim a slot of the overflow table (64-byte buckets of 4 slots, triangular spill).  False when the table is full.
__device__ __forceinline__ bool ovf_insert(Slot *ovf, u64 ovf_mask, u64 key, u32 val)
{
    u64 ob = ovf_bucket(key, ovf_mask);
    for (u64 step = 0; step <= ovf_mask; ) {
        for (int s = 0; s < 4; ++s) {
            Slot *sl = &ovf[ob * 4 + (u64)s];
            if (atomicCAS(&sl->occ, 0u, 1u) == 0u) { sl->key = key; sl->val = val; return true; }
        }
        ob = (ob + (++step)) & ovf_mask;
    }
    return false;
}

__global__ __launch_bounds__(256) void minbucket_overflow_kernel(const u32 *__restrict__ flags, const u64 *__restrict__ keys,
                                                                 const u32 *__restrict__ vals, u64 n_buckets, const MinBucket *mbk,
                                                                 u32 n_mb, Slot *ovf, u64 ovf_mask, u32 k, MinSpec m, u32 *error)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_buckets; i += stride) {
        const u32 f = (flags[i >> 4] >> ((i & 0xfu) << 1)) & 3u;
        if (f) continue;
        const u64 key = keys[i];
        const u32 minh = key_minhash(key, k, m);
        const u32 home = bucket_of(minh, n_mb);
        u32 b = home;
        bool found = false, all_full = true;
        for (u32 chain = 0; chain < MINB_MAX_CHAIN && !found && all_full; ++chain) {
            const MinBucket *mb = &mbk[b];
            const u32 cnt = __atomic_load_n(&mb->n, __ATOMIC_RELAXED) & 0xFFu;          // (other threads may be flagging it)
            const u32 n = cnt < MINB_CAP ? cnt : MINB_CAP;
            for (u32 j = 0; j < n; ++j) found |= mb->keys[j] == key;
            all_full = n == MINB_CAP;
            ++b;
        }
        if (found || !all_full) continue;                      // (!all_full && !found cannot happen for a placed key)
        if (!ovf_insert(ovf, ovf_mask, key, vals[i])) *error = 1u;
        // the key's HOME bucket remembers that one of its keys lives in the overflow table (and, with that, beyond every bucket
        // of its chain): only lookups that start there walk the whole chain and go on to the overflow table (probe_minbucket)
        atomicOr(const_cast<u32 *>(&mbk[home].n), MINB_HOME_MASK | minb_tagbit(minh));
    }
}

// Last step of the load: one wavefront per bucket looks for the bucket's perfect-hash multiplier (mph_slot), 64 candidates
// per iteration, and moves the keys to their slots; unused slots get ~0, the header gets count | occupancy << 8 and S.
// A bucket without a multiplier (two keys with one fold, about one bucket in 10^8) has its keys moved to the overflow table; it
// then reads MINB_N_IN_OVF -- full, nothing in it -- and every moved key's HOME bucket gets all four home bits, so that lookups
// of those keys walk their whole chain and go on to the overflow table.  Other wavefronts may be setting home bits of THIS bucket
// at that moment: the header is updated with atomics only.
// test_fail_mod != 0 (tests only, bns_debug_set bit 0x100): every test_fail_mod-th bucket pretends to have found no multiplier.
__global__ __launch_bounds__(256) void minbucket_place_kernel(MinBucket *out, u64 n_bucket, u32 n_mb, Slot *ovf, u64 ovf_mask,
                                                              unsigned long long *n_moved, u32 *error, u32 test_fail_mod, u32 k, MinSpec m)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    constexpr u32 MAX_IT = 4096;                                         // 262144 candidates: P(miss) < e^-90 for a solvable bucket
    for (u64 b = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_bucket; b += n_waves) {
        MinBucket *mb = &out[b];
        const u32 raw = (u32)__builtin_amdgcn_readfirstlane((int)__atomic_load_n(&mb->n, __ATOMIC_RELAXED));
        const u32 cnt = raw & 0xFFu;
        const u32 n = cnt < MINB_CAP ? cnt : MINB_CAP;
        const u64 key = lane < n ? mb->keys[lane] : ~0ULL;
        const u32 val = lane < n ? mb->vals[lane] : 0u;
        const u32 x = mph_fold(key);
        u32 S = n ? 0u : 1u;                                             // an empty bucket needs no search
        for (u32 it = 0; it < MAX_IT && !S; ++it) {
            const u32 cand = mph_candidate(b, it * 64u + lane);
            u32 mask = 0;
            bool ok = true;
            for (u32 i = 0; i < n; ++i) {
                const u32 s = mph_slot((u32)__builtin_amdgcn_readlane((int)x, (int)i), cand);
                ok &= !((mask >> s) & 1u);
                mask |= 1u << s;
            }
            const u64 w = __builtin_amdgcn_ballot_w64(ok);
            if (w) S = (u32)__builtin_amdgcn_readlane((int)cand, __builtin_ctzll(w));
        }
        if (test_fail_mod && n && b % test_fail_mod == 0) S = 0u;
        if (lane < MINB_CAP) { mb->keys[lane] = ~0ULL; mb->vals[lane] = 0u; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // (same-address stores of one wave stay in order)
        if (!S) {                                                        // no perfect hash (two keys with one fold): off to the overflow table
            if (lane < n) {
                if (!ovf_insert(ovf, ovf_mask, key, val)) *error = 1u;
                const u32 minh = key_minhash(key, k, m);
                atomicOr(&out[bucket_of(minh, n_mb)].n, MINB_HOME_MASK | minb_tagbit(minh));
            }
            if (lane == 0) { atomicOr(&mb->n, MINB_N_IN_OVF); mb->pad = 1u; atomicAdd(n_moved, (unsigned long long)n); }   // (count <= 10: or-ing 0xFF sets it)
            continue;
        }
        const u32 slot = lane < n ? mph_slot(x, S) : 0u;
        u32 occ = lane < n ? 1u << slot : 0u;
        for (int off = 8; off >= 1; off >>= 1) occ |= (u32)__shfl_xor((int)occ, off);
        if (lane < n) { mb->keys[slot] = key; mb->vals[slot] = val; }
        if (lane == 0) { atomicOr(&mb->n, occ << 8); mb->pad = S; }      // (the count is already there; home bits stay as they are)
    }
}

// =====================================================================================================
// Database construction on device (feature_min.h:205-228 update_lca_map):
//   pass 1  every k-mer of every genome claims a khash slot (CAS on the key array; flags derived later)
//   pass 2  every k-mer folds its genome's taxid into the slot value with lca() (order-independent:
//           lca is associative/commutative on a rooted forest; first sighting stores the taxid itself)
//   pass 3  flags: present slots -> 00, others -> 10 (empty), keys/vals of empty slots zeroed (util.h:282-284)
// The layout satisfies the kh_get invariant (no empty slot before a key on its triangular probe path).
// =====================================================================================================
constexpr u64 BUILD_MAX_CHAIN = 16384; // probe steps after which build_kernel pass 1 declares the table too small
constexpr u64 BUILD_EMPTY = ~0ULL;      // not a legal k-mer unless k == 32 non-canonical; rejected by the host for that case
// tvals start at 0: lca() treats 0 as the identity (util.h:646-647), so "first sighting stores the taxid"
// and "later sightings store lca(taxid, old)" are the same fold.

template <bool SPACED, int PASS>
__global__ __launch_bounds__(256) void build_kernel(ClassifyParams p, const u32 *__restrict__ taxid, u64 n_buckets,
                                                    u64 *__restrict__ tkeys, u32 *__restrict__ tvals,
                                                    unsigned long long *n_inserted)
{
    __shared__ u64 s_win[4][256];
    const int lane = lane_id();
    const u32 rdesc = SPACED ? run_desc(p) : 0u;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u64 wave = (u64)blockIdx.x * 4 + (u64)wv;
    const u64 n_waves = (u64)gridDim.x * 4;
    const u32 k = p.k, c = p.c;
    const bool windowed = p.w > c;                                // db thinned by windowed minimizers (bonsai build -w)
    const u32 span = windowed ? p.w : c;
    const u32 rounds_per_chunk = (2048u - (span - 1u)) / 64u;
    const u64 mask = n_buckets - 1;
    u32 local = 0;
    auto apply = [&](u64 kmer, bool valid, u32 tx) {
        if (!valid) return;
        u64 i = wang64(kmer) & mask, step = 0;
        if (PASS == 1) {
            // Bounded: an undersized table must come back as BNS_ERR_TABLE, not hang.  Triangular probing visits every slot of
            // a power-of-two table within n_buckets steps; a chain beyond BUILD_MAX_CHAIN steps cannot occur at a load the
            // khash contract allows (<= 0.77: probability ~ 0.77^16384), so it means "too small" as well.  n_inserted[1] is the
            // error flag; once it is up every lane stops probing (checked each 64 steps).
            const u64 limit = n_buckets < BUILD_MAX_CHAIN ? n_buckets : BUILD_MAX_CHAIN;
            for (;;) {
                const u64 old = atomicCAS((unsigned long long *)&tkeys[i], (unsigned long long)BUILD_EMPTY,
                                          (unsigned long long)kmer);
                if (old == BUILD_EMPTY) { ++local; break; }
                if (old == kmer) break;
                i = (i + (++step)) & mask;
                if (step >= limit) { atomicExch(&n_inserted[1], 1ULL); break; }
                if ((step & 63u) == 0 && __atomic_load_n(&n_inserted[1], __ATOMIC_RELAXED)) break;
            }
        } else {
            while (tkeys[i] != kmer) i = (i + (++step)) & mask;
            u32 cur = tvals[i];
            for (;;) {
                const u32 want = cur == tx ? tx : lca_dev(p.nodes, p.n_nodes, tx, cur);
                if (want == cur) break;
                const u32 prev = atomicCAS(&tvals[i], cur, want);
                if (prev == cur) break;
                cur = prev;
            }
        }
    };
    if (!SPACED && windowed && (!p.canon || p.score == 2)) {
        // for_each_uncanon_unspaced_windowed: the windows run over the emitted stream, so a sequence is one work item
        for (u64 r = wave; r < p.n_units; r += n_waves) {
            const u32 tx = taxid[r];
            auto put = [&](u64 kmer, bool on) {
                const u64 prev = ((u64)dpp<DPP_WAVE_SHR1>((u32)(kmer >> 32)) << 32) | dpp<DPP_WAVE_SHR1>((u32)kmer);
                apply(kmer, on && (lane == 0 || prev != kmer), tx);              // consecutive windows mostly repeat their minimizer
            };
            if (p.w - p.k + 1u > 64u) uncanon_windowed_seq<true>(p, r, p.win_scratch + wave * 2 * (u64)(p.w - p.k + 65u), put);
            else                      uncanon_windowed_seq<false>(p, r, s_win[wv], put);
        }
        if (PASS == 1 && local) atomicAdd(n_inserted, (unsigned long long)local);
        return;
    }
    // work item = (genome, chunk): genomes are long, so chunks of one genome are spread over many waves
    for (u64 r = 0; r < p.n_units; ++r) {
        const u64 o = p.offsets[r];
        const u64 Lg = p.offsets[r + 1] - o;
        const u64 wb = (o >> 5) + r;
        const u64 n_words = (Lg + 31u) >> 5;
        const u64 nk = Lg >= span ? Lg - span + 1u : 0u;
        const u32 tx = taxid[r];
        const u64 chunk_k = (u64)rounds_per_chunk * 64u;
        const u64 n_chunks = (nk + chunk_k - 1) / chunk_k;
        for (u64 ch = wave; ch < n_chunks; ch += n_waves) {
            const u64 j0 = ch * chunk_k;
            const u64 wi = (j0 >> 5) + (u64)lane;
            const u64 W = wi < n_words ? p.words[wb + wi] : 0ULL;
            const u32 M = wi < n_words ? p.nmask[wb + wi] : 0xFFFFFFFFu;
            const u32 chunk_nk = (nk - j0) < chunk_k ? (u32)(nk - j0) : (u32)chunk_k;
            for (u32 rd = 0; rd * 64u < chunk_nk; ++rd) {
                const u32 jl = rd * 64u + (u32)lane;
                u64 kmer;
                bool valid;
                if (windowed) { kmer = windowed_round<SPACED>(W, M, rd, p, rdesc, s_win[wv]); valid = !SPACED || kmer != ~0ULL; }
                else if (SPACED) valid = p.n_runs ? extract_spaced_runs(W, M, rd, p, rdesc, kmer) : extract_spaced(W, M, rd, k, rdesc, kmer);
                else        valid = extract_unspaced(W, M, rd, k, kmer);
                valid = valid && jl < chunk_nk;
                if (!SPACED && !windowed && p.canon) kmer = canonical(kmer, k);
                if (windowed) {                                   // consecutive windows mostly repeat their minimizer: skip repeats
                    const u64 prev = ((u64)dpp<DPP_WAVE_SHR1>((u32)(kmer >> 32)) << 32) | dpp<DPP_WAVE_SHR1>((u32)kmer);
                    if (lane != 0 && prev == kmer) valid = false;
                }
                apply(kmer, valid, tx);
            }
        }
    }
    if (PASS == 1 && local) atomicAdd(n_inserted, (unsigned long long)local);
}

__global__ __launch_bounds__(256) void build_finish_kernel(u64 n_buckets, u32 *__restrict__ flags, u64 *__restrict__ tkeys,
                                                           u32 *__restrict__ tvals)
{
    // one thread per flag word (16 slots)
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 n_fw = n_buckets < 16 ? 1 : n_buckets >> 4;
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_fw; w += stride) {
        u32 fw = 0;
        for (u32 s = 0; s < 16; ++s) {
            const u64 i = w * 16 + s;
            if (i >= n_buckets) { fw |= 2u << (2 * s); continue; }
            if (tkeys[i] == BUILD_EMPTY) { fw |= 2u << (2 * s); tkeys[i] = 0; tvals[i] = 0; }
        }
        flags[w] = fw;
    }
}

// {taxon, missing, ambig, n_hits} records -> the caller's separate arrays (any of the last three may be null)
__global__ __launch_bounds__(256) void unpack_kernel(const uint4 *__restrict__ rec, u64 n, u32 *__restrict__ taxon,
                                                     u32 *__restrict__ missing, u32 *__restrict__ ambig, u32 *__restrict__ n_hits)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 r = rec[i];
        taxon[i] = r.x;
        if (missing) missing[i] = r.y;
        if (ambig) ambig[i] = r.z;
        if (n_hits) n_hits[i] = r.w;
    }
}

// Run-length form of the ordered hit stream (what the Kraken line prints, classifier.h:45-61): one wavefront per group of
// HIT_RUNS_GROUP consecutive units.  Pass 1 counts every unit's runs; the group reserves its output entries with ONE atomicAdd
// (one per unit was 112 k same-address atomics per CLI chunk: 1.4 ms of a 2.3 ms call) and a prefix over the lanes gives each
// unit its start (placement of a group differs from launch to launch, a unit's content does not); pass 2 walks a unit's hits
// backwards so that every run start knows where the next run begins without any cross-lane memory traffic.
constexpr u32 HIT_RUNS_GROUP = 16;
__global__ __launch_bounds__(256) void hit_runs_kernel(const u32 *__restrict__ hits, const u64 *__restrict__ offsets, u32 nmates,
                                                       const u32 *__restrict__ n_hits, u64 n_units, u64 *__restrict__ run_start,
                                                       u32 *__restrict__ n_runs, u32 *__restrict__ run_tax, u32 *__restrict__ run_len,
                                                       unsigned long long *cursor)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    const u64 n_groups = (n_units + HIT_RUNS_GROUP - 1) / HIT_RUNS_GROUP;
    auto starts = [&](const u32 *h, u32 nh, u32 i0) -> u64 {
        const u32 i = i0 + lane;
        const bool st = i < nh && (i == 0 || h[i] != h[i - 1]);
        return __builtin_amdgcn_ballot_w64(st);
    };
    for (u64 g = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); g < n_groups; g += n_waves) {
        const u64 u0 = g * HIT_RUNS_GROUP;
        const u32 nu = (u32)(n_units - u0 < HIT_RUNS_GROUP ? n_units - u0 : HIT_RUNS_GROUP);
        u32 my_cnt = 0;                                          // lane j: runs of unit u0 + j
        for (u32 j = 0; j < nu; ++j) {
            const u32 *h = hits + offsets[(u0 + j) * nmates];
            const u32 nh = n_hits[u0 + j];
            u32 cnt = 0;
            for (u32 i0 = 0; i0 < nh; i0 += 64) cnt += (u32)__popcll(starts(h, nh, i0));
            if (lane == j) my_cnt = cnt;
        }
        u32 incl = my_cnt;
#pragma unroll
        for (u32 off = 1; off < HIT_RUNS_GROUP; off <<= 1) {
            const u32 t = (u32)__shfl_up((int)incl, (int)off);
            if (lane >= off) incl += t;
        }
        const u32 total = (u32)__shfl((int)incl, (int)HIT_RUNS_GROUP - 1);
        u64 rb = 0;
        if (lane == 0 && total) rb = atomicAdd(cursor, (unsigned long long)total);
        rb = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)rb);
        const u32 my_off = incl - my_cnt;
        if (lane < nu) { run_start[u0 + lane] = rb + my_off; n_runs[u0 + lane] = my_cnt; }
        for (u32 j = 0; j < nu; ++j) {
            const u32 *h = hits + offsets[(u0 + j) * nmates];
            const u32 nh = n_hits[u0 + j];
            const u32 cnt = (u32)__shfl((int)my_cnt, (int)j);
            const u64 ub = rb + (u32)__shfl((int)my_off, (int)j);
            u32 later = 0, next_start = nh;                      // runs in the chunks already done (behind us), first start among them
            for (u32 i0 = nh ? ((nh - 1) & ~63u) : 0; nh; i0 -= 64) {
                const u64 B = starts(h, nh, i0);
                const u32 i = i0 + lane;
                if ((B >> lane) & 1) {
                    const u64 above = lane == 63 ? 0ULL : (B & ~((2ULL << lane) - 1ULL));
                    const u32 nxt = above ? i0 + (u32)__builtin_ctzll(above) : next_start;
                    const u64 idx = ub + cnt - later - (u32)__popcll(B & ~((1ULL << lane) - 1ULL));
                    run_tax[idx] = h[i];
                    run_len[idx] = nxt - i;
                }
                if (B) { next_start = i0 + (u32)__builtin_ctzll(B); later += (u32)__popcll(B); }
                if (i0 == 0) break;
            }
        }
    }
}

// =====================================================================================================
// RollingHasher<u64, CyclicHash<u64>> without a window (encoder.h:644-865, rollinghash/cyclichash.h):
//   forward   H_j = rotl1(H_{j-1}) ^ rotl_{k%64}(T[s_{j-k}]) ^ T[s_j]
//   reverse   R_j = rotr1(R_{j-1} ^ rotl_{k%64}(Trc[rc s_j]) ^ Trc[rc s_{j-k}])        (canonical path: emits min(H, R))
// Both recurrences are linear over GF(2) up to a rotation, so rotating position j's term by -j (forward) / +(j-1)
// (reverse) turns them into plain prefix XORs: one wavefront per sequence scans 64 positions per step.  The start
// values per segment (a segment = the run of valid characters the reference restarts on, k + 1 characters after an
// invalid one) are a k-term XOR for H and, for R, the reference's fill that eats the window's LAST base's complement
// k times (encoder.h:721).  The 256-entry character tables are an argument (parity unpinned).
// =====================================================================================================
__device__ __forceinline__ u64 rotl64v(u64 x, u32 r) { r &= 63u; return r ? (x << r) | (x >> (64u - r)) : x; }
__device__ __forceinline__ u64 rotr64v(u64 x, u32 r) { r &= 63u; return r ? (x >> r) | (x << (64u - r)) : x; }
__device__ __forceinline__ u64 wave_xor_scan(u64 v)                     // inclusive prefix XOR over the wavefront
{
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 lo = (u32)__shfl_up((int)(u32)v, off), hi = (u32)__shfl_up((int)(u32)(v >> 32), off);
        if (lane >= off) v ^= ((u64)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ u64 wave_xor_all(u64 v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v ^= ((u64)(u32)__shfl_xor((int)(u32)(v >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)v, off);
    return v;
}

__global__ __launch_bounds__(256) void rolling_hash_kernel(const u8 *__restrict__ bases, const u64 *__restrict__ offsets, u64 n_seqs,
                                                           u32 k, int canon, int raw, const u64 *__restrict__ tf, const u64 *__restrict__ tr,
                                                           u64 *__restrict__ out, u32 *__restrict__ n_out)
{
    // raw (the input of stream_window_kernel): the canonical path stores BOTH strands' hashes, forward then reverse, two
    // entries per position (what the windowed RollingHasher queues, encoder.h:724-725,730-731) instead of their minimum
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    const u32 myr = k & 63u;
    const bool both = raw && canon;
    const u64 per = both ? 2 : 1;
    for (u64 q = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_seqs; q += n_waves) {
        const u8 *s = bases + offsets[q];
        const u64 l = offsets[q + 1] - offsets[q];
        u64 *o = out + per * offsets[q];
        u64 n = 0;                                                       // values emitted so far (wave-uniform)
        auto code_at = [&](u64 i, u32 &bad) -> u32 { return base_code(s[i], bad); };
        u64 r = 0;                                                       // segment start
        while (l >= k && r + k <= l) {
            // first invalid character at or after r
            u64 inv = l;
            for (u64 c0 = r; c0 < l && inv == l; c0 += 64) {
                u32 bad = 0;
                if (c0 + lane < l) (void)code_at(c0 + lane, bad);
                const u64 m = __builtin_amdgcn_ballot_w64(bad != 0);
                if (m) inv = c0 + (u64)__builtin_ctzll(m);
            }
            if (inv >= r + k) {                                          // the fill completes: values for j = r+k-1 .. inv-1
                const u64 j0 = r + k - 1;
                // H_{j0} = XOR_t rotl^{k-1-t}(T[s_{r+t}]);  R_{j0} = XOR_t rotl^{t}(Trc[rc s_{j0}])
                u64 h0 = 0, g0 = 0;
                u32 bad;
                const u64 tlast = canon ? tr[3u - code_at(j0, bad)] : 0ULL;
                for (u32 t = lane; t < k; t += 64) {
                    h0 ^= rotl64v(tf[code_at(r + t, bad)], k - 1u - t);
                    g0 ^= rotl64v(tlast, t);
                }
                h0 = wave_xor_all(h0); g0 = wave_xor_all(g0);
                if (lane == 0) { if (both) { o[2 * n] = h0; o[2 * n + 1] = g0; } else o[n] = canon ? (h0 < g0 ? h0 : g0) : h0; }
                // normalised running values: P = rotr^{j}(H_j), Q = rotl^{j}(R_j)
                u64 P = rotr64v(h0, (u32)j0), Q = rotl64v(g0, (u32)j0);
                for (u64 c0 = j0 + 1; c0 < inv; c0 += 64) {
                    const u64 j = c0 + lane;
                    u64 ef = 0, er = 0;
                    if (j < inv) {
                        const u32 cin = code_at(j, bad), cout = code_at(j - k, bad);
                        ef = rotr64v(rotl64v(tf[cout], myr) ^ tf[cin], (u32)j);
                        if (canon) er = rotl64v(rotl64v(tr[3u - cin], myr) ^ tr[3u - cout], (u32)(j - 1));
                    }
                    const u64 pf = P ^ wave_xor_scan(ef), qr = Q ^ wave_xor_scan(er);
                    if (j < inv) {
                        const u64 hj = rotl64v(pf, (u32)j), gj = rotr64v(qr, (u32)j);
                        const u64 at = n + 1 + (j - (j0 + 1));
                        if (both) { o[2 * at] = hj; o[2 * at + 1] = gj; } else o[at] = canon ? (hj < gj ? hj : gj) : hj;
                    }
                    P = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(pf >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)pf, 63);
                    Q = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(qr >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)qr, 63);
                }
                n += inv - j0;
            }
            if (inv >= l) break;
            if (canon && inv + 2 * (u64)k >= l) break;                   // encoder.h:714
            r = inv + (u64)k + 1;                                        // i += k_, then the loop's ++i
        }
        if (lane == 0) n_out[q] = (u32)(per * n);
    }
}

// RollingHasher<__uint128_t, CyclicHash<__uint128_t>> without a window (the instantiation test/encoding.cpp:152 constructs): the
// same two recurrences over a 128-bit word.  A value is a (lo, hi) pair of u64; rotations are 128-bit, myr = k % 128; the prefix
// XOR runs over both halves.  Tables: 256 entries x (lo, hi).  Same segment rules as rolling_hash_kernel.
struct W128 { u64 lo, hi; };
__device__ __forceinline__ W128 operator^(W128 a, W128 b) { return W128{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ bool less128(W128 a, W128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__device__ __forceinline__ W128 rotl128v(W128 x, u32 r)
{
    r &= 127u;
    if (r >= 64u) { const u64 t = x.lo; x.lo = x.hi; x.hi = t; r -= 64u; }
    if (r == 0u) return x;
    return W128{(x.lo << r) | (x.hi >> (64u - r)), (x.hi << r) | (x.lo >> (64u - r))};
}
__device__ __forceinline__ W128 rotr128v(W128 x, u32 r) { return rotl128v(x, 128u - (r & 127u)); }
__device__ __forceinline__ W128 ld128(const u64 *t, u32 i) { return W128{t[2u * i], t[2u * i + 1u]}; }
__device__ __forceinline__ u64 bcast63(u64 v)
{
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, 63);
}
// raw (the input of stream_window128_kernel): the canonical path stores BOTH strands' values, forward then reverse, as separate
// entries (the windowed hasher queues them one after the other, encoder.h:724-725); n_out counts entries.
__global__ __launch_bounds__(256) void rolling_hash128_kernel(const u8 *__restrict__ bases, const u64 *__restrict__ offsets, u64 n_seqs,
                                                              u32 k, int canon, int raw, const u64 *__restrict__ tf, const u64 *__restrict__ tr,
                                                              u64 *__restrict__ out, u32 *__restrict__ n_out)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    const u32 myr = k & 127u;
    const bool both = raw && canon;
    for (u64 q = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_seqs; q += n_waves) {
        const u8 *s = bases + offsets[q];
        const u64 l = offsets[q + 1] - offsets[q];
        u64 *o = out + (both ? 4 : 2) * offsets[q];
        u64 n = 0;
        auto code_at = [&](u64 i, u32 &bad) -> u32 { return base_code(s[i], bad); };
        auto put = [&](u64 at, W128 h, W128 g) {
            if (both) { o[4 * at] = h.lo; o[4 * at + 1] = h.hi; o[4 * at + 2] = g.lo; o[4 * at + 3] = g.hi; return; }
            const W128 v = canon ? (less128(h, g) ? h : g) : h; o[2 * at] = v.lo; o[2 * at + 1] = v.hi;
        };
        u64 r = 0;
        while (l >= k && r + k <= l) {
            u64 inv = l;
            for (u64 c0 = r; c0 < l && inv == l; c0 += 64) {
                u32 bad = 0;
                if (c0 + lane < l) (void)code_at(c0 + lane, bad);
                const u64 m = __builtin_amdgcn_ballot_w64(bad != 0);
                if (m) inv = c0 + (u64)__builtin_ctzll(m);
            }
            if (inv >= r + k) {
                const u64 j0 = r + k - 1;
                W128 h0{0, 0}, g0{0, 0};
                u32 bad;
                const W128 tlast = canon ? ld128(tr, 3u - code_at(j0, bad)) : W128{0, 0};
                for (u32 t = lane; t < k; t += 64) {
                    h0 = h0 ^ rotl128v(ld128(tf, code_at(r + t, bad)), k - 1u - t);
                    g0 = g0 ^ rotl128v(tlast, t);
                }
                h0 = W128{wave_xor_all(h0.lo), wave_xor_all(h0.hi)}; g0 = W128{wave_xor_all(g0.lo), wave_xor_all(g0.hi)};
                if (lane == 0) put(n, h0, g0);
                W128 P = rotr128v(h0, (u32)j0), Q = rotl128v(g0, (u32)j0);
                for (u64 c0 = j0 + 1; c0 < inv; c0 += 64) {
                    const u64 j = c0 + lane;
                    W128 ef{0, 0}, er{0, 0};
                    if (j < inv) {
                        const u32 cin = code_at(j, bad), cout = code_at(j - k, bad);
                        ef = rotr128v(rotl128v(ld128(tf, cout), myr) ^ ld128(tf, cin), (u32)j);
                        if (canon) er = rotl128v(rotl128v(ld128(tr, 3u - cin), myr) ^ ld128(tr, 3u - cout), (u32)(j - 1));
                    }
                    const W128 pf = P ^ W128{wave_xor_scan(ef.lo), wave_xor_scan(ef.hi)}, qr = Q ^ W128{wave_xor_scan(er.lo), wave_xor_scan(er.hi)};
                    if (j < inv) put(n + 1 + (j - (j0 + 1)), rotl128v(pf, (u32)j), rotr128v(qr, (u32)j));
                    P = W128{bcast63(pf.lo), bcast63(pf.hi)};
                    Q = W128{bcast63(qr.lo), bcast63(qr.hi)};
                }
                n += inv - j0;
            }
            if (inv >= l) break;
            if (canon && inv + 2 * (u64)k >= l) break;                   // encoder.h:714
            r = inv + (u64)k + 1;
        }
        if (lane == 0) n_out[q] = (u32)(both ? 2 * n : n);
    }
}

// QueueMap over a finished stream of 128-bit values ((lo, hi) pairs): stream_window_kernel for RollingHasher<__uint128_t> with a
// window.  score = lex_score128 (the reference's is sketch's CEHasher, un-vendored: restated, parity unpinned -- the number of
// values, which is what test/encoding.cpp:152-156 pins, does not depend on it).  `per` = entries per base of the buffer layout.
__device__ __forceinline__ u64 lex_score128(u64 lo, u64 hi) { return kmer_score(lo ^ kmer_score(hi, 0), 0); }
__global__ __launch_bounds__(256) void stream_window128_kernel(const u64 *__restrict__ in, const u32 *__restrict__ n_in, const u64 *__restrict__ offsets,
                                                               u64 n_seqs, u32 per, u32 ws, u64 *__restrict__ out, u32 *__restrict__ n_out)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    for (u64 q = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_seqs; q += n_waves) {
        const u64 *v = in + 2ULL * per * offsets[q];
        u64 *o = out + 2ULL * per * offsets[q];
        const u32 n = n_in[q];
        u32 emitted = 0;
        auto less = [](u64 sa, u64 alo, u64 ahi, u64 sb, u64 blo, u64 bhi) { return sa < sb || (sa == sb && (ahi < bhi || (ahi == bhi && alo < blo))); };
        if (n >= ws) {
            const u32 nw = n - ws + 1u;
            for (u32 i0 = 0; i0 < nw; i0 += 64u) {
                const u32 i = i0 + lane;
                u64 blo = ~0ULL, bhi = ~0ULL, bs = ~0ULL;
                if (i < nw) {
                    blo = v[2 * i]; bhi = v[2 * i + 1]; bs = lex_score128(blo, bhi);
                    for (u32 j = 1; j < ws; ++j) {
                        const u64 elo = v[2 * (i + j)], ehi = v[2 * (i + j) + 1], sc = lex_score128(elo, ehi);
                        if (less(sc, elo, ehi, bs, blo, bhi)) { bs = sc; blo = elo; bhi = ehi; }
                    }
                }
                const bool on = i < nw && !(blo == ~0ULL && bhi == ~0ULL);
                const u64 vm = ballot64(on);
                if (on) { const u32 at = emitted + (u32)__popcll(vm & lanemask_lt()); o[2 * at] = blo; o[2 * at + 1] = bhi; }
                emitted += (u32)__popcll(vm);
            }
        } else if (n) {
            if (lane == 0) {                                   // (a stream shorter than its window: rare and short -- one lane)
                u64 blo = v[0], bhi = v[1], bs = lex_score128(blo, bhi);
                for (u32 i = 1; i < n; ++i) {
                    const u64 elo = v[2 * i], ehi = v[2 * i + 1], sc = lex_score128(elo, ehi);
                    if (less(sc, elo, ehi, bs, blo, bhi)) { bs = sc; blo = elo; bhi = ehi; }
                }
                o[0] = blo; o[1] = bhi;                        // (max_in_queue().el_ is emitted as is)
            }
            emitted = 1;
        }
        if (lane == 0) n_out[q] = emitted;
    }
}

// =====================================================================================================
// Encoder::for_each_hash (encoder.h:355-394): ntHash of every k-window the reference's loop visits.  One wavefront per
// sequence.  The loop, in closed form: for every maximal run [a, b) of A/C/G/T (either case; a NUL byte ends the string,
// encoder.h:371-377) with b - a >= k, the windows a .. b-k in order -- except that a run whose FIRST window ends exactly at
// the end of the string emits nothing (`if(*p2 == 0) return;` is tested before `p2 - p == k`, encoder.h:378-379).
// NTC64 (bcgsc/ntHash, un-vendored: restated from the published definition, Mohamadi et al. 2016, as ntHash 1.0.x ships it):
//   forward  fh(w) = XOR_j rol^{k-1-j}(T[s_j])      rolling  fh' = rol1(fh) ^ rol^k(T[out]) ^ T[in]
//   reverse  rh(w) = XOR_j rol^{j}(T[s_j & 7])      rolling  rh' = ror1(rh ^ T[out & 7] ^ rol^k(T[in & 7]))
//   (the complement's seed sits at index c & 7 of the same table: A&7 = 1 holds T's seed, C&7 = 3 G's, G&7 = 7 C's,
//    T&7 = 4 A's -- the geometry make_nthash_lut builds in-tree, encoder.h:93-103);  canonical value = min(fh, rh).
// Both rolling forms are prefix XORs after rotating window i's term by -i / +i, as in rolling_hash_kernel.
// The 256-entry table is an argument (parity unpinned: SURVEY F10).
// =====================================================================================================
__global__ __launch_bounds__(256) void nthash_kernel(const u8 *__restrict__ bases, const u64 *__restrict__ offsets, u64 n_seqs,
                                                     u32 k, int canon, const u64 *__restrict__ T, u64 *__restrict__ out, u32 *__restrict__ n_out)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    const u32 kr = k & 63u;
    for (u64 q = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_seqs; q += n_waves) {
        const u8 *s = bases + offsets[q];
        u64 l = offsets[q + 1] - offsets[q];
        u64 *o = out + offsets[q];
        u64 n = 0;
        auto invalid = [&](u64 i) -> bool { u32 bad; (void)base_code(s[i], bad); return bad != 0; };
        u64 a = 0;                                                       // scan position
        while (a + k <= l) {
            // first valid character at or after a (a NUL ends the string), then the end of its run
            u64 b = l;
            bool found_start = false;
            for (u64 c0 = a; c0 < l; c0 += 64) {
                const bool in = c0 + lane < l;
                const u8 ch = in ? s[c0 + lane] : (u8)1;
                const u64 nul = __builtin_amdgcn_ballot_w64(in && ch == 0);
                u64 inv = __builtin_amdgcn_ballot_w64(in && invalid(c0 + lane));
                if (nul) { const int z = __builtin_ctzll(nul); l = c0 + (u64)z; inv &= (1ULL << z) - 1ULL; }
                const u64 within = (l - c0 >= 64) ? ~0ULL : ((1ULL << (l - c0)) - 1ULL);
                if (!found_start) {
                    const u64 ok = ~inv & within;
                    if (!ok) { if (nul) break; continue; }
                    const int f = __builtin_ctzll(ok);
                    a = c0 + (u64)f; found_start = true;
                    inv &= ~((2ULL << f) - 1ULL);                        // invalid characters after the start only
                }
                if (inv & within) { b = c0 + (u64)__builtin_ctzll(inv & within); break; }
                if (nul) break;
            }
            if (!found_start) break;
            if (b > l) b = l;
            if (b - a >= (u64)k && !(a + k == l)) {
                u64 fh = 0, rh = 0;
                for (u32 t = lane; t < k; t += 64) {
                    const u8 ch = s[a + t];
                    fh ^= rotl64v(T[ch], k - 1u - t);
                    rh ^= rotl64v(T[ch & 7u], t);
                }
                fh = wave_xor_all(fh); rh = wave_xor_all(rh);
                if (lane == 0) o[n] = canon ? (rh < fh ? rh : fh) : fh;
                u64 P = rotr64v(fh, (u32)a), Q = rotl64v(rh, (u32)a);
                const u64 last = b - k;                                  // last window of the run
                for (u64 c0 = a + 1; c0 <= last; c0 += 64) {
                    const u64 i = c0 + lane;
                    u64 ef = 0, er = 0;
                    if (i <= last) {
                        const u8 cout = s[i - 1], cin = s[i + k - 1];
                        ef = rotr64v(rotl64v(T[cout], kr) ^ T[cin], (u32)i);
                        er = rotl64v(T[cout & 7u] ^ rotl64v(T[cin & 7u], kr), (u32)(i - 1));
                    }
                    const u64 pf = P ^ wave_xor_scan(ef), qr = Q ^ wave_xor_scan(er);
                    if (i <= last) {
                        const u64 f = rotl64v(pf, (u32)i), r = rotr64v(qr, (u32)i);
                        o[n + (i - a)] = canon ? (r < f ? r : f) : f;
                    }
                    P = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(pf >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)pf, 63);
                    Q = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(qr >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)qr, 63);
                }
                n += last - a + 1;
            }
            if (b >= l) break;
            a = b + 1;
        }
        if (lane == 0) n_out[q] = (u32)n;
    }
}

// QueueMap over a finished stream (qmap.h:79-87 as RollingHasher uses it, encoder.h:706-710,771-776,735-736,794-795): window i
// = entries i .. i+ws-1 of sequence q's stream, its value the entry with the smallest (lex_score(v), v); a value equal to
// ENCODE_OVERFLOW is not emitted; a stream shorter than the window gives one value, its minimum.  `per` = entries per base
// the stream buffers are laid out with (sequence q starts at per * offsets[q]).  One wavefront per sequence.
__global__ __launch_bounds__(256) void stream_window_kernel(const u64 *__restrict__ in, const u32 *__restrict__ n_in, const u64 *__restrict__ offsets,
                                                            u64 n_seqs, u32 per, u32 ws, u64 *__restrict__ out, u32 *__restrict__ n_out)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 n_waves = (u64)gridDim.x * 4;
    for (u64 q = (u64)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_seqs; q += n_waves) {
        const u64 *v = in + (u64)per * offsets[q];
        u64 *o = out + (u64)per * offsets[q];
        const u32 n = n_in[q];
        u32 emitted = 0;
        auto less = [](u64 sa, u64 a, u64 sb, u64 b) { return sa < sb || (sa == sb && a < b); };
        if (n >= ws) {
            const u32 nw = n - ws + 1u;
            for (u32 i0 = 0; i0 < nw; i0 += 64u) {
                const u32 i = i0 + lane;
                u64 be = ~0ULL, bs = ~0ULL;
                if (i < nw) {
                    be = v[i]; bs = kmer_score(be, 0);
                    for (u32 j = 1; j < ws; ++j) {
                        const u64 e = v[i + j], sc = kmer_score(e, 0);
                        if (less(sc, e, bs, be)) { bs = sc; be = e; }
                    }
                }
                const bool on = i < nw && be != ~0ULL;
                const u64 vm = ballot64(on);
                if (on) o[emitted + (u32)__popcll(vm & lanemask_lt())] = be;
                emitted += (u32)__popcll(vm);
            }
        } else if (n) {
            u64 be = ~0ULL, bs = ~0ULL;
            bool have = false;
            for (u32 i = lane; i < n; i += 64u) {
                const u64 e = v[i], sc = kmer_score(e, 0);
                if (!have || less(sc, e, bs, be)) { bs = sc; be = e; have = true; }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                const u64 oe = ((u64)(u32)__shfl_xor((int)(u32)(be >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)be, off);
                const u64 os = ((u64)(u32)__shfl_xor((int)(u32)(bs >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)bs, off);
                const bool oh = __shfl_xor((int)have, off) != 0;
                if (oh && (!have || less(os, oe, bs, be))) { bs = os; be = oe; have = true; }
            }
            if (lane == 0) o[0] = be;                          // (max_in_queue().el_ is emitted as is, even ~0)
            emitted = 1;
        }
        if (lane == 0) n_out[q] = emitted;
    }
}

__global__ __launch_bounds__(256) void fill_u64_kernel(u64 *p, u64 n, u64 v)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void fill_u32_kernel(u32 *p, u64 n, u32 v)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ __launch_bounds__(256) void max_len_kernel(const u64 *__restrict__ offsets, u64 n_reads, u32 *out)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 m = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += stride) {
        const u32 L = (u32)(offsets[i + 1] - offsets[i]);
        m = L > m ? L : m;
    }
    for (int off = 32; off >= 1; off >>= 1) { const u32 o = (u32)__shfl_xor((int)m, off); m = o > m ? o : m; }
    if (lane_id() == 0 && m) atomicMax(out, m);
}

// resolve_tree over a batch of explicit counters (bns_resolve_batch).  One wavefront per unit; the
// counter is read in place, the interval scratch lives at the unit's own offset.
__global__ __launch_bounds__(64) void resolve_kernel(const u32 *__restrict__ keys, const u32 *__restrict__ counts,
                                                     const u64 *__restrict__ starts, u64 n_units, u32 *scratch, u64 total,
                                                     const TaxNode *__restrict__ nodes, u32 n_nodes, u32 *__restrict__ taxon)
{
    for (u64 u = blockIdx.x; u < n_units; u += gridDim.x) {
        const u64 b0 = starts[u], b1 = starts[u + 1];
        const u32 D = (u32)(b1 - b0);
        u32 t;
        if (D >= 2u && D <= 64u) {                                // the register path classify_kernel takes for these sizes
            const u32 lane = (u32)lane_id();
            t = resolve_regs(lane < D ? keys[b0 + lane] : 0u, lane < D ? counts[b0 + lane] : 0u, D, nodes, n_nodes);
        } else t = resolve_wave(keys + b0, counts + b0, scratch + b0, scratch + total + b0, D, nodes, n_nodes);
        if (lane_id() == 0) taxon[u] = t;
    }
}

// ---- explicit instantiations used by the host side ------------------------------------------------------
#define BNS_INST(SP, LY)                                                                            \
    template __global__ void classify_kernel<SP, LY, 0, 0>(ClassifyParams);                            \
    template __global__ void classify_overflow_kernel<SP, LY>(ClassifyParams, u32 *, u64);
BNS_INST(false, 0) BNS_INST(false, 1) BNS_INST(true, 0) BNS_INST(true, 1) BNS_INST(false, 2) BNS_INST(true, 2)
#undef BNS_INST
template __global__ void classify_kernel<false, 2, 0, 0, 8, false, true>(ClassifyParams);
template __global__ void classify_overflow_kernel<false, 2, true>(ClassifyParams, u32 *, u64);
// packed input: the generic kernels of every layout (the k = 31 ones are instantiated where they are launched)
#define BNS_INSTP(SP, LY)                                                                           \
    template __global__ void classify_kernel<SP, LY, 0, 0, 8, false, false, true>(ClassifyParams);     \
    template __global__ void classify_overflow_kernel<SP, LY, false, true>(ClassifyParams, u32 *, u64);
BNS_INSTP(false, 0) BNS_INSTP(false, 1) BNS_INSTP(true, 0) BNS_INSTP(true, 1) BNS_INSTP(false, 2) BNS_INSTP(true, 2)
#undef BNS_INSTP
template __global__ void classify_kernel<false, 2, 0, 0, 8, false, true, true>(ClassifyParams);
template __global__ void classify_overflow_kernel<false, 2, true, true>(ClassifyParams, u32 *, u64);
template __global__ void encode_kernel<false>(ClassifyParams, u64 *, u32 *);
template __global__ void encode_kernel<true>(ClassifyParams, u64 *, u32 *);
template __global__ void probe_kernel<0>(ClassifyParams, const u64 *, u64, u32 *, u8 *);
template __global__ void probe_kernel<1>(ClassifyParams, const u64 *, u64, u32 *, u8 *);
template __global__ void probe_kernel<2>(ClassifyParams, const u64 *, u64, u32 *, u8 *);
template __global__ void probe_kernel<2, 1>(ClassifyParams, const u64 *, u64, u32 *, u8 *);
template __global__ void probe_kernel<2, 2>(ClassifyParams, const u64 *, u64, u32 *, u8 *);
template __global__ void build_kernel<false, 1>(ClassifyParams, const u32 *, u64, u64 *, u32 *, unsigned long long *);
template __global__ void build_kernel<false, 2>(ClassifyParams, const u32 *, u64, u64 *, u32 *, unsigned long long *);
template __global__ void build_kernel<true, 1>(ClassifyParams, const u32 *, u64, u64 *, u32 *, unsigned long long *);
template __global__ void build_kernel<true, 2>(ClassifyParams, const u32 *, u64, u64 *, u32 *, unsigned long long *);

}  // namespace bns
