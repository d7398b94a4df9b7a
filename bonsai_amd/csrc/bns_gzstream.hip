// bns_gzstream.hip -- ONE plain gzip stream inflated on the device (included at the end of bns_inflate.hip: same handle, same stream).
//
// What it replaces in the reference: gzread under kseq (kseq_declare.h:112-145, klib/kseq.h:177-225) -- one zlib inflate, ~0.4 GB/s of
// text.  DEFLATE has no sync points and every block may copy from the 32 KiB in front of it, so a stream is serial by its definition;
// but a dynamic-Huffman block header is redundant enough to be FOUND (the scheme of pugz / rapidgzip, which csrc/host/pgzip.cpp runs on
// CPU threads; this is the same idea laid out for 4096 wavefronts):
//   1. gz_search_kernel   the call's compressed bytes are cut into chunks of CH bytes; one wavefront per chunk looks, bit position by
//                         bit position (a lane each), for a header that is not the member's last, is dynamic, and whose code lengths
//                         form a COMPLETE code-length code, a complete literal/length code with an end-of-block symbol, and a complete
//                         distance code.  A true header that fails the test (a block with one distance code, say) only makes the
//                         chunk in front longer; a false one is caught below.
//   2. gz_plan_kernel     the chunks that found one, in order: each decodes from its header to the next chunk's.
//   3. gz_decode_kernel   bns_inflate_wave.hpp's decoder, one chunk per wavefront, writing 16-bit SYMBOLS: a byte, or a marker
//                         0x8000 | j for "byte j of the 32 KiB in front of my entry point" -- the 32768 markers themselves lie in front
//                         of the chunk's output, so a match that reaches back there copies them like any other text.  Chunk 0 starts
//                         where the caller says a block starts (the member's first block, or where the call before ended) and has
//                         the real 32 KiB in front of it.
//   4. gz_valid_kernel    which chunks count: chunk c only if the one in front ended EXACTLY at its header (the decoder stops at the
//                         first block boundary at or behind the next header: landing behind it means that header was no header), its
//                         text fits, and no member ended in front of it.  Where each chunk's text goes (a prefix sum).
//      gz_compose_kernel  what the markers mean.  The 32 KiB behind chunk c are its last 32768 symbols looked up in the 32 KiB in front of
//      gz_groups_kernel   it: a chain over the chunks, 15 us a step when one block walks it (60 ms per 4 k chunks: measured) -- but "look
//                         up in the window in front" COMPOSES: groups of G chunks are walked side by side, each from the identity (its
//                         step k leaves P_k: symbols that say what byte j in front of chunk k is in terms of the window in front of the
//                         GROUP), then one block walks the groups' windows (n / G steps).
//   5. gz_translate_kernel  symbols -> bytes: a marker of chunk k through P_k, and what is still a marker through its group's window;
//                         compacted into the caller's text buffer.
//   6. gz_crc_kernel      CRC-32 of every chunk's text in four parts (64 slices each), folded into the call's (zlib's x^n mod p arithmetic).
// A call returns the text of the chunks that chained, where that text ends in the stream (a block header: the next call's entry point),
// the 32 KiB behind it, and whether the member ended there (the caller checks CRC-32 and ISIZE against the trailer and looks for the
// next member).  The chunks behind a break are simply decoded again by the next call, from a header that is known to be one; a call
// whose chunk 0 makes no progress reports why, and the caller goes back to the host inflater.
namespace gzs {
using bns_inf::u8;
using bns_inf::u16;
using bns_inf::u32;
using bns_inf::u64;

constexpr u32 WINDOW = 32768u;
constexpr u32 NONE32 = 0xFFFFFFFFu;
constexpr u64 NONE64 = ~0ULL;

struct ChunkOut {
    u32 n_out;           // symbols behind the prefix that belong to whole blocks
    u32 status;          // INF_OK when n_out > 0 or the decoder stopped at a boundary
    u64 end_bit;         // absolute bit position (in the call's compressed bytes) behind the last whole block
    u32 member_end;
    u32 pad;
};

struct CallOut {         // what comes back to the host
    u64 text_bytes;
    u64 end_bit;
    u32 member_end;
    u32 crc;
    u32 n_chunks;        // chunks that found a header (chunk 0 included)
    u32 n_good;          // ... that chained
    u32 status0;         // chunk 0's status
    u32 stop_why;        // 0: all chunks chained; 1: a chunk did not end at the next header; 2: text room; 3: member end; 4: a chunk failed
};

// 64 bits of the stream at bit position p (the buffer is padded: reads behind the end see zeros)
__device__ __forceinline__ u64 bits_at(const u8 *comp, u64 p)
{
    const u8 *a = comp + (p >> 3);
    const u32 s = (u32)p & 7u;
    const u64 lo = bns_inf::load64u(a);
    const u64 hi = a[8];
    return (lo >> s) | ((hi << 1) << (63u - s));
}

// Is there a block header at bit p that passes the test above?  One lane's work (divergent among the lanes of a wavefront: few get
// past the first look).  tab: 128 bytes of LDS of this lane.
__device__ __forceinline__ bool header_at(const u8 *comp, u64 p, u64 total_bits, u8 *tab)
{
    u64 v = bits_at(comp, p);
    // BFINAL 0, BTYPE 2 (bits 1-2 = 10b LSB first: value 2)
    if ((v & 7u) != 4u) return false;
    const u32 hlit = ((u32)(v >> 3) & 31u) + 257u, hdist = ((u32)(v >> 8) & 31u) + 1u, hclen = ((u32)(v >> 13) & 15u) + 4u;
    if (hlit > 286u || hdist > 30u) return false;
    // the code-length code: 3 bits each, in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
    u64 pos = p + 17u;
    v = bits_at(comp, pos);                                   // (19 x 3 = 57 bits: one look)
    u64 clpack = 0ULL;                                        // 3 bits per symbol 0..18
    u32 kraft = 0u;                                           // in units of 2^-7
    for (u32 i = 0u; i < hclen; ++i) {
        const u32 j = i - 4u;
        const u32 ord = i < 3u ? 16u + i : (i == 3u ? 0u : ((j & 1u) ? 7u - (j >> 1) : 8u + (j >> 1)));
        const u32 l = (u32)(v >> (3u * i)) & 7u;
        clpack |= (u64)l << (3u * ord);
        if (l) kraft += 128u >> l;
    }
    if (kraft != 128u) return false;
    pos += 3u * hclen;
    // its direct table (7 bits): symbol << 3 | length
    {
        u32 cnt[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) cnt[l] = 0u;
        for (u32 sy = 0u; sy < 19u; ++sy) {
            const u32 l = (u32)(clpack >> (3u * sy)) & 7u;
#pragma unroll
            for (int k = 1; k < 8; ++k) cnt[k] += l == (u32)k ? 1u : 0u;
        }
        u32 next[8];
        u32 code = 0u;
        next[0] = 0u;
#pragma unroll
        for (int l = 1; l < 8; ++l) { code = (code + cnt[l - 1]) << 1; next[l] = code; }       // (cnt[0] stays 0: unused symbols are not counted)
        for (u32 sy = 0u; sy < 19u; ++sy) {
            const u32 l = (u32)(clpack >> (3u * sy)) & 7u;
            if (!l) continue;
            u32 cd = 0u;
#pragma unroll
            for (int k = 1; k < 8; ++k) if (l == (u32)k) { cd = next[k]; next[k] = cd + 1u; }
            const u32 r = bns_inf::brev32(cd) >> (32u - l);
            for (u32 j = r; j < 128u; j += 1u << l) tab[j] = (u8)((sy << 3) | l);
        }
    }
    // the code lengths of the two codes: Kraft sums on the way (units of 2^-15), nothing stored
    const u32 total = hlit + hdist;
    u32 i = 0u, prev = 0u, lit_k = 0u, dist_k = 0u;
    bool eob = false;
    while (i < total) {
        if (pos + 14u > total_bits) return false;
        v = bits_at(comp, pos);
        const u32 e = tab[(u32)v & 127u];
        const u32 cl = e & 7u, sy = e >> 3;
        pos += cl; v >>= cl;
        u32 rep = 1u, val = sy;
        if (sy == 16u) { if (i == 0u) return false; val = prev; rep = 3u + ((u32)v & 3u); pos += 2u; }
        else if (sy == 17u) { val = 0u; rep = 3u + ((u32)v & 7u); pos += 3u; }
        else if (sy == 18u) { val = 0u; rep = 11u + ((u32)v & 127u); pos += 7u; }
        if (i + rep > total) return false;
        if (val) {
            const u32 in_lit = i < hlit ? min(rep, hlit - i) : 0u;
            lit_k += in_lit * (32768u >> val);
            dist_k += (rep - in_lit) * (32768u >> val);
            if (i <= 256u && 256u < i + rep) eob = true;
            if (lit_k > 32768u || dist_k > 32768u) return false;
        }
        prev = val;
        i += rep;
    }
    return eob && lit_k == 32768u && dist_k == 32768u;
}

// One wavefront per chunk c >= 1: the first bit position in [c * CH * 8, (c + 1) * CH * 8) that passes -> start[c] (NONE64: none).
// Three sieves, each run by 64 lanes on 64 positions that got through the one in front (positions wait in LDS queues until there are 64;
// a wavefront pays for an instruction whether one lane needs it or all): the three header bits and the two counts, from a 2 KiB slab of
// the stream in LDS (one position in 9 passes; ~15 instructions per 64 positions); the Kraft sum of the code-length code (one in 250 of
// those; ~120 instructions per 64 candidates); header_at (a table, up to 316 code lengths).  Measured on 64 KiB chunks of a FASTQ stream:
// 5.8 ms per 3962 chunks with every lane doing all of it for its own position, 4.1 with the third sieve queued, 2.1 with the second.
constexpr u32 SLAB_BITS = 16384u;
__global__ __launch_bounds__(64) void gz_search_kernel(const u8 *__restrict__ comp, u64 comp_bytes, u32 ch_bytes, u32 n_chunks, u64 first_bit, u64 *__restrict__ start)
{
    __shared__ u8 tabs[64][128];
    __shared__ u32 slab[SLAB_BITS / 32u + 4u];
    __shared__ u32 q1[128], q2[128];
    const u32 c = blockIdx.x + 1u;
    if (c >= n_chunks) return;
    const u32 lane = threadIdx.x;
    const u64 total_bits = comp_bytes * 8ULL;
    const u64 lo = (first_bit & ~7ULL) + (u64)c * ch_bytes * 8ULL, hi = min(lo + (u64)ch_bytes * 8ULL, total_bits);
    const u32 *words = reinterpret_cast<const u32 *>(comp);   // (the call's bytes start at an aligned address; the buffer has 4 KiB of room behind them)
    u64 found = NONE64;
    u32 n1 = 0u, n2 = 0u;
    u64 s0 = lo & ~31ULL;
    // the third sieve on the first n positions of q2 (relative to lo), a lane each -> the first one that passes
    auto sieve3 = [&](u32 n) {
        bool ok = false;
        if (lane < n) ok = header_at(comp, lo + q2[lane], total_bits, tabs[lane]);
        const u64 m = __builtin_amdgcn_ballot_w64(ok);
        if (m) found = lo + q2[(u32)__builtin_ctzll(m)];
    };
    // the second on the first n positions of q1 (relative to the slab): what passes goes to q2
    auto sieve2 = [&](u32 n) {
        bool cand = false;
        u32 rel = 0u;
        if (lane < n) {
            rel = q1[lane];
            const u32 w = rel >> 5, sh = rel & 31u;
            const u32 x0 = slab[w], x1 = slab[w + 1u], x2 = slab[w + 2u], x3 = slab[w + 3u];
            const u64 lo64 = ((u64)x1 << 32) | x0, mid64 = ((u64)x2 << 32) | x1, hi64 = ((u64)x3 << 32) | x2;
            const u32 hclen = ((u32)(lo64 >> (sh + 13u)) & 15u) + 4u;        // (sh + 13 + 4 <= 48)
            // the code-length code's lengths: 3 bits each from bit 17 on (57 bits at most): bits [r2, r2 + 57) of x0..x3
            const u32 r2 = sh + 17u;
            const u64 c3 = r2 < 32u ? ((lo64 >> r2) | ((mid64 >> r2) << 32)) : ((mid64 >> (r2 - 32u)) | ((hi64 >> (r2 - 32u)) << 32));
            u32 kraft = 0u;
#pragma unroll
            for (u32 i = 0u; i < 19u; ++i) {
                const u32 l = (u32)(c3 >> (3u * i)) & 7u;
                kraft += (i < hclen && l) ? (128u >> l) : 0u;
            }
            cand = kraft == 128u;
        }
        const u64 m = __builtin_amdgcn_ballot_w64(cand);
        if (m) {
            if (cand) q2[n2 + bns_infw::below(m)] = (u32)(s0 + rel - lo);
            n2 += (u32)__builtin_popcountll(m);
            __syncthreads();
            if (n2 >= 64u) {
                sieve3(64u);
                __syncthreads();
                const u32 rest = lane + 64u < n2 ? q2[lane + 64u] : 0u;
                __syncthreads();
                q2[lane] = rest;
                n2 -= 64u;
                __syncthreads();
            }
        }
    };
    for (; s0 < hi && found == NONE64; s0 += SLAB_BITS) {
        __syncthreads();
        const u64 w0 = s0 >> 5;
        for (u32 i = lane; i < SLAB_BITS / 32u + 4u; i += 64u) slab[i] = words[w0 + i];
        __syncthreads();
        const u64 e0 = min(hi, s0 + (u64)SLAB_BITS);
        for (u64 p0 = max(lo, s0); p0 < e0 && found == NONE64; p0 += 64u) {
            const u64 p = p0 + lane;
            bool pass = false;
            const u32 rel = (u32)(p - s0);
            if (p < e0 && p + 80u < total_bits) {
                const u32 w = rel >> 5, sh = rel & 31u;
                const u64 lo64 = ((u64)slab[w + 1u] << 32) | slab[w];
                const u32 v = (u32)(lo64 >> sh);
                pass = (v & 7u) == 4u && ((v >> 3) & 31u) <= 29u && ((v >> 8) & 31u) <= 29u;
            }
            const u64 m = __builtin_amdgcn_ballot_w64(pass);
            if (m) {
                if (pass) q1[n1 + bns_infw::below(m)] = rel;
                n1 += (u32)__builtin_popcountll(m);
                __syncthreads();
                if (n1 >= 64u) {
                    sieve2(64u);
                    __syncthreads();
                    const u32 rest = lane + 64u < n1 ? q1[lane + 64u] : 0u;
                    __syncthreads();
                    q1[lane] = rest;
                    n1 -= 64u;
                    __syncthreads();
                }
            }
        }
        if (n1 && found == NONE64) { __syncthreads(); sieve2(n1); __syncthreads(); }      // (the slab goes: its candidates now)
        n1 = 0u;
    }
    if (found == NONE64 && n2) { __syncthreads(); sieve3(n2); }
    if (lane == 0u) start[c] = found < hi ? found : NONE64;
}

// The chunks with a header, in order (one wavefront, 64 chunks a step).  entry[k] = header position, stop[k] = the next one's.
__global__ __launch_bounds__(64) void gz_plan_kernel(const u64 *__restrict__ start, u32 n_chunks, u64 *__restrict__ entry, u64 *__restrict__ stop, u32 *__restrict__ n_entries)
{
    const u32 lane = threadIdx.x;
    u32 k = 0u;
    for (u32 c0 = 0u; c0 < n_chunks; c0 += 64u) {
        const u32 c = c0 + lane;
        const u64 s = c < n_chunks ? start[c] : NONE64;
        const u64 m = __builtin_amdgcn_ballot_w64(s != NONE64);
        if (s != NONE64) entry[k + bns_infw::below(m)] = s;
        k += (u32)__builtin_popcountll(m);
    }
    __threadfence();
    __syncthreads();
    for (u32 j = lane; j < k; j += 64u) stop[j] = j + 1u < k ? entry[j + 1u] : NONE64;
    if (lane == 0u) *n_entries = k;
}

// One wavefront per entry: symbols into sym[k * stride ...): WINDOW elements of prefix, then the text.
__global__ __launch_bounds__(64) void gz_decode_kernel(const u8 *__restrict__ comp, const u8 *__restrict__ comp_end, u64 comp_bytes, const u64 *__restrict__ entry,
                                                       const u64 *__restrict__ stop, const u32 *__restrict__ n_entries, const u8 *__restrict__ window0, u16 *sym,
                                                       u64 stride, u64 text_cap, ChunkOut *__restrict__ res)
{
    __shared__ bns_infw::WaveLdsT<u16> S;
    const u32 k = blockIdx.x;
    if (k >= *n_entries) return;
    const u32 lane = threadIdx.x;
    u16 *out = sym + (u64)k * stride;
    // what lies in front: the real bytes for entry 0, markers for the others
    if (k == 0u) {
        for (u32 j = lane * 8u; j < WINDOW; j += 512u) {
            u16 v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = window0 ? (u16)window0[j + b] : (u16)0;
            __builtin_memcpy(out + j, v, 16);
        }
    } else {
        for (u32 j = lane * 8u; j < WINDOW; j += 512u) {
            u16 v[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) v[b] = (u16)(0x8000u | (j + b));
            __builtin_memcpy(out + j, v, 16);
        }
    }
    __syncthreads();
    const u64 e = entry[k], st = stop[k];
    const u64 byte0 = e >> 3;
    bns_infw::StreamIO io;
    io.bit0 = (u32)e & 7u;
    const u64 rel_stop = st == NONE64 ? (u64)NONE32 : st - byte0 * 8ULL;
    io.stop_bit = rel_stop >= (u64)NONE32 ? NONE32 : (u32)rel_stop;
    io.prefix = WINDOW;
    io.end_bit = 0u; io.member_end = 0u;
    const u64 left = comp_bytes - byte0;
    const u32 in_len = (u32)min(left, (u64)((1u << 28) - 1u));
    // (entry 0 takes no more whole blocks than the caller's text has room for: a call with little room still moves on)
    const u64 cap = min(min(stride, (u64)0xFFFF0000u), k == 0u ? (u64)WINDOW + text_cap : ~0ULL);
    u32 got = 0u;
    const u32 status = bns_infw::inflate_member_wave<u16, true>(&S, comp + byte0, in_len, comp_end, out, (u32)cap, &got, &io);
    if (lane == 0u) {
        ChunkOut r;
        r.n_out = got; r.status = status; r.end_bit = byte0 * 8ULL + io.end_bit; r.member_end = io.member_end; r.pad = 0u;
        res[k] = r;
    }
}

// One block: which chunks count and where their text goes.
__global__ __launch_bounds__(1024) void gz_valid_kernel(const u64 *__restrict__ entry, const u32 *__restrict__ n_entries, const ChunkOut *__restrict__ res, u64 text_cap,
                                                        u64 *text_off, CallOut *__restrict__ call)
{
    __shared__ u64 scan[1024];
    __shared__ u32 s_fail, s_why;
    const u32 t = threadIdx.x;
    const u32 n = *n_entries;
    if (t == 0u) { s_fail = n; s_why = 0u; }
    __syncthreads();
    u64 base = 0ULL;
    for (u32 k0 = 0u; k0 < n; k0 += 1024u) {
        const u32 k = k0 + t;
        u32 bad = 0u, nout = 0u;
        if (k < n) {
            const ChunkOut r = res[k];
            nout = r.n_out;
            if (r.status != bns_inf::INF_OK) bad = 4u;
            else if (k) {
                const ChunkOut f = res[k - 1u];
                if (f.member_end) bad = 3u;
                else if (entry[k] != f.end_bit) bad = 1u;
            }
        }
        scan[t] = nout;
        __syncthreads();
        for (u32 d = 1u; d < 1024u; d <<= 1) {
            const u64 v = t >= d ? scan[t - d] : 0ULL;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        const u64 cum = base + scan[t];
        if (k < n && !bad && cum > text_cap) bad = 2u;
        if (bad) atomicMin(&s_fail, k);
        __syncthreads();
        const u32 f = s_fail;
        if (bad && k == f) s_why = bad;
        if (k < f) text_off[k] = cum - nout;
        base += scan[1023];
        __syncthreads();
        if (f < n) break;
    }
    __threadfence_block();
    __syncthreads();
    if (t == 0u) {
        const u32 f = s_fail;
        if (f == 0u) { call->text_bytes = 0ULL; call->end_bit = n ? entry[0] : 0ULL; call->member_end = 0u; }
        else {                                                 // the last chunk that counts (its text_off: written above, by this block)
            const ChunkOut r = res[f - 1u];
            call->text_bytes = text_off[f - 1u] + r.n_out; call->end_bit = r.end_bit; call->member_end = r.member_end;
        }
        call->n_chunks = n; call->n_good = f; call->status0 = n ? res[0].status : (u32)bns_inf::INF_BAD_BLOCK;
        call->stop_why = s_why ? s_why : (f && res[f - 1u].member_end ? 3u : 0u); call->crc = 0u;
    }
}

// Groups of G chunks side by side, one block each: P starts as the identity (marker j = byte j of the window in front of the GROUP);
// step k stores P as P_k and replaces it by chunk k's last WINDOW symbols looked up in it.  What is left is the group's own function.
__global__ __launch_bounds__(1024) void gz_compose_kernel(const CallOut *__restrict__ call, const ChunkOut *__restrict__ res, const u16 *__restrict__ sym, u64 stride, u32 G,
                                                          u16 *__restrict__ pbuf, u16 *__restrict__ fbuf)
{
    __shared__ u16 P[WINDOW];
    const u32 t = threadIdx.x, g = blockIdx.x;
    const u32 good = call->n_good;
    const u32 k0 = g * G;
    if (k0 >= good) return;
    const u32 k1 = min(k0 + G, good);
    for (u32 j = t; j < WINDOW; j += 1024u) P[j] = (u16)(0x8000u | j);
    __syncthreads();
    for (u32 k = k0; k < k1; ++k) {
        u16 *pk = pbuf + (u64)k * WINDOW;
        for (u32 j = t * 8u; j < WINDOW; j += 8192u) *reinterpret_cast<uint4 *>(pk + j) = *reinterpret_cast<const uint4 *>(P + j);
        const u16 *tail = sym + (u64)k * stride + res[k].n_out;     // (= prefix + n_out - WINDOW: the last WINDOW symbols of prefix + text)
        u16 nw[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const u16 s = tail[(u32)q * 1024u + t];
            nw[q] = s < 0x8000u ? s : P[s & 0x7FFFu];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) P[(u32)q * 1024u + t] = nw[q];
        __syncthreads();
    }
    u16 *fg = fbuf + (u64)g * WINDOW;
    for (u32 j = t * 8u; j < WINDOW; j += 8192u) *reinterpret_cast<uint4 *>(fg + j) = *reinterpret_cast<const uint4 *>(P + j);
}

// One block: the window in front of every group (wg[g]), and behind the last chunk that counts (window_out).
__global__ __launch_bounds__(1024) void gz_groups_kernel(const CallOut *__restrict__ call, const u16 *__restrict__ fbuf, u32 G, const u8 *__restrict__ window0,
                                                         u8 *__restrict__ wg, u8 *__restrict__ window_out)
{
    __shared__ u8 W[WINDOW];
    const u32 t = threadIdx.x;
    const u32 good = call->n_good;
    const u32 ng = (good + G - 1u) / G;
    for (u32 j = t; j < WINDOW; j += 1024u) W[j] = window0 ? window0[j] : (u8)0;
    __syncthreads();
    for (u32 g = 0u; g < ng; ++g) {
        u8 *w = wg + (u64)g * WINDOW;
        for (u32 j = t * 16u; j < WINDOW; j += 16384u) *reinterpret_cast<uint4 *>(w + j) = *reinterpret_cast<const uint4 *>(W + j);
        const u16 *f = fbuf + (u64)g * WINDOW;
        u8 nw[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const u16 s = f[(u32)q * 1024u + t];
            nw[q] = s < 0x8000u ? (u8)s : W[s & 0x7FFFu];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) W[(u32)q * 1024u + t] = nw[q];
        __syncthreads();
    }
    for (u32 j = t; j < WINDOW; j += 1024u) window_out[j] = W[j];
}

// symbols -> bytes: block (x, k) takes symbols [x * TILE, ...) of chunk k
constexpr u32 TR_TILE = 8192u;
__global__ __launch_bounds__(256) void gz_translate_kernel(const CallOut *__restrict__ call, const ChunkOut *__restrict__ res, const u16 *__restrict__ sym, u64 stride, u32 G,
                                                           const u16 *__restrict__ pbuf, const u8 *__restrict__ wg, const u64 *__restrict__ text_off, u8 *__restrict__ text)
{
    const u32 k = blockIdx.y;
    if (k >= call->n_good) return;
    const u32 n = res[k].n_out;
    const u32 lo = blockIdx.x * TR_TILE;
    if (lo >= n) return;
    const u32 hi = min(n, lo + TR_TILE);
    const u16 *s = sym + (u64)k * stride + WINDOW;
    const u16 *pk = pbuf + (u64)k * WINDOW;
    const u8 *w = wg + (u64)(k / G) * WINDOW;
    u8 *o = text + text_off[k];
    auto byte_of = [&](u16 x) -> u8 {
        if (x < 0x8000u) return (u8)x;
        const u16 p = pk[x & 0x7FFFu];
        return p < 0x8000u ? (u8)p : w[p & 0x7FFFu];
    };
    for (u32 i = lo + threadIdx.x * 8u; i < hi; i += 256u * 8u) {
        if (i + 8u <= hi) {
            u16 v[8];
            __builtin_memcpy(v, s + i, 16);
            u8 b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = byte_of(v[q]);
            __builtin_memcpy(o + i, b, 8);
        } else {
            for (u32 q = i; q < hi; ++q) o[q] = byte_of(s[q]);
        }
    }
}

// CRC-32 of the call's text: every chained chunk's text in four parts, a wavefront each (64 slices), each part's CRC moved over the
// bytes behind it (zlib's x^n mod p arithmetic; XOR is the combination, so the parts fold theirs in whatever order they finish)
constexpr u32 CRC_PARTS = 4u;
__global__ __launch_bounds__(64) void gz_crc_kernel(CallOut *__restrict__ call, const ChunkOut *__restrict__ res, const u64 *__restrict__ text_off,
                                                    const u8 *__restrict__ text)
{
    __shared__ u32 tbl[256];
    const u32 k = blockIdx.x, part = blockIdx.y;
    if (k >= call->n_good) return;
    const u32 n = res[k].n_out;
    const u32 q = ((n + CRC_PARTS - 1u) / CRC_PARTS + 3u) & ~3u;
    const u32 lo = min(n, part * q), hi = part + 1u == CRC_PARTS ? n : min(n, (part + 1u) * q);
    if (hi <= lo) return;
    const u32 c = bns_infw::crc32_wave(tbl, text + text_off[k] + lo, hi - lo);
    if (threadIdx.x == 0) {
        const u64 behind = call->text_bytes - (text_off[k] + hi);
        atomicXor(&call->crc, bns_infw::multmodp(bns_infw::x2nmodp((u32)behind, 3u), c));
    }
}
}  // namespace gzs

extern "C" {

uint32_t bns_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    // zlib's crc32_combine (crc32.c): crc1 advanced over len2 zero bytes, then crc2 folded in
    uint32_t p = 1u << 31;                                     // x^0
    uint32_t sq = 1u << 23;                                    // x^8: one byte
    for (uint64_t n = len2; n; n >>= 1) {
        if (n & 1u) p = bns_infw::multmodp(sq, p);
        sq = bns_infw::multmodp(sq, sq);
    }
    return bns_infw::multmodp(p, crc1) ^ crc2;
}

namespace {
struct GzPlan { u32 CH, n_chunks, G, n_groups; u64 stride; size_t tab_bytes, sym_bytes, win_bytes; };
GzPlan gz_plan(u64 bytes_from_first, int n_cu, unsigned room)
{
    auto env_num = [](const char *name, u64 dflt) { const char *e = getenv(name); return e && atol(e) > 0 ? (u64)atol(e) : dflt; };
    GzPlan p;
    p.CH = (u32)std::min<u64>(std::max<u64>(env_num("BNS_GZ_CHUNK_KB", 64) << 10, 4096), 1u << 24);
    // No more chunks than TWELVE wavefronts per CU: the decode kernel is one block per entry, 11.2 KB of LDS each -- fourteen fit a CU and
    // fill its LDS for the 15-20 ms a call's entries take, and while they do the classify kernels of the caller's other stream (18.9 KB a
    // block) find room only between two calls: one bns_classify_text call in two took a second instead of 20 ms (measured).  Twelve leave
    // 25 KB.  (BNS_GZ_WAVES_PER_CU: measurements)
    {
        const u64 per_cu = std::min<u64>(std::max<u64>(env_num("BNS_GZ_WAVES_PER_CU", 12), 1), 64);
        const u64 most = per_cu * (u64)std::max(n_cu, 1);
        if ((bytes_from_first + p.CH - 1) / p.CH > most) p.CH = (u32)std::min<u64>((((bytes_from_first + most - 1) / most) + 4095) & ~4095ULL, 1u << 24);
    }
    const u64 ratio = std::min<u64>(std::max<u64>(room ? room : env_num("BNS_GZ_RATIO_CAP", 16), 2), 1024);
    p.n_chunks = (u32)((bytes_from_first + p.CH - 1) / p.CH);
    p.stride = ((u64)gzs::WINDOW + ratio * p.CH + 1024u + 7u) & ~7ULL;       // symbols per chunk (prefix included)
    p.G = std::max<u32>(4u, (u32)std::ceil(std::sqrt((double)p.n_chunks)));
    p.n_groups = (p.n_chunks + p.G - 1) / p.G;
    // tables: start[n], entry[n], stop[n], text_off[n] (u64), res[n] (24 B), n_entries, CallOut
    p.tab_bytes = (size_t)p.n_chunks * (4 * 8 + sizeof(gzs::ChunkOut)) + 512;
    p.sym_bytes = (size_t)p.n_chunks * (size_t)p.stride * 2;
    // P_k per chunk and the function of every group (u16[WINDOW]), the window in front of every group (u8[WINDOW])
    p.win_bytes = ((size_t)p.n_chunks + p.n_groups) * gzs::WINDOW * 2 + (size_t)p.n_groups * gzs::WINDOW;
    return p;
}
}  // namespace

int bns_inflate_stream_reserve(bns_inflater *h, uint64_t comp_bytes)
{
    if (!h || comp_bytes >= (1ULL << 31)) return BNS_ERR_ARG;
    INFCHK(h, hipSetDevice(h->device));
    const GzPlan p = gz_plan(comp_bytes, h->n_cu, h->stream_room);
    int rc;
    if ((rc = ensure(h, h->d_comp, (size_t)comp_bytes + 64)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_tab, p.tab_bytes)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_scratch, p.sym_bytes)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_res, p.win_bytes)) != BNS_OK) return rc;
    return BNS_OK;
}

int bns_inflate_stream_room(bns_inflater *h, uint32_t symbols_per_byte)
{
    if (!h || symbols_per_byte > 1024) return BNS_ERR_ARG;
    h->stream_room = symbols_per_byte;
    return BNS_OK;
}

int bns_inflate_stream_prefetch(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes)
{
    if (!h || !comp || !comp_bytes || comp_bytes >= (1ULL << 31)) return BNS_ERR_ARG;
    INFCHK(h, hipSetDevice(h->device));
    if (!h->copy_stream) {
        INFCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : h->pre_done) INFCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int i = h->pre_turn;
    h->pre_turn ^= 1;
    h->pre_host[i] = nullptr;
    int rc = ensure(h, h->d_pre[i], (size_t)comp_bytes + 4096);
    if (rc != BNS_OK) return rc;
    if (h->called) INFCHK(h, hipStreamWaitEvent(h->copy_stream, h->done, 0));       // (the kernels of the last call may still be reading this buffer's old bytes)
    INFCHK(h, hipMemcpyAsync(h->d_pre[i].p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, h->copy_stream));
    INFCHK(h, hipMemsetAsync((bns_inf::u8 *)h->d_pre[i].p + comp_bytes, 0, 64, h->copy_stream));
    INFCHK(h, hipEventRecord(h->pre_done[i], h->copy_stream));
    h->pre_host[i] = comp; h->pre_bytes[i] = (size_t)comp_bytes;
    return BNS_OK;
}

int bns_inflate_stream_device(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, uint64_t start_bit, const void *d_window, void *d_text,
                              uint64_t text_cap, void *d_window_out, bns_gz_result *out)
{
    using namespace gzs;
    if (!h || !comp || !d_text || !d_window_out || !out || start_bit >= comp_bytes * 8ULL || comp_bytes >= (1ULL << 31) || text_cap >= (1ULL << 32)) return BNS_ERR_ARG;
    *out = bns_gz_result{};
    INFCHK(h, hipSetDevice(h->device));
    // Bytes that lie inside a range bns_inflate_stream_prefetch brought up are read where they are: from the 4-byte boundary in front of
    // them (the search kernel reads words), every position of this call `shift` bytes further on
    int pre_i = -1;
    for (int i = 0; i < 2; ++i)
        if (h->pre_host[i] && comp >= h->pre_host[i] && comp + comp_bytes <= h->pre_host[i] + h->pre_bytes[i]) pre_i = i;
    u64 shift = 0;
    const u8 *pre_dev = nullptr;
    if (pre_i >= 0) {
        const u64 off = (u64)(comp - h->pre_host[pre_i]);
        shift = off & 3u;
        pre_dev = (const u8 *)h->d_pre[pre_i].p + (off - shift);
        comp_bytes += shift; start_bit += 8u * shift;
    }
    const u64 first_byte = start_bit >> 3;
    const GzPlan pl = gz_plan(comp_bytes - first_byte, h->n_cu, h->stream_room);
    const u32 CH = pl.CH, n_chunks = pl.n_chunks, G = pl.G, n_groups = pl.n_groups;
    const u64 stride = pl.stride;
    int rc;
    if (!pre_dev && (rc = ensure(h, h->d_comp, (size_t)comp_bytes + 64)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_tab, pl.tab_bytes)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_scratch, pl.sym_bytes)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_res, pl.win_bytes)) != BNS_OK) return rc;
    hipStream_t st = h->stream;
    u64 *d_start = (u64 *)h->d_tab.p, *d_entry = d_start + n_chunks, *d_stop = d_entry + n_chunks, *d_off = d_stop + n_chunks;
    ChunkOut *d_res = (ChunkOut *)(d_off + n_chunks);
    u32 *d_n = (u32 *)(d_res + n_chunks);
    CallOut *d_call = (CallOut *)(((uintptr_t)(d_n + 1) + 15) & ~(uintptr_t)15);
    u16 *d_sym = (u16 *)h->d_scratch.p;
    u16 *d_pbuf = (u16 *)h->d_res.p, *d_fbuf = d_pbuf + (size_t)n_chunks * WINDOW;
    u8 *d_wg = (u8 *)(d_fbuf + (size_t)n_groups * WINDOW);
    const u8 *d_comp = pre_dev ? pre_dev : (const u8 *)h->d_comp.p;
    if (pre_dev) INFCHK(h, hipStreamWaitEvent(st, h->pre_done[pre_i], 0));
    else {
        INFCHK(h, hipMemcpyAsync(h->d_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
        INFCHK(h, hipMemsetAsync((u8 *)h->d_comp.p + comp_bytes, 0, 64, st));
    }
    INFCHK(h, hipMemsetAsync(d_start, 0xFF, (size_t)n_chunks * 8, st));
    INFCHK(h, hipMemcpyAsync(d_start, &start_bit, 8, hipMemcpyHostToDevice, st));          // (chunk 0's header is the caller's)
    INFCHK(h, hipEventRecord(h->ev0, st));
    if (n_chunks > 1u)
        hipLaunchKernelGGL(gz_search_kernel, dim3(n_chunks - 1u), dim3(64), 0, st, d_comp, (u64)comp_bytes, CH, n_chunks, (u64)start_bit, d_start);
    hipLaunchKernelGGL(gz_plan_kernel, dim3(1), dim3(64), 0, st, (const u64 *)d_start, n_chunks, d_entry, d_stop, d_n);
    const u8 *ce = d_comp + (((size_t)comp_bytes + 64) & ~(size_t)3);
    hipLaunchKernelGGL(gz_decode_kernel, dim3(n_chunks), dim3(64), 0, st, d_comp, ce, (u64)comp_bytes, (const u64 *)d_entry, (const u64 *)d_stop, (const u32 *)d_n,
                       (const u8 *)d_window, d_sym, stride, (u64)text_cap, d_res);
    hipLaunchKernelGGL(gz_valid_kernel, dim3(1), dim3(1024), 0, st, (const u64 *)d_entry, (const u32 *)d_n, (const ChunkOut *)d_res, (u64)text_cap, d_off, d_call);
    hipLaunchKernelGGL(gz_compose_kernel, dim3(n_groups), dim3(1024), 0, st, (const CallOut *)d_call, (const ChunkOut *)d_res, (const u16 *)d_sym, stride, G, d_pbuf, d_fbuf);
    hipLaunchKernelGGL(gz_groups_kernel, dim3(1), dim3(1024), 0, st, (const CallOut *)d_call, (const u16 *)d_fbuf, G, (const u8 *)d_window, d_wg, (u8 *)d_window_out);
    const u32 tiles = (u32)((stride - WINDOW + TR_TILE - 1) / TR_TILE);
    hipLaunchKernelGGL(gz_translate_kernel, dim3(tiles, n_chunks), dim3(256), 0, st, (const CallOut *)d_call, (const ChunkOut *)d_res, (const u16 *)d_sym, stride, G,
                       (const u16 *)d_pbuf, (const u8 *)d_wg, (const u64 *)d_off, (u8 *)d_text);
    hipLaunchKernelGGL(gz_crc_kernel, dim3(n_chunks, CRC_PARTS), dim3(64), 0, st, d_call, (const ChunkOut *)d_res, (const u64 *)d_off, (const u8 *)d_text);
    INFCHK(h, hipGetLastError());
    INFCHK(h, hipEventRecord(h->ev1, st));
    CallOut co{};
    INFCHK(h, hipMemcpyAsync(&co, d_call, sizeof(co), hipMemcpyDeviceToHost, st));
    INFCHK(h, hipEventRecord(h->done, st));
    INFCHK(h, hipEventSynchronize(h->done));
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_kernel_ms = ms;
    if (getenv("BNS_GZ_DEBUG")) {                               // (what every chunk did: the last entries)
        std::vector<u64> he(co.n_chunks), hs(co.n_chunks);
        std::vector<ChunkOut> hr(co.n_chunks);
        (void)hipMemcpy(he.data(), d_entry, co.n_chunks * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs.data(), d_stop, co.n_chunks * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hr.data(), d_res, co.n_chunks * sizeof(ChunkOut), hipMemcpyDeviceToHost);
        for (u32 k = co.n_chunks > 6 ? co.n_chunks - 6 : 0; k < co.n_chunks; ++k)
            fprintf(stderr, "[gz] entry %u: header at bit %llu (byte %llu), stop %lld, status %u, %u symbols, ends at bit %llu (byte %llu bit %u), member_end %u\n", k, (unsigned long long)he[k],
                    (unsigned long long)(he[k] >> 3), (long long)hs[k], hr[k].status, hr[k].n_out, (unsigned long long)hr[k].end_bit, (unsigned long long)(hr[k].end_bit >> 3), (unsigned)(hr[k].end_bit & 7), hr[k].member_end);
        fprintf(stderr, "[gz] %u entries, %u taken, why %u, text %llu, end bit %llu\n", co.n_chunks, co.n_good, co.stop_why, (unsigned long long)co.text_bytes, (unsigned long long)co.end_bit);
    }
    h->called = true;
    out->text_bytes = co.text_bytes; out->end_bit = co.end_bit - 8u * shift; out->member_end = co.member_end; out->crc32 = co.crc;
    out->n_chunks = co.n_chunks; out->n_chained = co.n_good; out->status = co.n_good ? (u32)BNS_INF_OK : (co.status0 ? co.status0 : (u32)BNS_INF_OUT_OVERFLOW);
    out->stop_why = co.stop_why;
    return BNS_OK;
}

}  // extern "C"
