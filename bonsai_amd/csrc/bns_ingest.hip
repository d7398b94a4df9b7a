// bns_ingest.hip -- FASTA / FASTQ text parsed on the device (gfx950, wave64): the host-ingest row of SURVEY 8f-2 without a host
// parser.  Included at the end of bns_api.hip (one translation unit: bns_ctx, HIPCHK, ensure, classify_device_impl).
//
// Replaces kseq_read (klib/kseq.h:177-225) + bseq_read's loop (kseq_declare.h:112-145) for text in the REGULAR FORM -- the
// form in which kseq_read's character-level state machine is a function of whole lines:
//
//     stream  := blank* record*
//     record  := H (S | blank)* [ P Q blank* ]
//     H  a line whose first byte is '>' or '@'          S  a non-empty line whose first byte is none of '>' '@' '+'
//     P  a line whose first byte is '+'                 Q  THE line behind a P, whatever it starts with, as long as the S lines together
//
// and no line ends in '\r'.  Why kseq_read yields exactly these records on such text (klib/kseq.h line numbers):
//   * :183-186 skips to the next '>' / '@' CHARACTER; with only blank lines in front of H that is H's first byte.
//   * :190-191 name = H up to the first isspace byte; the rest of the line is the comment.
//   * :197-201 reads lines until one STARTS with '>', '+' or '@' (first byte tested, '\n' skipped: blank lines), appending
//     every other line whole: the S lines.  '>' / '@' ends a FASTA record (:202) -- the next H; '+' (:210) is P.
//   * :215 skips the rest of P; :217 reads quality lines until they are as long as the sequence: ONE line when Q is exactly that
//     long (shorter: kseq reads on -- not regular; longer: error -2 at :220 -- not regular).
// The only line whose role is not decided by its own first byte is Q (it may start with '@' or '+').  Q is "the line behind a
// true P", and a line that starts with '+' is a true P unless it is itself a Q: in a run of consecutive lines that all start with
// '+', the first is P (the line in front of it does not start with '+', so cannot be a P that makes it a Q), the second its Q,
// and so on alternately -- a bounded look-back per line (line_role_kernel), no state carried along the text.  Everything the
// grammar does not allow (S or P where only H or a blank may stand, a Q of the wrong length) is checked per record
// (record_kernel) and reported as BNS_TEXT_IRREGULAR: nothing is guessed, the caller's host parser takes that stretch.
//
// Kernels (all HBM-bound byte / integer work: ~2 passes over the text, the rest over 4 bytes per line):
//   text_count_kernel    '\n' per 16 KiB tile (64 bytes per lane as 4 x dwordx4, SWAR byte compare)
//   scan_*               exclusive prefix sums (tile counts -> line numbers; header flags -> record numbers; lengths -> offsets)
//   line_write_kernel    line_start[] (u32 per line)
//   line_role_kernel     role per line (the '+'-run rule)
//   decide_kernel        how many records this call takes (complete ones in front of `limit`; the minimum over a pair of files)
//   record_kernel        one lane per record: lines walked, grammar checked, sequence length, name length
//   pack_text_kernel     one wavefront per record: bases gathered from the record's lines -> 2-bit words + invalid-base flags
//   names_kernel         names gathered into one blob
// then classify_device_impl on the packed words, and hit_runs_kernel when the caller prints runs.
#include <new>

namespace bns {
namespace ingest {

constexpr u32 TILE = 16384;                     // text bytes per 256-thread block
constexpr u32 SCAN_ITEMS = 16;                  // elements per thread of a scan block (4096 per block)
constexpr u32 MAX_REC_LINES = 4096;
constexpr u32 MAX_PLUS_RUN = 16;
enum : u8 { ROLE_E = 0, ROLE_S = 1, ROLE_H = 2, ROLE_P = 3, ROLE_Q = 4 };

struct StreamInfo {
    u32 n_nl, n_lines, n_hdr, n_take;
    u32 consumed, why, lo, hi;
    u32 n_eff, pad;                             // headers that yield a record (a final text that ends in a bare '>' / '@' byte: that one does not)
};
struct CallInfo {
    StreamInfo s[2];
    u32 n_take;                                 // records per stream this call takes
    u32 n_reads;                                // n_take * n_streams
    u32 max_len, why;
    u32 total_bases, names_bytes;
    u32 pad[2];
};

// ---- 256-thread block helpers -------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_incl_scan(u32 v)
{
    const u32 lane = threadIdx.x & 63u;
#pragma unroll
    for (u32 off = 1; off < 64; off <<= 1) {
        const u32 t = (u32)__shfl_up((int)v, (int)off);
        if (lane >= off) v += t;
    }
    return v;
}
// exclusive prefix of v over the block's 256 threads; total = the block's sum (same in every thread)
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 &total, u32 *lds4)
{
    const u32 incl = wave_incl_scan(v);
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    __syncthreads();                                           // (lds4 may still be read from a previous use)
    if (lane == 63) lds4[w] = incl;
    __syncthreads();
    u32 base = 0;
    total = 0;
#pragma unroll
    for (u32 i = 0; i < 4; ++i) { const u32 t = lds4[i]; if (i < w) base += t; total += t; }
    return base + incl - v;
}

// 4-bit mask of the bytes of w that equal '\n' (exact zero-byte test of w ^ 0x0A0A0A0A, bits gathered by one multiply)
__device__ __forceinline__ u32 nl_nibble(u32 w)
{
    const u32 x = w ^ 0x0A0A0A0Au;
    const u32 t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    const u32 z = ~(t | 0x7F7F7F7Fu);                          // 0x80 in every byte that was '\n'
    return (((z >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// '\n' mask of the 64 bytes at text + base (base a multiple of 64); only offsets in [lo, hi) count
__device__ __forceinline__ u64 nl_mask64(const u8 *__restrict__ text, u32 base, u32 lo, u32 hi)
{
    if (base >= hi || base + 64u <= lo) return 0;
    const uint4 *p = reinterpret_cast<const uint4 *>(text + base);
    u64 m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 v = p[j];
        const u32 n = nl_nibble(v.x) | (nl_nibble(v.y) << 4) | (nl_nibble(v.z) << 8) | (nl_nibble(v.w) << 12);
        m |= (u64)n << (16 * j);
    }
    if (base < lo) m &= ~0ULL << (lo - base);
    if (hi - base < 64u) m &= (1ULL << (hi - base)) - 1ULL;
    return m;
}

__global__ __launch_bounds__(256) void text_count_kernel(const u8 *__restrict__ text, u32 lo, u32 hi, u32 tile0, u32 *__restrict__ tile_cnt)
{
    __shared__ u32 lds4[4];
    const u32 base = (tile0 + blockIdx.x) * TILE + threadIdx.x * 64u;
    const u32 cnt = (u32)__popcll(nl_mask64(text, base, lo, hi));
    u32 total;
    (void)block_excl_scan(cnt, total, lds4);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// ---- exclusive scan over n elements: in(i) -> out(i, prefix, value); n comes from device memory ------------------------------
// three launches: per-block sums (4096 elements each), one block over the sums, per-block rescan with the block's base.
template <class In>
__global__ __launch_bounds__(256) void scan_sums_kernel(In in, const u32 *__restrict__ n_ptr, u32 n_mul, u32 *__restrict__ sums)
{
    __shared__ u32 lds4[4];
    const u32 n = *n_ptr * n_mul;
    const u32 i0 = (blockIdx.x * 256u + threadIdx.x) * SCAN_ITEMS;
    u32 v = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) if (i0 + j < n) v += in(i0 + j);
    u32 total;
    (void)block_excl_scan(v, total, lds4);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// sums[0..m) -> exclusive prefixes in place, the grand total at sums[m] (and at *total_out)
__global__ __launch_bounds__(256) void scan_top_kernel(u32 *__restrict__ sums, u32 m, u32 *__restrict__ total_out)
{
    __shared__ u32 lds4[4];
    u32 carry = 0;
    for (u32 b = 0; b < m; b += 256u * SCAN_ITEMS) {
        const u32 i0 = b + threadIdx.x * SCAN_ITEMS;
        u32 loc[SCAN_ITEMS];
        u32 v = 0;
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; ++j) { loc[j] = i0 + j < m ? sums[i0 + j] : 0u; v += loc[j]; }
        u32 total;
        u32 pre = carry + block_excl_scan(v, total, lds4);
#pragma unroll
        for (u32 j = 0; j < SCAN_ITEMS; ++j) { if (i0 + j < m) sums[i0 + j] = pre; pre += loc[j]; }
        carry += total;
    }
    if (threadIdx.x == 0) { sums[m] = carry; if (total_out) *total_out = carry; }
}
template <class In, class Out>
__global__ __launch_bounds__(256) void scan_apply_kernel(In in, const u32 *__restrict__ n_ptr, u32 n_mul, const u32 *__restrict__ sums, Out out)
{
    __shared__ u32 lds4[4];
    const u32 n = *n_ptr * n_mul;
    const u32 i0 = (blockIdx.x * 256u + threadIdx.x) * SCAN_ITEMS;
    u32 loc[SCAN_ITEMS];
    u32 v = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) { loc[j] = i0 + j < n ? in(i0 + j) : 0u; v += loc[j]; }
    u32 total;
    u32 pre = sums[blockIdx.x] + block_excl_scan(v, total, lds4);
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) { if (i0 + j < n) out(i0 + j, pre, loc[j]); pre += loc[j]; }
    if (i0 <= n && n < i0 + SCAN_ITEMS) out.total(n, pre - 0u);      // (the thread whose range holds index n: pre has run over every element below n)
}

// ---- lines ---------------------------------------------------------------------------------------------------------------------
// line_start[j + 1] = offset behind the j-th '\n' of [lo, hi); line_start[0] = lo; line_start[n_nl + 1] = hi + 1 (so that
// "length of line i" = line_start[i + 1] - 1 - line_start[i] also holds for a last line without a newline)
__global__ __launch_bounds__(256) void line_write_kernel(const u8 *__restrict__ text, u32 lo, u32 hi, u32 tile0, u32 n_tiles,
                                                         const u32 *__restrict__ tile_base, u32 *__restrict__ ls, u32 cap_lines,
                                                         StreamInfo *__restrict__ si)
{
    __shared__ u32 lds4[4];
    const u32 base = (tile0 + blockIdx.x) * TILE + threadIdx.x * 64u;
    u64 m = nl_mask64(text, base, lo, hi);
    u32 total;
    u32 idx = tile_base[blockIdx.x] + block_excl_scan((u32)__popcll(m), total, lds4) + 1u;
    while (m) {
        const u32 b = (u32)__builtin_ctzll(m);
        m &= m - 1;
        if (idx < cap_lines) ls[idx] = base + b + 1u;
        ++idx;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u32 n_nl = tile_base[n_tiles];
        ls[0] = lo;
        si->n_nl = n_nl; si->lo = lo; si->hi = hi;
        if (n_nl + 2u <= cap_lines) { ls[n_nl + 1u] = hi + 1u; si->n_lines = n_nl + 1u; }
        else { si->n_lines = 0; si->why |= BNS_TEXT_WHY_LINES; }
    }
}

__device__ __forceinline__ bool starts_plus(const u8 *__restrict__ text, const u32 *__restrict__ ls, u32 i)
{
    const u32 s = ls[i];
    return ls[i + 1] - 1u > s && text[s] == '+';
}

__global__ __launch_bounds__(256) void line_role_kernel(const u8 *__restrict__ text, const u32 *__restrict__ ls, StreamInfo *__restrict__ si,
                                                        u8 *__restrict__ role)
{
    const u32 n = si->n_lines;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const u32 s = ls[i], len = ls[i + 1] - 1u - s;
        const u8 c0 = len ? text[s] : (u8)0;
        u32 run = 0;
        for (u32 j = i; j > 0 && run <= MAX_PLUS_RUN; ) { --j; if (starts_plus(text, ls, j)) ++run; else break; }
        if (run > MAX_PLUS_RUN) atomicOr(&si->why, BNS_TEXT_WHY_PLUS_RUN);
        u8 r;
        if (run & 1u) r = ROLE_Q;
        else if (c0 == '>' || c0 == '@') r = ROLE_H;
        else if (c0 == '+') r = ROLE_P;
        else r = len ? ROLE_S : ROLE_E;
        role[i] = r;
    }
}

struct InIsHeader { const u8 *role; __device__ u32 operator()(u32 i) const { return role[i] == ROLE_H ? 1u : 0u; } };
struct OutHeaderLines {
    u32 *hline; u32 cap; StreamInfo *si;
    __device__ void operator()(u32 i, u32 pre, u32 v) const { if (v && pre < cap) hline[pre] = i; }
    __device__ void total(u32, u32 t) const { si->n_hdr = t; if (t > cap) atomicOr(&si->why, BNS_TEXT_WHY_LINES); }
};
struct InU32 { const u32 *a; __device__ u32 operator()(u32 i) const { return a[i]; } };
struct OutOffsets64 { u64 *off; u64 base; u32 *total_out; __device__ void operator()(u32 i, u32 pre, u32) const { off[i] = base + pre; } __device__ void total(u32 n, u32 t) const { off[n] = base + t; *total_out = t; } };
struct OutOffsets32 { u32 *off; u32 base; u32 *total_out; __device__ void operator()(u32 i, u32 pre, u32) const { off[i] = base + pre; } __device__ void total(u32 n, u32 t) const { off[n] = base + t; *total_out = t; } };

// How many records this call takes.  Per stream: the headers in front of `limit` (stream 0), of which the last one is only
// complete when the text is final or another header follows; a pair of files takes the minimum.  consumed = where the first
// record not taken starts.  One thread.
__global__ void decide_kernel(CallInfo *__restrict__ ci, u32 n_streams, u32 limit, int final_text,
                              const u32 *__restrict__ ls0, const u32 *__restrict__ hl0, const u8 *__restrict__ role0,
                              const u32 *__restrict__ ls1, const u32 *__restrict__ hl1, const u8 *__restrict__ role1)
{
    u32 take = 0xFFFFFFFFu;
    for (u32 s = 0; s < n_streams; ++s) {
        StreamInfo &si = ci->s[s];
        const u32 *ls = s ? ls1 : ls0, *hl = s ? hl1 : hl0;
        const u8 *role = s ? role1 : role0;
        if (si.why) { ci->why |= si.why; si.n_take = 0; take = 0; continue; }
        // text in front of the first header: blank lines only (kseq skips to the next '>' / '@' byte wherever it stands)
        const u32 lead = si.n_hdr ? hl[0] : si.n_lines;
        for (u32 i = 0; i < lead; ++i) {
            if (role[i] != ROLE_E) {
                // (the unfinished last line of a text that is not final may be anything: it is not looked at yet)
                if (!(i + 1 == si.n_lines && !final_text)) { ci->why |= BNS_TEXT_WHY_LEADING; }
                break;
            }
            if (i >= MAX_REC_LINES) { ci->why |= BNS_TEXT_WHY_LONG_RECORD; break; }
        }
        // klib/kseq.h:189: a header byte with NOTHING behind it (the last byte of the input) ends the stream without a record
        u32 n_eff = si.n_hdr;
        if (final_text && n_eff && ls[hl[n_eff - 1]] + 1u == si.hi) --n_eff;
        si.n_eff = n_eff;
        u32 t = n_eff;
        if (s == 0 && limit < si.hi) {                          // headers that start in front of the limit
            u32 a = 0, b = n_eff;
            while (a < b) { const u32 m = (a + b) >> 1; if (ls[hl[m]] < limit) a = m + 1; else b = m; }
            t = a;
        }
        if (t == n_eff && !final_text && t) --t;                // the last header's record ends where the next text begins
        si.n_take = t;
        take = take < t ? take : t;
    }
    if (ci->why) take = 0;
    ci->n_take = take;
    ci->n_reads = take * n_streams;
    for (u32 s = 0; s < n_streams; ++s) {
        StreamInfo &si = ci->s[s];
        const u32 *ls = s ? ls1 : ls0, *hl = s ? hl1 : hl0;
        // the first record not taken; with every header taken (a final text) the end of the text; nothing there yet: the start
        if (ci->why) si.consumed = si.lo;
        else if (take < si.n_eff) si.consumed = ls[hl[take]];
        else si.consumed = final_text ? si.hi : si.lo;
    }
}

// per-record arrays (mates interleaved: record r of stream s at R = r * n_streams + s)
struct RecArrays {
    u32 *seq_len, *name_len, *pos, *line0, *line1, *single;    // single: text offset of the one S line, ~0 when none or several
};

__device__ __forceinline__ bool is_space(u8 c) { return c == ' ' || (c >= 9 && c <= 13); }

__global__ __launch_bounds__(256) void record_kernel(const u8 *__restrict__ text, const u32 *__restrict__ ls, const u8 *__restrict__ role,
                                                     const u32 *__restrict__ hline, u32 *__restrict__ line_off, CallInfo *__restrict__ ci, u32 s,
                                                     u32 n_streams, int trim_readno, RecArrays ra)
{
    const u32 n = ci->n_take;
    const StreamInfo &si = ci->s[s];
    u32 my_max = 0, my_why = 0;
    for (u32 r = blockIdx.x * 256u + threadIdx.x; r < n; r += gridDim.x * 256u) {
        const u32 h = hline[r];
        const u32 end = r + 1 < si.n_hdr ? hline[r + 1] : si.n_lines;
        const u32 R = r * n_streams + s;
        // name: the header line behind its first byte, up to the first isspace byte (klib/kseq.h:190)
        const u32 hs = ls[h], he = ls[h + 1] - 1u;
        u32 nl = 0;
        while (hs + 1u + nl < he && !is_space(text[hs + 1u + nl])) ++nl;
        if (trim_readno && nl > 2 && text[hs + nl - 1u] == '/' && text[hs + nl] >= '0' && text[hs + nl] <= '9') nl -= 2;   // kseq_declare.h:106-110
        if (he > hs && text[he - 1u] == '\r') my_why |= BNS_TEXT_WHY_CR;
        u32 total = 0, n_s = 0, first = 0xFFFFFFFFu;
        u32 state = 0;                                          // 0: sequence lines, 1: behind the quality line
        if (end - h > MAX_REC_LINES) my_why |= BNS_TEXT_WHY_LONG_RECORD;
        else
        for (u32 l = h + 1; l < end; ++l) {
            const u8 rl = role[l];
            const u32 st = ls[l], len = ls[l + 1] - 1u - st;
            line_off[l] = total;
            if (len && text[st + len - 1u] == '\r') my_why |= BNS_TEXT_WHY_CR;
            if (rl == ROLE_E) continue;
            if (state) { my_why |= BNS_TEXT_WHY_AFTER_QUAL; break; }
            if (rl == ROLE_S) { if (!n_s) first = st; ++n_s; total += len; continue; }
            // ROLE_P: the next line is the quality line (klib/kseq.h:215-220); at the very end of a final text there may be none
            // (a '+' line that the input ends in, without its newline: kseq's error -2 at :216)
            u32 ql = 0xFFFFFFFFu;
            if (l + 1 < end) { ql = ls[l + 2] - 1u - ls[l + 1]; line_off[l + 1] = total; if (ql && text[ls[l + 2] - 2u] == '\r') my_why |= BNS_TEXT_WHY_CR; }
            if (ql != total) my_why |= BNS_TEXT_WHY_QUAL_LEN;
            ++l;
            state = 1;
        }
        ra.seq_len[R] = total; ra.name_len[R] = nl; ra.pos[R] = hs; ra.line0[R] = h + 1; ra.line1[R] = end;
        ra.single[R] = n_s == 1 ? first : 0xFFFFFFFFu;
        my_max = my_max > total ? my_max : total;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const u32 o = (u32)__shfl_xor((int)my_max, off), w = (u32)__shfl_xor((int)my_why, off);
        my_max = my_max > o ? my_max : o; my_why |= w;
    }
    if ((threadIdx.x & 63u) == 0) { if (my_max) atomicMax(&ci->max_len, my_max); if (my_why) atomicOr(&ci->why, my_why); }
}

// ---- pack: one wavefront per record, 256 bases a pass (4 per lane), the word layout of pack_kernel ----------------------------
struct PackSrc { const u8 *text; const u32 *ls; const u32 *line_off; };

// (offsets and the record arrays start at the slice's first record, which is record R0 of the batch's packed image: word base (offset >> 5) + index)
__global__ __launch_bounds__(256) void pack_text_kernel(PackSrc s0, PackSrc s1, u32 n_streams, RecArrays ra, const u64 *__restrict__ offsets, u32 R0,
                                                        const CallInfo *__restrict__ ci, u64 *__restrict__ words, u32 *__restrict__ nmask)
{
    const u32 lane = threadIdx.x & 63u;
    const u32 n = ci->n_reads;
    if (ci->why) return;
    const u32 n_waves = gridDim.x * 4u;
    for (u32 R = blockIdx.x * 4u + (threadIdx.x >> 6); R < n; R += n_waves) {
        const PackSrc &src = (n_streams == 2 && (R & 1u)) ? s1 : s0;
        const u32 L = ra.seq_len[R];
        const u64 wb = (offsets[R] >> 5) + R0 + R;
        const u32 n_words = (L + 31u) >> 5;
        const u32 single = ra.single[R];
        const u32 l0 = ra.line0[R], l1 = ra.line1[R];
        for (u32 p = 0; p < (n_words << 5); p += 256u) {
            const u32 bi = p + lane * 4u;
            u32 w = 0;                                          // up to four bytes of sequence, first base in the low byte
            if (bi < L) {
                const u32 nb = L - bi < 4u ? L - bi : 4u;
                if (single != 0xFFFFFFFFu) {
                    const u32 addr = single + bi, mis = addr & 3u;
                    const u32 *ap = reinterpret_cast<const u32 *>(src.text + (addr - mis));
                    const u32 lo = ap[0], hi = (mis + nb > 4u) ? ap[1] : 0u;
                    w = (u32)((((u64)hi << 32) | lo) >> (8u * mis));
                } else {
                    // the line that holds base bi: the last line of the record whose sequence offset is <= bi (blank lines share the
                    // offset of the line behind them and come first)
                    u32 a = l0, b = l1;
                    while (b - a > 1u) { const u32 m = (a + b) >> 1; if (src.line_off[m] <= bi) a = m; else b = m; }
                    u32 l = a, at = bi - src.line_off[l], st = src.ls[l], len = src.ls[l + 1] - 1u - st;
                    for (u32 i = 0; i < nb; ++i) {
                        while (at >= len) { ++l; at = 0; st = src.ls[l]; len = src.ls[l + 1] - 1u - st; }   // (bi + i < L: a line with bases follows)
                        w |= (u32)src.text[st + at] << (8u * i);
                        ++at;
                    }
                }
            }
            u32 codes = 0, bads = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32 bad;
                const u32 cd = base_code((w >> (8 * i)) & 0xFFu, bad);
                if (bi + (u32)i >= L) bad = 1u;
                codes = (codes << 2) | (bad ? 0u : cd);
                bads = (bads << 1) | bad;
            }
            const u32 g = lane & 7u;
            u32 hi32 = g < 4 ? codes << (24 - 8 * g) : 0u;
            u32 lo32 = g >= 4 ? codes << (24 - 8 * (g - 4)) : 0u;
            u32 nm = bads << (28 - 4 * g);
            hi32 |= dpp<QP_XOR1>(hi32); lo32 |= dpp<QP_XOR1>(lo32); nm |= dpp<QP_XOR1>(nm);
            hi32 |= dpp<QP_XOR2>(hi32); lo32 |= dpp<QP_XOR2>(lo32); nm |= dpp<QP_XOR2>(nm);
            hi32 |= (u32)__shfl_xor((int)hi32, 4); lo32 |= (u32)__shfl_xor((int)lo32, 4); nm |= (u32)__shfl_xor((int)nm, 4);
            const u32 wi = (p >> 5) + (lane >> 3);
            if (g == 0 && wi < n_words) { words[wb + wi] = ((u64)hi32 << 32) | lo32; nmask[wb + wi] = nm; }
        }
    }
}

__global__ __launch_bounds__(256) void names_kernel(const u8 *__restrict__ t0, const u8 *__restrict__ t1, u32 n_streams, RecArrays ra,
                                                    const u32 *__restrict__ name_off, u32 name_base, const CallInfo *__restrict__ ci,
                                                    char *__restrict__ names, u64 *__restrict__ pos64, u32 rel0, u32 rel1)
{
    const u32 n = ci->n_reads;
    if (ci->why) return;
    for (u32 R = blockIdx.x * 256u + threadIdx.x; R < n; R += gridDim.x * 256u) {
        const u8 *src = ((n_streams == 2 && (R & 1u)) ? t1 : t0) + ra.pos[R] + 1u;
        char *dst = names + (name_off[R] - name_base);
        const u32 len = ra.name_len[R];
        for (u32 i = 0; i < len; ++i) dst[i] = (char)src[i];
        pos64[R] = ra.pos[R] - ((n_streams == 2 && (R & 1u)) ? rel1 : rel0);     // (the caller's offsets, not the buffer's)
    }
}

}  // namespace ingest
}  // namespace bns

// ---------------------------------------------------------------------------------------------------------------------------------
// host side of the entry point
namespace {
using namespace bns::ingest;

// Text on its way to (or in) one of a stream's two device buffers: host[0, bytes) goes up in pieces on the copy stream, an event
// behind each.  bns_text_prefetch starts one for the NEXT call while the current one computes; a call whose text lies inside a
// pending upload uses it (whatever offset it starts at), any other call starts its own.
struct Upload {
    DevBuf buf;
    const char *host = nullptr;
    u64 bytes = 0, piece = 0;
    u32 n_pieces = 0;
    bool pending = false;                               // uploaded (or on its way) and not yet classified
    hipEvent_t ev[64] = {};
};
constexpr u32 MAX_PIECES = 64;

struct TextWork {                                       // the context's workspace for bns_classify_text (grow-only)
    Upload up[2][2];                                    // [stream][buffer]
    DevBuf ls[2], role[2], hline[2], line_off[2], tile[2], sums, info, offsets, words, nmask, hits;
    DevBuf rec_slice[5];                                // name_len, pos, line0, line1, single of the slice being parsed (its own pack / names kernels use them up)
    // what goes back to the host, TWICE: batch b's results are copied (on the back stream) while batch b + 1 is parsed and classified into the other set
    DevBuf seq_len[2], name_off[2], names[2], pos64[2], out[2][4], runs[2][4];
    hipEvent_t ev_done[2] = {}, tc0[2] = {}, tc1[2] = {};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    CallInfo *h_info = nullptr;                         // page-locked
    unsigned long long *h_cursor = nullptr;
};

template <class In, class Out>
int device_scan(bns_ctx *ctx, TextWork &tw, hipStream_t st, In in, const u32 *n_ptr, u32 n_mul, u32 n_cap, Out out)
{
    const u32 per_block = 256u * SCAN_ITEMS;
    const u32 blocks = (n_cap + per_block) / per_block + 0u;     // (index n itself -- the total -- must fall into a block)
    int rc = ensure(ctx, tw.sums, (size_t)(blocks + 2) * 4);
    if (rc != BNS_OK) return rc;
    u32 *sums = (u32 *)tw.sums.p;
    hipLaunchKernelGGL((scan_sums_kernel<In>), dim3(blocks), dim3(256), 0, st, in, n_ptr, n_mul, sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, st, sums, blocks, (u32 *)nullptr);
    hipLaunchKernelGGL((scan_apply_kernel<In, Out>), dim3(blocks), dim3(256), 0, st, in, n_ptr, n_mul, (const u32 *)sums, out);
    HIPCHK(ctx, hipGetLastError());
    return BNS_OK;
}
}  // namespace

struct bns_text_work : TextWork {};

void text_work_free(bns_ctx *ctx)
{
    bns_text_work *tw = ctx->text_work;
    if (!tw) return;
    std::vector<DevBuf *> bufs = {&tw->up[0][0].buf, &tw->up[0][1].buf, &tw->up[1][0].buf, &tw->up[1][1].buf, &tw->ls[0], &tw->ls[1], &tw->role[0], &tw->role[1],
                                  &tw->hline[0], &tw->hline[1], &tw->line_off[0], &tw->line_off[1], &tw->tile[0], &tw->tile[1], &tw->sums, &tw->info, &tw->offsets,
                                  &tw->words, &tw->nmask, &tw->hits};
    for (DevBuf &b : tw->rec_slice) bufs.push_back(&b);
    for (int q = 0; q < 2; ++q) {
        bufs.push_back(&tw->seq_len[q]);
        for (DevBuf &b : tw->out[q]) bufs.push_back(&b);
        for (DevBuf &b : tw->runs[q]) bufs.push_back(&b);
        bufs.push_back(&tw->name_off[q]); bufs.push_back(&tw->names[q]); bufs.push_back(&tw->pos64[q]);
    }
    for (DevBuf *b : bufs) release(*b);
    for (auto &row : tw->up) for (Upload &u : row) for (hipEvent_t e : u.ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {tw->t0, tw->t1, tw->ev_done[0], tw->ev_done[1], tw->tc0[0], tw->tc0[1], tw->tc1[0], tw->tc1[1]}) if (e) (void)hipEventDestroy(e);
    if (tw->h_info) (void)hipHostFree(tw->h_info);
    if (tw->h_cursor) (void)hipHostFree(tw->h_cursor);
    delete tw;
    ctx->text_work = nullptr;
}

extern "C" {

int bns_dev_copy(bns_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BNS_OK;
}

static size_t text_piece_bytes(const bns_ctx *ctx, u64 bytes)
{
    size_t piece = (size_t)64 << 20;
    if (const char *e = std::getenv("BNS_TEXT_PIECE_MB")) { const long v = std::atol(e); if (v > 0) piece = (size_t)v << 20; }   // (measurements)
    if (ctx->dbg & BNS_DBG_SLICE_8K) piece = (size_t)8 << 10;
    const u64 least = ((bytes + MAX_PIECES - 1) / MAX_PIECES + 63) & ~63ULL;
    return (size_t)std::max<u64>(piece, least);
}

// host[0, bytes) -> u.buf, piece by piece on the copy stream
static int start_upload(bns_ctx *ctx, Upload &u, const char *host, u64 bytes)
{
    int rc = ensure(ctx, u.buf, (size_t)bytes + 256);
    if (rc != BNS_OK) return rc;
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    u.host = host; u.bytes = bytes; u.piece = text_piece_bytes(ctx, bytes);
    u.n_pieces = (u32)std::max<u64>(1, (bytes + u.piece - 1) / u.piece);
    for (u32 j = 0; j < u.n_pieces; ++j) {
        const u64 a = (u64)j * u.piece, b = std::min<u64>(bytes, a + u.piece);
        if (b > a) HIPCHK(ctx, hipMemcpyAsync((char *)u.buf.p + a, host + a, (size_t)(b - a), hipMemcpyHostToDevice, ctx->copy_stream));
        if (!u.ev[j]) HIPCHK(ctx, hipEventCreateWithFlags(&u.ev[j], hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(u.ev[j], ctx->copy_stream));
    }
    u.pending = true;
    return BNS_OK;
}

int bns_text_prefetch(bns_ctx *ctx, const char *const *text, const uint64_t *text_bytes, int n_streams)
{
    if (!ctx) return BNS_ERR_ARG;
    if (n_streams == 0) {                                       // forget what was prefetched (the caller is about to give the buffers up)
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (ctx->text_work) for (auto &row : ctx->text_work->up) for (Upload &u : row) { u.pending = false; u.host = nullptr; }
        return BNS_OK;
    }
    if (!text || !text_bytes || (n_streams != 1 && n_streams != 2)) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->text_work) ctx->text_work = new (std::nothrow) bns_text_work();
    if (!ctx->text_work) return BNS_ERR_NOMEM;
    TextWork &tw = *ctx->text_work;
    for (int s = 0; s < n_streams; ++s) {
        if (text_bytes[s] && !text[s]) return BNS_ERR_ARG;
        if (text_bytes[s] >= (1ULL << 31)) return fail(ctx, BNS_ERR_ARG, "bns_text_prefetch: at most 2^31 - 1 bytes of text per stream");
        Upload *u = !tw.up[s][0].pending ? &tw.up[s][0] : (!tw.up[s][1].pending ? &tw.up[s][1] : nullptr);
        if (!u) return fail(ctx, BNS_ERR_STATE, "bns_text_prefetch: two uploads are waiting for their bns_classify_text call already");
        const int rc = start_upload(ctx, *u, text[s], text_bytes[s]);
        if (rc != BNS_OK) return rc;
    }
    return BNS_OK;
}

int bns_classify_text(bns_ctx *ctx, const char *const *text, const uint64_t *text_bytes, int n_streams, uint64_t limit, int flags,
                      uint64_t cap_records, const bns_text_out *out, bns_text_info *info)
{
    if (!ctx || !text || !text_bytes || !out || !info || (n_streams != 1 && n_streams != 2)) return BNS_ERR_ARG;
    const bool parse_only = (flags & BNS_TEXT_PARSE_ONLY) != 0, on_device = (flags & BNS_TEXT_DEVICE) != 0;
    const bool final_text = (flags & BNS_TEXT_FINAL) != 0;
    int rc = ready(ctx, !parse_only, !parse_only);
    if (rc != BNS_OK) return rc;
    if (!parse_only && !out->taxon) return BNS_ERR_ARG;
    if ((out->run_start != nullptr) != (out->n_runs != nullptr)) return BNS_ERR_ARG;
    if (out->name_off && !out->names) return BNS_ERR_ARG;
    if ((out->run_tax != nullptr) != (out->run_len != nullptr) || (out->run_tax && !out->run_start)) return BNS_ERR_ARG;
    const u32 ns = (u32)n_streams;
    for (u32 s = 0; s < ns; ++s) {
        if (text_bytes[s] && !text[s]) return BNS_ERR_ARG;
        if (text_bytes[s] >= (1ULL << 31)) return fail(ctx, BNS_ERR_ARG, "bns_classify_text: at most 2^31 - 1 bytes of text per stream and call");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::memset(info, 0, sizeof(*info));
    if (!ctx->text_work) ctx->text_work = new (std::nothrow) bns_text_work();
    if (!ctx->text_work) return BNS_ERR_NOMEM;
    TextWork &tw = *ctx->text_work;
    hipStream_t st = ctx->stream;
    if (!tw.h_info) {
        HIPCHK(ctx, hipHostMalloc((void **)&tw.h_info, sizeof(CallInfo), hipHostMallocDefault));
        HIPCHK(ctx, hipHostMalloc((void **)&tw.h_cursor, 16, hipHostMallocDefault));
        HIPCHK(ctx, hipEventCreate(&tw.t0)); HIPCHK(ctx, hipEventCreate(&tw.t1));
        for (int q = 0; q < 2; ++q) {
            HIPCHK(ctx, hipEventCreateWithFlags(&tw.ev_done[q], hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreate(&tw.tc0[q])); HIPCHK(ctx, hipEventCreate(&tw.tc1[q]));
        }
    }
    // ---- where the text is: inside an upload that bns_text_prefetch started (at any offset `rel` of it), in one started now, or in
    // the caller's device memory.  Everything below works in the coordinates of that buffer -- whose base is aligned -- and the call's
    // own offsets come back as buffer offset - rel.  Piece k of the upload ends slice k: it is parsed -- together with what the
    // pieces in front of it left unfinished, which simply lies in front of it in the same buffer -- and classified while k + 1 ... travel.
    struct Src { const u8 *base = nullptr; u32 rel = 0, end = 0, j0 = 0, n = 1; u64 piece = 0; Upload *up = nullptr; } src[2];
    for (u32 s = 0; s < ns; ++s) {
        Src &q = src[s];
        if (on_device) {
            // (the kernels want an aligned base: the 64-byte boundary in front of the text, the text at offset rel of it)
            q.rel = (u32)((uintptr_t)text[s] & 63u);
            q.base = (const u8 *)text[s] - q.rel; q.end = q.rel + (u32)text_bytes[s];
            // text that is all there already is ONE slice (nothing travels that a second slice's parse could hide), unless pieces are asked for
            const bool pieces_forced = (ctx->dbg & BNS_DBG_SLICE_8K) || std::getenv("BNS_TEXT_PIECE_MB");
            q.piece = pieces_forced ? text_piece_bytes(ctx, text_bytes[s]) : std::max<u64>(64, ((u64)q.end + 63) & ~63ULL);
        } else {
            Upload *u = nullptr;
            for (Upload &c : tw.up[s])
                if (c.pending && c.host && text[s] >= c.host && text[s] + text_bytes[s] <= c.host + c.bytes) u = &c;     // (an empty text at its end included)
            if (!u) {
                u = !tw.up[s][0].pending ? &tw.up[s][0] : (!tw.up[s][1].pending ? &tw.up[s][1] : nullptr);
                if (!u) return fail(ctx, BNS_ERR_STATE, "bns_classify_text: two prefetched texts are waiting and this call's text is neither");
                if ((rc = start_upload(ctx, *u, text[s], text_bytes[s])) != BNS_OK) return rc;
            }
            q.up = u; q.base = (const u8 *)u->buf.p; q.rel = (u32)(text[s] - u->host); q.end = q.rel + (u32)text_bytes[s]; q.piece = u->piece;
        }
        q.j0 = (u32)(q.rel / q.piece);
        q.n = q.end > q.rel ? (u32)((q.end - 1) / q.piece) - q.j0 + 1 : 1;
    }
    const u32 n_slices = std::max(src[0].n, ns == 2 ? src[1].n : 1u);
    auto up_to = [&](u32 s, u32 k) -> u32 {                      // where stream s's text ends after slice k (buffer coordinates)
        const Src &q = src[s];
        return k + 1 >= q.n ? q.end : (u32)std::min<u64>(q.end, (u64)(q.j0 + k + 1) * q.piece);
    };
    auto release_uploads = [&] { for (u32 s = 0; s < ns; ++s) if (src[s].up) src[s].up->pending = false; };
    auto bail = [&](int code) { if (!on_device && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamSynchronize(st); if (ctx->back_stream) (void)hipStreamSynchronize(ctx->back_stream); release_uploads(); return code; };
    // every error exit from here on goes through bail(): uploads in flight are waited for and given up, the three streams drained
#define TXCHK(expr)                                                                                                            \
    do {                                                                                                                        \
        hipError_t _e = (expr);                                                                                                 \
        if (_e != hipSuccess) {                                                                                                 \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                                       \
            return bail(_e == hipErrorOutOfMemory ? BNS_ERR_NOMEM : BNS_ERR_HIP);                                               \
        }                                                                                                                       \
    } while (0)
    if ((out->words || out->nmask) && n_slices > 1) return bail(fail(ctx, BNS_ERR_ARG, "bns_classify_text: the packed words come back for one-slice calls only (<= 64 MiB of text)"));
    // the largest stretch one parse may cover: two slices' worth (a record longer than a slice is the host parser's)
    u64 range_cap = 0, call_text = 0;
    for (u32 s = 0; s < ns; ++s) { range_cap = std::max<u64>(range_cap, n_slices == 1 ? text_bytes[s] : 2 * src[s].piece + 64); call_text += text_bytes[s]; }
    const u32 cap_lines = (u32)(range_cap / 8 + 1024);          // (more lines than one per 8 bytes: BNS_TEXT_WHY_LINES)
    const u32 cap_rec = (u32)(range_cap / 16 + 512);            // (more records than one per 16 bytes: BNS_TEXT_WHY_LINES as well)
    // ---- batches.  A slice is what one parse covers (one upload piece and what the pieces in front left unfinished); a BATCH is what one
    // classify launch covers: the records of consecutive slices, appended to ONE packed image (offsets, words, flag words, the result
    // rows of the batch's set) until it holds `batch_reads` records (or batch_bases / batch_names), or the call ends.  The fused kernel
    // at 210 k records a launch -- one 64 MiB slice -- ran at a fifth of its rate (26 records per wavefront slot: launch, the first
    // fetches of a cold table and the tail are most of the 430 us); classifier.h:307-318 hands classify_seqs its batch whole, too.
    // Capacities are "what a batch may hold when it is flushed" + one slice's worst case: a slice is packed before its counts are known.
    u64 batch_reads = 2u << 20;
    if (const char *e = std::getenv("BNS_TEXT_BATCH_READS")) { const long long v = std::atoll(e); if (v > 0) batch_reads = (u64)v; }   // (measurements)
    if (ctx->dbg & BNS_DBG_BATCH_TINY) batch_reads = 64;         // (tests: many batches on small texts)
    const u64 batch_bases = 320u << 20, batch_names = 64u << 20;
    const u64 slice_reads_cap = (u64)cap_rec * ns, slice_bases_cap = range_cap * ns;
    const u64 carry_reads = n_slices == 1 ? 0 : std::min<u64>(batch_reads, call_text / 64 + 64);
    const u64 carry_bases = n_slices == 1 ? 0 : std::min<u64>(batch_bases, call_text);
    const u64 carry_names = n_slices == 1 ? 0 : std::min<u64>(batch_names, call_text);
    const u64 cap_reads = slice_reads_cap + carry_reads, cap_bases = slice_bases_cap + carry_bases, cap_names = slice_bases_cap + carry_names;
    if (cap_reads >= (1ULL << 31) || cap_bases >= (1ULL << 32) - (1ULL << 20)) return bail(fail(ctx, BNS_ERR_ARG, "bns_classify_text: text too large for one batch"));
    for (u32 s = 0; s < ns; ++s) {
        if ((rc = ensure(ctx, tw.ls[s], (size_t)cap_lines * 4 + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.role[s], (size_t)cap_lines + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.line_off[s], (size_t)cap_lines * 4 + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.hline[s], (size_t)cap_rec * 4 + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.tile[s], (size_t)(range_cap / TILE + 8) * 4)) != BNS_OK) return bail(rc);
    }
    // per slice (used up by the slice's own pack / names kernels): name_len, pos, line0, line1, single
    for (int i = 0; i < 5; ++i) if ((rc = ensure(ctx, tw.rec_slice[i], (size_t)slice_reads_cap * 4 + 64)) != BNS_OK) return bail(rc);
    for (int q = 0; q < 2; ++q) {
        if ((rc = ensure(ctx, tw.seq_len[q], (size_t)cap_reads * 4 + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.name_off[q], (size_t)(cap_reads + 1) * 4)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.pos64[q], (size_t)cap_reads * 8)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.names[q], (size_t)cap_names + 64)) != BNS_OK) return bail(rc);
    }
    if ((rc = ensure(ctx, tw.offsets, (size_t)(cap_reads + 1) * 8)) != BNS_OK) return bail(rc);
    if ((rc = ensure(ctx, tw.words, ((size_t)cap_bases / 32 + cap_reads + 2) * 8)) != BNS_OK) return bail(rc);
    if ((rc = ensure(ctx, tw.nmask, ((size_t)cap_bases / 32 + cap_reads + 2) * 4)) != BNS_OK) return bail(rc);
    if ((rc = ensure(ctx, tw.info, sizeof(CallInfo))) != BNS_OK) return bail(rc);
    const bool want_runs = out->run_start != nullptr && !parse_only;
    if (!parse_only) {
        for (int q = 0; q < 2; ++q) for (int i = 0; i < 4; ++i) if ((rc = ensure(ctx, tw.out[q][i], (size_t)cap_reads * 4 + 64)) != BNS_OK) return bail(rc);
        if (want_runs) {
            if ((rc = ensure(ctx, tw.hits, (size_t)cap_bases * 4 + 64)) != BNS_OK) return bail(rc);
            for (int q = 0; q < 2; ++q) {
                if ((rc = ensure(ctx, tw.runs[q][0], (size_t)cap_reads * 8 + 64)) != BNS_OK) return bail(rc);
                if ((rc = ensure(ctx, tw.runs[q][1], (size_t)cap_reads * 4 + 64)) != BNS_OK) return bail(rc);
                if ((rc = ensure(ctx, tw.runs[q][2], (size_t)cap_bases * 4 + 64)) != BNS_OK) return bail(rc);
                if ((rc = ensure(ctx, tw.runs[q][3], (size_t)cap_bases * 4 + 64)) != BNS_OK) return bail(rc);
            }
        }
    }
    TXCHK(hipStreamSynchronize(st));                            // (workspaces of an earlier call on this stream are free now)
    const u8 *d_text[2] = {src[0].base, src[1].base};

    if (want_runs) TXCHK(hipMemsetAsync(&((SmallLayout *)ctx->small.p)->runs_cursor, 0, 8, st));
    CallInfo *d_ci = (CallInfo *)tw.info.p;
    if (!ctx->back_stream) TXCHK(hipStreamCreateWithFlags(&ctx->back_stream, hipStreamNonBlocking));
    hipStream_t bs = ctx->back_stream;
    // Results leave on the back stream: batch b's arrays (set b & 1) are copied out while batch b + 1 is parsed and classified into the
    // other set.  Before a batch is classified the back stream is drained -- nothing of it then reads the set that batch writes (its
    // last user was two batches ago, and its parse started behind the drain of the batch in between) -- and the run count of the batch
    // in front is known: its runs are copied now.
    u32 batch_no = 0;                                           // batches classified so far
    u64 runs_done = 0;                                          // runs whose copy to the host has been queued
    bool runs_overflow = false;                                 // the caller's run arrays are full: the batch whose runs did not fit (and what follows) is not his
    // accepted so far (slices parsed and counted, classified or waiting in the open batch) / the open batch / the state in front of the
    // last classified batch
    u64 done_reads = 0, names_done = 0, bases_done = 0;
    u32 cons[2] = {src[0].rel, src[1].rel};
    u64 acc_reads = 0, acc_bases = 0, acc_names = 0; u32 acc_max_len = 0;
    u64 open_reads0 = 0, open_names0 = 0, open_bases0 = 0; u32 open_cons0[2] = {cons[0], cons[1]};             // where the open batch began
    u64 reads_before_prev = 0, names_before_prev = 0, bases_before_prev = 0; u32 cons_before_prev[2] = {cons[0], cons[1]};
    auto flush_runs_of_prev = [&]() -> int {                    // (after a drain of the back stream) the runs of batch batch_no - 1
        if (!want_runs || batch_no == 0) return BNS_OK;
        const u32 pq = (batch_no - 1u) & 1u;
        const u64 n_tot = tw.h_cursor[pq];                      // runs so far, that batch's included
        if (out->run_tax) {                                     // the caller's own arrays
            if (n_tot > out->runs_cap) { runs_overflow = true; return BNS_OK; }
            if (n_tot > runs_done) {
                HIPCHK(ctx, hipMemcpyAsync(out->run_tax + runs_done, tw.runs[pq][2].p, (size_t)(n_tot - runs_done) * 4, hipMemcpyDeviceToHost, bs));
                HIPCHK(ctx, hipMemcpyAsync(out->run_len + runs_done, tw.runs[pq][3].p, (size_t)(n_tot - runs_done) * 4, hipMemcpyDeviceToHost, bs));
            }
            runs_done = n_tot;
            return BNS_OK;
        }
        if (ctx->h_run_cap < n_tot) {
            const size_t want = (size_t)n_tot + (size_t)n_tot / 2 + 1024;
            u32 *nt = nullptr, *nl = nullptr;
            HIPCHK(ctx, hipHostMalloc((void **)&nt, want * 4, hipHostMallocDefault));
            HIPCHK(ctx, hipHostMalloc((void **)&nl, want * 4, hipHostMallocDefault));
            if (runs_done) { std::memcpy(nt, ctx->h_run_tax, (size_t)runs_done * 4); std::memcpy(nl, ctx->h_run_len, (size_t)runs_done * 4); }
            if (ctx->h_run_tax) (void)hipHostFree(ctx->h_run_tax);
            if (ctx->h_run_len) (void)hipHostFree(ctx->h_run_len);
            ctx->h_run_tax = nt; ctx->h_run_len = nl; ctx->h_run_cap = want;
        }
        if (n_tot > runs_done) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_run_tax + runs_done, tw.runs[pq][2].p, (size_t)(n_tot - runs_done) * 4, hipMemcpyDeviceToHost, bs));
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_run_len + runs_done, tw.runs[pq][3].p, (size_t)(n_tot - runs_done) * 4, hipMemcpyDeviceToHost, bs));
        }
        runs_done = n_tot;
        return BNS_OK;
    };
    const u32 lim = limit >= text_bytes[0] ? 0xFFFFFFFFu : (u32)limit + src[0].rel;
    const unsigned pgrid = (unsigned)ctx->n_cu * 8;
    int status = BNS_TEXT_OK;
    u32 why = 0, n_launches = 0;
    float ms_parse = 0, ms_classify = 0;
    bool rolled_back = false;
    auto roll_back = [&] {                                      // the last classified batch is not the caller's after all (its runs did not fit his arrays): nor is what came behind it
        if (rolled_back) return;
        rolled_back = true;
        done_reads = reads_before_prev; names_done = names_before_prev; bases_done = bases_before_prev;
        for (u32 s = 0; s < ns; ++s) cons[s] = cons_before_prev[s];
        acc_reads = acc_bases = acc_names = 0; acc_max_len = 0;
    };
    // ---- the open batch -> one classify launch; its results behind those of the batches in front
    auto flush_batch = [&]() -> int {
        if (rolled_back || acc_reads == 0) return BNS_OK;
        const u32 q = batch_no & 1u;
        const u64 n_reads = acc_reads, n_units = acc_reads / ns, u_done = open_reads0 / ns;
        // (the back stream drained: the out / runs arrays of set q are free, and the run count of the batch in front is on the host)
        HIPCHK(ctx, hipStreamSynchronize(bs));
        if (batch_no && ctx->timing && !parse_only) { float ms = 0; if (hipEventElapsedTime(&ms, tw.tc0[q ^ 1u], tw.tc1[q ^ 1u]) == hipSuccess) ms_classify += ms; }
        int frc = flush_runs_of_prev();
        if (frc != BNS_OK) return frc;
        if (runs_overflow) { roll_back(); status = BNS_TEXT_CAP; return BNS_OK; }
        u32 *o0 = (u32 *)tw.out[q][0].p, *o1 = (u32 *)tw.out[q][1].p, *o2 = (u32 *)tw.out[q][2].p, *o3 = (u32 *)tw.out[q][3].p;
        unsigned long long *d_cur = &((SmallLayout *)ctx->small.p)->runs_cursor;     // (zeroed at the start of the call: it runs on over the batches)
        if (!parse_only) {
            if (ctx->timing) HIPCHK(ctx, hipEventRecord(tw.tc0[q], st));
            frc = classify_device_impl(ctx, nullptr, (const u64 *)tw.words.p, (const u32 *)tw.nmask.p, (const u64 *)tw.offsets.p, n_reads, acc_bases,
                                       std::max<u32>(acc_max_len, 1u), ns == 2 ? 1 : 0, o0, out->missing || want_runs ? o1 : nullptr,
                                       out->ambig || want_runs ? o2 : nullptr, (out->n_hits || want_runs) ? o3 : nullptr, want_runs ? (u32 *)tw.hits.p : nullptr, st);
            if (frc != BNS_OK) return frc;
            ++n_launches;
            if (ctx->timing) HIPCHK(ctx, hipEventRecord(tw.tc1[q], st));
            if (want_runs) {
                hipLaunchKernelGGL(hit_runs_kernel, dim3(grid_for(ctx, (n_units + HIT_RUNS_GROUP - 1) / HIT_RUNS_GROUP, 4)), dim3(256), 0, st, (const u32 *)tw.hits.p,
                                   (const u64 *)tw.offsets.p, ns, (const u32 *)o3, (u64)n_units, (u64 *)tw.runs[q][0].p, (u32 *)tw.runs[q][1].p,
                                   (u32 *)tw.runs[q][2].p - runs_done, (u32 *)tw.runs[q][3].p - runs_done, d_cur);
                HIPCHK(ctx, hipGetLastError());
            }
        }
        // the copies, on the back stream behind this batch's kernels; the next batch is parsed and classified meanwhile
        HIPCHK(ctx, hipEventRecord(tw.ev_done[q], st));
        HIPCHK(ctx, hipStreamWaitEvent(bs, tw.ev_done[q], 0));
        if (!parse_only) {
            HIPCHK(ctx, hipMemcpyAsync(out->taxon + u_done, o0, (size_t)n_units * 4, hipMemcpyDeviceToHost, bs));
            if (out->missing) HIPCHK(ctx, hipMemcpyAsync(out->missing + u_done, o1, (size_t)n_units * 4, hipMemcpyDeviceToHost, bs));
            if (out->ambig) HIPCHK(ctx, hipMemcpyAsync(out->ambig + u_done, o2, (size_t)n_units * 4, hipMemcpyDeviceToHost, bs));
            if (out->n_hits) HIPCHK(ctx, hipMemcpyAsync(out->n_hits + u_done, o3, (size_t)n_units * 4, hipMemcpyDeviceToHost, bs));
            if (want_runs) {
                HIPCHK(ctx, hipMemcpyAsync(out->run_start + u_done, tw.runs[q][0].p, (size_t)n_units * 8, hipMemcpyDeviceToHost, bs));
                HIPCHK(ctx, hipMemcpyAsync(out->n_runs + u_done, tw.runs[q][1].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, bs));
                HIPCHK(ctx, hipMemcpyAsync(&tw.h_cursor[q], d_cur, 8, hipMemcpyDeviceToHost, bs));
            }
        }
        if (out->seq_len) HIPCHK(ctx, hipMemcpyAsync(out->seq_len + open_reads0, tw.seq_len[q].p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, bs));
        if (out->rec_pos) HIPCHK(ctx, hipMemcpyAsync(out->rec_pos + open_reads0, tw.pos64[q].p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, bs));
        if (out->name_off) {
            HIPCHK(ctx, hipMemcpyAsync(out->name_off + open_reads0, tw.name_off[q].p, (size_t)(n_reads + 1) * 4, hipMemcpyDeviceToHost, bs));
            if (acc_names) HIPCHK(ctx, hipMemcpyAsync(out->names + open_names0, tw.names[q].p, (size_t)acc_names, hipMemcpyDeviceToHost, bs));
        }
        if (out->words) HIPCHK(ctx, hipMemcpyAsync(out->words, tw.words.p, (size_t)bns_packed_words(acc_bases, n_reads) * 8, hipMemcpyDeviceToHost, bs));
        if (out->nmask) HIPCHK(ctx, hipMemcpyAsync(out->nmask, tw.nmask.p, (size_t)bns_packed_words(acc_bases, n_reads) * 4, hipMemcpyDeviceToHost, bs));
        ++batch_no;
        reads_before_prev = open_reads0; names_before_prev = open_names0; bases_before_prev = open_bases0;
        for (u32 s = 0; s < ns; ++s) cons_before_prev[s] = open_cons0[s];
        acc_reads = acc_bases = acc_names = 0; acc_max_len = 0;
        return BNS_OK;
    };
    // One round = one parse over [cons, hi) of every stream, hi = what piece k has brought up -- cut to the window one parse may
    // cover (of a pair of files the denser one is ahead of what its mate lets it hand over: its unparsed text waits, it does not grow
    // the window) -- whose records are appended to the open batch.  k moves on with the uploads; when they are all up the rounds go on
    // until the text is used up.
    const u32 window = (u32)std::min<u64>(range_cap - 64, 0x7FFFFFFFu);
    u32 k = 0, rounds = 0;
    for (;; ++rounds) {
        // room for one more slice's worst case?  (else the open batch goes first)
        if (acc_reads + slice_reads_cap > cap_reads || acc_bases + slice_bases_cap > cap_bases || acc_names + slice_bases_cap > cap_names) {
            if ((rc = flush_batch()) != BNS_OK) return bail(rc);
            if (rolled_back) break;
        }
        u32 hi[2] = {0, 0};
        bool last = true, capped = false;
        for (u32 s = 0; s < ns; ++s) {
            hi[s] = up_to(s, k);
            if (n_slices > 1 && hi[s] - cons[s] > window) { hi[s] = cons[s] + window; capped = true; }
            if (hi[s] != src[s].end) last = false;
        }
        const int fin = (last && final_text) ? 1 : 0;
        const u32 q = batch_no & 1u;                            // the set of result arrays the open batch writes
        const u32 R0 = (u32)acc_reads;
        RecArrays ra{(u32 *)tw.seq_len[q].p + R0, (u32 *)tw.rec_slice[0].p, (u32 *)tw.rec_slice[1].p, (u32 *)tw.rec_slice[2].p, (u32 *)tw.rec_slice[3].p,
                     (u32 *)tw.rec_slice[4].p};
        if (ctx->timing) TXCHK(hipEventRecord(tw.t0, st));
        TXCHK(hipMemsetAsync(d_ci, 0, sizeof(CallInfo), st));
        for (u32 s = 0; s < ns; ++s) {
            // (the piece that ends this slice -- and with it every piece in front of it: the copy stream is in order)
            if (src[s].up) TXCHK(hipStreamWaitEvent(st, src[s].up->ev[std::min(src[s].j0 + k, src[s].j0 + src[s].n - 1)], 0));
            const u32 lo = cons[s];
            const u32 tile0 = lo / TILE, n_tiles = hi[s] > lo ? (hi[s] - 1) / TILE - tile0 + 1 : 1;
            u32 *tile = (u32 *)tw.tile[s].p;
            StreamInfo *d_si = &d_ci->s[s];
            hipLaunchKernelGGL(text_count_kernel, dim3(n_tiles), dim3(256), 0, st, d_text[s], lo, hi[s], tile0, tile);
            hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, st, tile, n_tiles, (u32 *)nullptr);
            hipLaunchKernelGGL(line_write_kernel, dim3(n_tiles), dim3(256), 0, st, d_text[s], lo, hi[s], tile0, n_tiles, (const u32 *)tile, (u32 *)tw.ls[s].p,
                               cap_lines, d_si);
            hipLaunchKernelGGL(line_role_kernel, dim3(pgrid), dim3(256), 0, st, d_text[s], (const u32 *)tw.ls[s].p, d_si, (u8 *)tw.role[s].p);
            TXCHK(hipGetLastError());
            if ((rc = device_scan(ctx, tw, st, InIsHeader{(const u8 *)tw.role[s].p}, &d_si->n_lines, 1u, cap_lines,
                                  OutHeaderLines{(u32 *)tw.hline[s].p, cap_rec, d_si})) != BNS_OK) return bail(rc);
        }
        hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(1), 0, st, d_ci, ns, lim, fin, (const u32 *)tw.ls[0].p, (const u32 *)tw.hline[0].p,
                           (const u8 *)tw.role[0].p, (const u32 *)tw.ls[1].p, (const u32 *)tw.hline[1].p, (const u8 *)tw.role[1].p);
        for (u32 s = 0; s < ns; ++s)
            hipLaunchKernelGGL(record_kernel, dim3(pgrid), dim3(256), 0, st, d_text[s], (const u32 *)tw.ls[s].p, (const u8 *)tw.role[s].p,
                               (const u32 *)tw.hline[s].p, (u32 *)tw.line_off[s].p, d_ci, s, ns, (flags & BNS_TEXT_TRIM_READNO) ? 1 : 0, ra);
        TXCHK(hipGetLastError());
        // the slice's records go behind those the open batch holds: offsets from acc_bases on, names from names_done on
        u64 *d_off = (u64 *)tw.offsets.p + R0;
        u32 *d_name_off = (u32 *)tw.name_off[q].p + R0;
        if ((rc = device_scan(ctx, tw, st, InU32{ra.seq_len}, &d_ci->n_reads, 1u, (u32)slice_reads_cap, OutOffsets64{d_off, acc_bases, &d_ci->total_bases})) != BNS_OK) return bail(rc);
        if ((rc = device_scan(ctx, tw, st, InU32{ra.name_len}, &d_ci->n_reads, 1u, (u32)slice_reads_cap,
                              OutOffsets32{d_name_off, (u32)names_done, &d_ci->names_bytes})) != BNS_OK) return bail(rc);
        PackSrc p0{d_text[0], (const u32 *)tw.ls[0].p, (const u32 *)tw.line_off[0].p}, p1{d_text[1], (const u32 *)tw.ls[1].p, (const u32 *)tw.line_off[1].p};
        hipLaunchKernelGGL(pack_text_kernel, dim3(pgrid), dim3(256), 0, st, p0, p1, ns, ra, (const u64 *)d_off, R0, (const CallInfo *)d_ci, (u64 *)tw.words.p,
                           (u32 *)tw.nmask.p);
        hipLaunchKernelGGL(names_kernel, dim3(pgrid), dim3(256), 0, st, d_text[0], d_text[1], ns, ra, (const u32 *)d_name_off, (u32)(names_done - acc_names),
                           (const CallInfo *)d_ci, (char *)tw.names[q].p, (u64 *)tw.pos64[q].p + R0, src[0].rel, src[1].rel);
        TXCHK(hipGetLastError());
        if (ctx->timing) TXCHK(hipEventRecord(tw.t1, st));
        TXCHK(hipMemcpyAsync(tw.h_info, d_ci, sizeof(CallInfo), hipMemcpyDeviceToHost, st));
        TXCHK(hipStreamSynchronize(st));
        const CallInfo ci = *tw.h_info;
        if (ctx->timing) { float ms = 0; if (hipEventElapsedTime(&ms, tw.t0, tw.t1) == hipSuccess) ms_parse += ms; }
        if (ci.why) { status = BNS_TEXT_IRREGULAR; why = ci.why; break; }
        const u64 n_reads = ci.n_reads;
        if (n_reads == 0) {
            // nothing complete in this stretch: more text may complete it (the next slice is parsed together with this one).  At the
            // end of the text: a final text is done (blank lines; a pair whose one file has run out; headers behind the limit only);
            // otherwise the caller has handed over less than one record
            if (!last && k + 1 < n_slices) { ++k; continue; }
            if (!last && capped) { status = BNS_TEXT_NO_RECORD; break; }      // (a whole window without one complete record)
            for (u32 s = 0; s < ns; ++s) cons[s] = ci.s[s].consumed;
            const bool behind_limit = lim != 0xFFFFFFFFu && ci.s[0].n_hdr && cons[0] >= lim;
            if (!fin && !behind_limit) status = BNS_TEXT_NO_RECORD;
            break;
        }
        if (done_reads + n_reads > cap_records || (out->names && names_done + ci.names_bytes > out->names_cap)) { status = BNS_TEXT_CAP; break; }
        // ---- the slice is the open batch's (a batch opens with its first slice)
        if (acc_reads == 0) { open_reads0 = done_reads; open_names0 = names_done; open_bases0 = bases_done; for (u32 s = 0; s < ns; ++s) open_cons0[s] = cons[s]; }
        acc_reads += n_reads; acc_bases += ci.total_bases; acc_names += ci.names_bytes; acc_max_len = std::max(acc_max_len, ci.max_len);
        done_reads += n_reads; names_done += ci.names_bytes; bases_done += ci.total_bases;
        for (u32 s = 0; s < ns; ++s) cons[s] = ci.s[s].consumed;
        if (acc_reads >= batch_reads || acc_bases >= batch_bases || acc_names >= batch_names) {
            if ((rc = flush_batch()) != BNS_OK) return bail(rc);
            if (rolled_back) break;
        }
        // a stream that has handed over everything in front of the limit is done (the rest is the next stretch's)
        if (lim != 0xFFFFFFFFu && cons[0] >= lim) break;
        if (last) break;
        if (k + 1 < n_slices) ++k;
    }
    // the open batch (whatever ended the rounds, the records in front of that are the caller's); then what is still on its way: the
    // last batch's arrays, then its runs
    if ((rc = flush_batch()) != BNS_OK) return bail(rc);
    TXCHK(hipStreamSynchronize(st));
    TXCHK(hipStreamSynchronize(bs));
    if (batch_no && ctx->timing && !parse_only) { float ms = 0; if (hipEventElapsedTime(&ms, tw.tc0[(batch_no - 1u) & 1u], tw.tc1[(batch_no - 1u) & 1u]) == hipSuccess) ms_classify += ms; }
    if (!rolled_back) {
        if ((rc = flush_runs_of_prev()) != BNS_OK) return bail(rc);
        if (runs_overflow) { roll_back(); status = BNS_TEXT_CAP; }
    }
    TXCHK(hipStreamSynchronize(bs));
    if (!on_device && ctx->copy_stream) {
        // the caller's buffers are his again -- those of THIS call: an upload prefetched for the next one keeps travelling
        bool other_pending = false;
        for (u32 s = 0; s < ns; ++s) for (Upload &c : tw.up[s]) if (c.pending && &c != src[s].up) other_pending = true;
        if (!other_pending) TXCHK(hipStreamSynchronize(ctx->copy_stream));
        else for (u32 s = 0; s < ns; ++s) if (src[s].up) TXCHK(hipEventSynchronize(src[s].up->ev[src[s].up->n_pieces - 1]));
    }
#undef TXCHK
    release_uploads();
    info->n_records = done_reads;
    for (u32 s = 0; s < ns; ++s) info->consumed[s] = cons[s] - src[s].rel;
    info->total_bases = bases_done; info->names_bytes = names_done; info->n_runs_total = runs_done;
    info->run_tax = out->run_tax ? out->run_tax : ctx->h_run_tax; info->run_len = out->run_len ? out->run_len : ctx->h_run_len;
    info->status = status; info->why = why; info->n_slices = rounds + 1; info->n_launches = n_launches;
    info->ms_parse = ms_parse; info->ms_classify = ms_classify;
    return BNS_OK;
}

}  // extern "C"
