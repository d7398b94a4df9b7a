// bns_ingest.hip -- FASTA / FASTQ text parsed on the device (gfx950, wave64): the host-ingest row of SURVEY 8f-2 without a host
// parser.  Included at the end of bns_api.hip (one translation unit: bns_ctx, HIPCHK, ensure, classify_device_impl).
//
// Replaces kseq_read (klib/kseq.h:177-225) + bseq_read's loop (kseq_declare.h:112-145) for text in the REGULAR FORM -- the
// form in which kseq_read's character-level state machine is a function of whole lines:
//
//     stream  := blank* record*
//     record  := H (S | blank)* [ P Q+ blank* ]
//     H  a line whose first byte is '>' or '@'          S  a non-empty line whose first byte is none of '>' '@' '+'
//     P  a line whose first byte is '+'                 Q  the lines behind a P, whatever they start with, until they are together
//                                                          as long as the S lines together (round 6: one line or several)
//
// Why kseq_read yields exactly these records on such text (klib/kseq.h line numbers):
//   * :183-186 skips to the next '>' / '@' CHARACTER; with only blank lines in front of H that is H's first byte.
//   * :190-191 name = H up to the first isspace byte ('\r' is one); the rest of the line is the comment.
//   * :197-201 reads lines until one STARTS with '>', '+' or '@' (first byte tested, '\n' skipped: blank lines), appending
//     every other line whole: the S lines.  '>' / '@' ends a FASTA record (:202) -- the next H; '+' (:210) is P.
//   * :215 skips the rest of P; :217 appends quality LINES until they are as long as the sequence: regular when the running
//     length meets the sequence's exactly at a line end (shorter at the end of the input, or longer: error -2 at :220 -- not regular).
//   * :135 (ks_getuntil2, line mode): a line that is appended loses ONE trailing '\r' when the string it was appended to is then
//     longer than one byte -- so CRLF text reads like LF text, except that a lone "\r" line in front of any sequence (or quality)
//     byte stays in the string: a sequence with an invalid base in front.  Reproduced below (round 6; rounds 5 handed every '\r' back).
// A quality line may start with '@', '>' or '+', and with several of them per record no line-local rule tells a header from a
// quality line.  So EVERY line that starts with '>' or '@' is a CANDIDATE header: one lane walks the record that would start there
// (walk_kernel: S lines, P, Q lines until the lengths meet, blank lines, the line the next record starts at) and marks the
// candidates inside its own quality lines.  A candidate that nobody marks IS a header: the true records tile the text, so a
// candidate that is not one lies inside a true record -- whose walk marks it.  A marked candidate is decided by following the walks'
// "next" links from the nearest unmarked candidate in front of it (compact_kernel): a few hops, only where quality lines look like
// headers.  Everything the grammar does not allow (text between records, quality of the wrong length) is found by the true
// records' walks and reported as BNS_TEXT_IRREGULAR: nothing is guessed, the caller's host parser takes that stretch.
//
// Kernels (all HBM-bound byte / integer work; round 6: prefix sums inside their producers -- two-level sums for the lines, decoupled
// look-back over ~100 blocks for the records -- 6 kernel launches per slice, whether one file or a pair, instead of 15 / 22):
//   count_kernel,      '\n' and candidate headers per 16 KiB tile (64 bytes per lane as 4 x dwordx4, SWAR byte compares); line starts and
//   write_kernel       candidates written from the same masks, a tile's place from the counts in front (two-level sums, no scan launch)
//   walk_kernel        one lane per candidate: kseq_read's record from there -- name, sequence length, where it ends, what is wrong with it
//   compact_kernel     candidates -> records (true headers in order; look-back)
//   offsets_kernel     how many records this call takes (complete ones in front of `limit`; the minimum over a pair of files), then
//                      sequence lengths -> base offsets, name lengths -> name offsets (look-back)
//   pack_text_kernel   one wavefront per record: bases gathered from the record's lines -> 2-bit words + invalid-base flags; its name
// then classify_device_impl on the packed words, and hit_runs_kernel when the caller prints runs.
#include <new>

namespace bns {
namespace ingest {

constexpr u32 TILE = 16384;                     // text bytes per pass of a 256-thread block (64 per lane)
constexpr u32 SUB = 16;                         // lanes per record in pack_text_kernel (four records per wavefront)
constexpr u32 REC_ITEMS = 8;                    // candidates / records per thread of a scan block (2048 per block)
constexpr u32 REC_BLOCK = 256u * REC_ITEMS;
constexpr u32 MAX_REC_LINES = 4096;
constexpr u32 WALK_INCOMPLETE = 0x80000000u;    // (c_flags) the walk ran into the end of a text that is not final

struct StreamInfo {
    u32 n_nl, n_lines, n_cand, n_hdr;           // n_cand: lines that start with '>' / '@'; n_hdr: those that are headers
    u32 n_take, consumed, why, lo;
    u32 hi, n_eff, n_real, any_inside;          // n_eff: headers that yield a record (a final text that ends in a bare '>' / '@' byte: that one does not)
    u32 why_all, pad[3];                        // n_real: lines that may be looked at (not the unfinished last line of a text that goes on; not the nothing behind a final '\n')
};                                              // any_inside: a walk found candidates inside its record; why_all: what the walks found wrong, all candidates
struct CallInfo {
    StreamInfo s[2];
    u32 n_take;                                 // records per stream this call takes
    u32 n_reads;                                // n_take * n_streams
    u32 max_len, why;
    u32 total_bases, names_bytes;
    u32 ticket[6];                              // block tickets: lines_kernel [0..1], compact_kernel [2..3], offsets_kernel [4]; [5]: compact blocks that are done
    u32 fast, pad;                              // every candidate of either stream is a header: record r is candidate r
};

// what one stream's kernels work on (buffer coordinates: text + lo .. text + hi)
struct StreamArgs {
    const u8 *text;
    u32 lo, hi, tile0, n_tiles, cap_lines, cap_rec;
    u32 *ls, *line_off;                         // per line: where it starts; bases of its record in front of it (sequence lines)
    u32 *cand;                                  // per candidate: its line
    u32 *c_next, *c_seq, *c_name, *c_line1, *c_single, *c_flags, *c_inside, *c_pos;    // per candidate: walk_kernel's findings
    u32 *rec_cand;                              // per record: its candidate
    u64 *st_lines, *st_groups, *st_compact;     // '\n' / candidate counts per tile and per group of 64 tiles; look-back words of compact_kernel
};
struct ParseArgs {
    StreamArgs s[2];
    CallInfo *ci;
    u64 *st_offsets;
    u32 n_streams;
    int final_text, trim_readno;
};

// ---- 256-thread block helpers -------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 shfl_up64(u64 v, int off) { return ((u64)(u32)__shfl_up((int)(v >> 32), off) << 32) | (u32)__shfl_up((int)(u32)v, off); }
__device__ __forceinline__ u64 shfl_xor64(u64 v, int off) { return ((u64)(u32)__shfl_xor((int)(v >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)v, off); }
// exclusive prefix of v over the block's 256 threads; total = the block's sum (same in every thread).  v may hold two 31-bit
// counters (bits 0-30 and 31-61): sums stay below 2^31 each.
__device__ __forceinline__ u64 block_excl_scan(u64 v, u64 &total, u64 *lds4)
{
    const u32 lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    u64 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u64 t = shfl_up64(incl, off); if (lane >= (u32)off) incl += t; }
    __syncthreads();                                           // (lds4 may still be read from a previous use)
    if (lane == 63) lds4[w] = incl;
    __syncthreads();
    u64 base = 0;
    total = 0;
#pragma unroll
    for (u32 i = 0; i < 4; ++i) { const u64 t = lds4[i]; if (i < w) base += t; total += t; }
    return base + incl - v;
}

// Decoupled look-back over the blocks of one scan: state[b] = status << 62 | value; a block publishes its own sum (status 1), adds up the
// words of the blocks in front -- 64 at a time, one per lane of the first wavefront -- until it meets one that holds an inclusive prefix
// (status 2), and publishes its own.  Blocks take their numbers from a ticket counter, so every block in front of a waiting one is running
// or done.  -> the sum over the blocks in front (in every thread).
constexpr u64 ST_VAL = (1ULL << 62) - 1ULL;
__device__ __forceinline__ u64 lookback(u64 *__restrict__ state, u32 b, u64 sum, u64 *lds_pre)
{
    if (threadIdx.x < 64) {
        const u32 lane = threadIdx.x;
        u64 excl = 0;
        if (b == 0) { if (lane == 0) __hip_atomic_store(&state[0], (2ULL << 62) | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&state[b], (1ULL << 62) | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int pos = (int)b - 1;
            for (;;) {
                const int idx = pos - (int)lane;
                const u64 v = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ULL << 62);
                const u32 st = (u32)(v >> 62);
                const u64 b_inc = __ballot(st == 2u), b_emp = __ballot(st == 0u);
                const u32 first = b_inc ? (u32)__builtin_ctzll(b_inc) : 64u;
                const u64 need = first < 63u ? ((2ULL << first) - 1ULL) : ~0ULL;
                if (b_emp & need) { __builtin_amdgcn_s_sleep(1); continue; }      // (a block in front has not published yet)
                u64 c = lane <= first ? (v & ST_VAL) : 0ULL;
#pragma unroll
                for (int off = 32; off; off >>= 1) c += shfl_xor64(c, off);
                excl += c;
                if (first < 64u) break;
                pos -= 64;
            }
            if (lane == 0) __hip_atomic_store(&state[b], (2ULL << 62) | (excl + sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) *lds_pre = excl;
    }
    __syncthreads();
    return *lds_pre;
}

// 4-bit mask of the bytes of w that equal the byte repeated in `pat` (exact zero-byte test of w ^ pat, bits gathered by one multiply)
__device__ __forceinline__ u32 eq_nibble(u32 w, u32 pat)
{
    const u32 x = w ^ pat;
    const u32 t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    const u32 z = ~(t | 0x7F7F7F7Fu);                          // 0x80 in every byte that matched
    return (((z >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// masks of the 64 bytes at text + base (base a multiple of 64): m = '\n', h = '>' or '@'; only offsets in [lo, hi) count
__device__ __forceinline__ void masks64(const u8 *__restrict__ text, u32 base, u32 lo, u32 hi, u64 &m, u64 &h)
{
    m = h = 0;
    if (base >= hi || base + 64u <= lo) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(text + base);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 v = p[j];
        const u32 n = eq_nibble(v.x, 0x0A0A0A0Au) | (eq_nibble(v.y, 0x0A0A0A0Au) << 4) | (eq_nibble(v.z, 0x0A0A0A0Au) << 8) | (eq_nibble(v.w, 0x0A0A0A0Au) << 12);
        const u32 g = (eq_nibble(v.x, 0x3E3E3E3Eu) | eq_nibble(v.x, 0x40404040u)) | ((eq_nibble(v.y, 0x3E3E3E3Eu) | eq_nibble(v.y, 0x40404040u)) << 4) |
                      ((eq_nibble(v.z, 0x3E3E3E3Eu) | eq_nibble(v.z, 0x40404040u)) << 8) | ((eq_nibble(v.w, 0x3E3E3E3Eu) | eq_nibble(v.w, 0x40404040u)) << 12);
        m |= (u64)n << (16 * j);
        h |= (u64)g << (16 * j);
    }
    u64 keep = ~0ULL;
    if (base < lo) keep &= ~0ULL << (lo - base);
    if (hi - base < 64u) keep &= (1ULL << (hi - base)) - 1ULL;
    m &= keep; h &= keep;
}
__device__ __forceinline__ bool is_hdr_byte(u8 c) { return c == '>' || c == '@'; }

// ---- lines ---------------------------------------------------------------------------------------------------------------------
// Two launches without a dependency between blocks (round 6 first had ONE kernel with a look-back over its tiles: 40 us per 64 MiB on an
// idle device, 2.4 ms per 195 MiB beside the inflate kernels -- a chain of ticket, loads, barriers and polls per tile, every step of it
// slowed; profiles/r06_bgzf_trace2.txt):
//   count_kernel   per 16 KiB tile: '\n' and candidate lines (a '>' / '@' byte behind a '\n') counted from SWAR bit masks of 64 bytes per
//                  lane -> tile_cnt[t], and added to the sum of the tile's group of 64
//   write_kernel   a tile's place among the lines = the group sums in front + the tile counts of its own group in front (two rounds of
//                  loads, one lane each); the masks again (the text is in the cache), then ls[j + 1] = offset behind the j-th '\n' of
//                  [lo, hi) and cand[] = the lines whose first byte is '>' / '@'
// ls[0] = lo; ls[n_nl + 1] = hi + 1 (so that "length of line i" = ls[i + 1] - 1 - ls[i] also holds for a last line without a newline).
// (counts: '\n' in bits 0-30, candidates in bits 31-61)
__device__ __forceinline__ void tile_masks(const StreamArgs &a, u32 base, u64 &m, u64 &cm, u32 &fc)
{
    u64 h;
    masks64(a.text, base, a.lo, a.hi, m, h);
    cm = m & (h >> 1);                                          // the '\n' that a candidate line follows
    if ((m >> 63) && base + 64u < a.hi && is_hdr_byte(a.text[base + 64u])) cm |= 1ULL << 63;
    const bool owns_lo = a.lo < a.hi && a.lo >= base && a.lo < base + 64u;
    fc = owns_lo && ((h >> (a.lo - base)) & 1ULL) ? 1u : 0u;    // line 0
}

__global__ __launch_bounds__(256) void count_kernel(ParseArgs pa)
{
    __shared__ u64 lds4[4];
    const u32 s = blockIdx.y;
    const StreamArgs &a = pa.s[s];
    for (u32 t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        u64 m, cm; u32 fc;
        tile_masks(a, (a.tile0 + t) * TILE + threadIdx.x * 64u, m, cm, fc);
        u64 total;
        (void)block_excl_scan((u64)__popcll(m) | ((u64)((u32)__popcll(cm) + fc) << 31), total, lds4);
        if (threadIdx.x == 0) {
            a.st_lines[t] = total;
            if (total) atomicAdd((unsigned long long *)&a.st_groups[t >> 6], (unsigned long long)total);
        }
    }
}

__global__ __launch_bounds__(256) void write_kernel(ParseArgs pa)
{
    __shared__ u64 s_pre, lds4[4];
    const u32 s = blockIdx.y;
    const StreamArgs &a = pa.s[s];
    StreamInfo *si = &pa.ci->s[s];
    for (u32 t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const u32 lane = threadIdx.x, g = t >> 6;
            u64 c = lane < (t & 63u) ? a.st_lines[(t & ~63u) + lane] : 0ULL;
            for (u32 i = lane; i < g; i += 64u) c += a.st_groups[i];
#pragma unroll
            for (int off = 32; off; off >>= 1) c += shfl_xor64(c, off);
            if (lane == 0) s_pre = c;
        }
        const u32 base = (a.tile0 + t) * TILE + threadIdx.x * 64u;
        u64 m, cm; u32 fc;
        tile_masks(a, base, m, cm, fc);
        u64 total;
        const u64 ex = block_excl_scan((u64)__popcll(m) | ((u64)((u32)__popcll(cm) + fc) << 31), total, lds4);     // (its barriers publish s_pre)
        const u64 at = s_pre + ex;
        u32 li = (u32)(at & 0x7FFFFFFFu) + 1u;                   // the line behind this thread's first '\n'
        u32 cj = (u32)(at >> 31);
        if (fc) { if (cj < a.cap_rec) { a.cand[cj] = 0; a.c_inside[cj] = 0; } ++cj; }
        while (m) {
            const u32 b = (u32)__builtin_ctzll(m);
            if (li < a.cap_lines) a.ls[li] = base + b + 1u;
            if ((cm >> b) & 1ULL) { if (cj < a.cap_rec) { a.cand[cj] = li; a.c_inside[cj] = 0; } ++cj; }
            ++li;
            m &= m - 1;
        }
        if (t == a.n_tiles - 1 && threadIdx.x == 255) {          // (the last thread in text order: li, cj have run over everything)
            const u32 n_nl = li - 1u;
            a.ls[0] = a.lo;
            si->n_nl = n_nl; si->lo = a.lo; si->hi = a.hi;
            if (n_nl + 2u <= a.cap_lines && cj <= a.cap_rec) {
                a.ls[n_nl + 1u] = a.hi + 1u;
                si->n_lines = n_nl + 1u; si->n_cand = cj;
                const bool empty_last = a.hi == a.lo || a.text[a.hi - 1u] == '\n';
                si->n_real = pa.final_text ? n_nl + 1u - (empty_last ? 1u : 0u) : n_nl;
            } else { si->n_lines = 0; si->n_cand = 0; si->n_real = 0; si->why |= BNS_TEXT_WHY_LINES; }
        }
    }
}

__device__ __forceinline__ bool is_space(u8 c) { return c == ' ' || (c >= 9 && c <= 13); }
// bytes of [from, to) in front of the first isspace byte, four at a time (two aligned words; the buffers are readable past `to`)
__device__ __forceinline__ u32 name_length(const u8 *__restrict__ text, u32 from, u32 to)
{
    u32 n = 0;
    while (from + n < to) {
        const u32 pos = from + n, mis = pos & 3u;
        const u32 *ap = reinterpret_cast<const u32 *>(text + (pos - mis));
        const u32 w = (u32)((((u64)ap[1] << 32) | ap[0]) >> (8u * mis));
        const u32 left = to - pos;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) {
            if (i >= left || is_space((u8)(w >> (8u * i)))) return n + i;
        }
        n += 4;
    }
    return n;
}

// ---- one lane per candidate: the record kseq_read returns when it starts there ------------------------------------------------
__global__ __launch_bounds__(256) void walk_kernel(ParseArgs pa)
{
    const u32 s = blockIdx.y;
    const StreamArgs &a = pa.s[s];
    const StreamInfo &si = pa.ci->s[s];
    const u32 n = si.n_cand, n_lines = si.n_lines, n_real = si.n_real;
    const bool fin = pa.final_text != 0;
    const u8 *__restrict__ text = a.text;
    const u32 *__restrict__ ls = a.ls;
    u32 my_why = 0;
    for (u32 j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
        const u32 h = a.cand[j];
        const u32 hs = ls[h], he = ls[h + 1] - 1u;
        // name: the header line behind its first byte, up to the first isspace byte (klib/kseq.h:190)
        u32 nl = name_length(text, hs + 1u, he);
        if (pa.trim_readno && nl > 2 && text[hs + nl - 1u] == '/' && text[hs + nl] >= '0' && text[hs + nl] <= '9') nl -= 2;   // kseq_declare.h:106-110
        u32 total = 0, n_s = 0, first = 0xFFFFFFFFu, flags = 0;
        u32 next = fin ? n_lines : n_real;                      // (where the text ends: nothing behind this record)
        bool incomplete = false, plus = false;
        u32 l = h + 1u;
        // sequence lines, up to the line that starts with '>' / '@' (the next record) or '+' (klib/kseq.h:197-201)
        for (;; ++l) {
            if (l >= n_real) { incomplete = !fin; break; }      // (a final text ends the record here)
            if (l - h > MAX_REC_LINES) { flags |= BNS_TEXT_WHY_LONG_RECORD; break; }
            const u32 st = ls[l], e = ls[l + 1] - 1u, len = e - st;
            a.line_off[l] = total;
            if (!len) continue;
            const u8 c0 = text[st];
            if (is_hdr_byte(c0)) { next = l; break; }
            if (c0 == '+') { plus = true; break; }
            // (ks_getuntil2 :135: the appended line loses one trailing '\r' unless the sequence is then a single byte -- or the line is one
            // byte at the very end of the input, without a newline: ks_getc has taken that byte and ks_getuntil2 returns -1 in front of the test)
            const u32 eff = len - ((text[e - 1u] == '\r' && total + len > 1u && !(len == 1u && l == n_lines - 1u)) ? 1u : 0u);
            if (eff) { if (!n_s) first = st; ++n_s; }            // (a lone '\r' that was stripped: a blank line of CRLF text)
            total += eff;
        }
        const u32 line1 = l < n_lines ? l : n_lines;            // end of the sequence lines
        if (plus) {
            if (l == n_lines - 1u) flags |= BNS_TEXT_WHY_QUAL_LEN;                  // the input ends inside the '+' line (klib/kseq.h:215-216: error -2)
            else {
                // quality lines until they are as long as the sequence (klib/kseq.h:217): at least one line is read
                u32 q = 0;
                for (++l;;) {
                    if (l >= n_real) { incomplete = !fin; break; }
                    if (l - h > MAX_REC_LINES) { flags |= BNS_TEXT_WHY_LONG_RECORD; break; }
                    const u32 st = ls[l], e = ls[l + 1] - 1u, len = e - st;
                    q += len - ((len && text[e - 1u] == '\r' && q + len > 1u) ? 1u : 0u);
                    ++l;
                    if (q >= total) break;
                }
                if (!incomplete && !flags && q != total) flags |= BNS_TEXT_WHY_QUAL_LEN;
                // blank lines, then the next record's header (klib/kseq.h:183-186 skips to the next '>' / '@' BYTE: text in between is not regular)
                if (!incomplete && !flags)
                    for (;; ++l) {
                        if (l >= n_real) break;
                        if (l - h > MAX_REC_LINES) { flags |= BNS_TEXT_WHY_LONG_RECORD; break; }
                        const u32 st = ls[l], len = ls[l + 1] - 1u - st;
                        if (!len || (len == 1u && text[st] == '\r')) continue;                // (a blank line, LF or CRLF)
                        if (!is_hdr_byte(text[st])) flags |= BNS_TEXT_WHY_AFTER_QUAL;
                        next = l;
                        break;
                    }
            }
        }
        a.c_next[j] = next; a.c_seq[j] = total; a.c_name[j] = nl; a.c_line1[j] = line1;
        a.c_single[j] = n_s == 1 ? first : 0xFFFFFFFFu;
        a.c_flags[j] = flags | (incomplete ? WALK_INCOMPLETE : 0u);
        a.c_pos[j] = hs;
        my_why |= flags;
        // the candidates inside this record are not headers IF this one is (compact_kernel decides)
        const u32 reach = incomplete ? 0xFFFFFFFFu : (flags ? l : next);
        bool any = false;
        for (u32 jj = j + 1; jj < n && a.cand[jj] < reach; ++jj) { a.c_inside[jj] = 1u; any = true; }
        if (any) pa.ci->s[s].any_inside = 1u;
    }
    // (what is wrong with ANY candidate's record: the stream's why when every candidate turns out to be a header)
#pragma unroll
    for (int off = 32; off; off >>= 1) my_why |= (u32)__shfl_xor((int)my_why, off);
    if ((threadIdx.x & 63u) == 0 && my_why) atomicOr(&pa.ci->s[s].why_all, my_why);
}

// is candidate j a header?  Unmarked: yes.  Marked: follow the records from the nearest unmarked candidate in front.
__device__ __forceinline__ bool is_header(const StreamArgs &a, u32 j)
{
    if (!a.c_inside[j]) return true;
    u32 t = j;
    do { --t; } while (a.c_inside[t]);                          // (candidate 0 is never marked)
    const u32 target = a.cand[j];
    for (;;) {
        if (a.c_flags[t]) return false;                        // (a record that is cut short or not regular: what lies behind it is nobody's yet)
        const u32 nx = a.c_next[t];
        if (nx == target) return true;
        if (nx > target) return false;
        u32 lo = t + 1, hi = j;                                 // the candidate at line nx
        while (lo < hi) { const u32 m = (lo + hi) >> 1; if (a.cand[m] < nx) lo = m + 1; else hi = m; }
        if (lo >= j || a.cand[lo] != nx) return false;
        t = lo;
    }
}

// the candidate that is record r's header
__device__ __forceinline__ u32 hdr_of(const StreamArgs &a, bool fast, u32 r) { return fast ? r : a.rec_cand[r]; }

// How many records this call takes.  Per stream: the headers in front of `limit` (stream 0), of which the last one is only
// complete when the text is final or another header follows; a pair of files takes the minimum.  consumed = where the first
// record not taken starts.  One thread, behind the streams' n_hdr and why.
__device__ void decide(const ParseArgs &pa, u32 limit, bool fast)
{
    CallInfo *ci = pa.ci;
    const int final_text = pa.final_text;
    u32 take = 0xFFFFFFFFu, why = 0;
    for (u32 s = 0; s < pa.n_streams; ++s) {
        StreamInfo &si = ci->s[s];
        const StreamArgs &a = pa.s[s];
        si.n_take = 0; si.n_eff = 0;
        if (si.why) { why |= si.why; take = 0; continue; }
        // text in front of the first header: blank lines only (kseq skips to the next '>' / '@' byte wherever it stands)
        const u32 lead = si.n_hdr ? a.cand[hdr_of(a, fast, 0)] : si.n_real;
        for (u32 i = 0; i < lead; ++i) {
            const u32 len = a.ls[i + 1] - 1u - a.ls[i];
            if (len && !(len == 1u && a.text[a.ls[i]] == '\r')) { why |= BNS_TEXT_WHY_LEADING; break; }
            if (i >= MAX_REC_LINES) { why |= BNS_TEXT_WHY_LONG_RECORD; break; }
        }
        // klib/kseq.h:189: a header byte with NOTHING behind it (the last byte of the input) ends the stream without a record
        u32 n_eff = si.n_hdr;
        if (final_text && n_eff && a.c_pos[hdr_of(a, fast, n_eff - 1)] + 1u == si.hi) --n_eff;
        si.n_eff = n_eff;
        u32 t = n_eff;
        if (s == 0 && limit < si.hi) {                          // headers that start in front of the limit
            u32 x = 0, y = n_eff;
            while (x < y) { const u32 m = (x + y) >> 1; if (a.c_pos[hdr_of(a, fast, m)] < limit) x = m + 1; else y = m; }
            t = x;
        }
        if (t == n_eff && !final_text && t) --t;                // the last header's record ends where the next text begins
        si.n_take = t;
        take = take < t ? take : t;
    }
    if (why) take = 0;
    ci->n_take = take; ci->n_reads = take * pa.n_streams; ci->why = why; ci->fast = fast ? 1u : 0u;
    for (u32 s = 0; s < pa.n_streams; ++s) {
        StreamInfo &si = ci->s[s];
        const StreamArgs &a = pa.s[s];
        // the first record not taken; with every header taken (a final text) the end of the text; nothing there yet: the start
        if (why) si.consumed = si.lo;
        else if (take < si.n_eff) si.consumed = a.c_pos[hdr_of(a, fast, take)];
        else si.consumed = final_text ? si.hi : si.lo;
    }
}

// candidates -> records (their headers in order), then decide().  As a rule no walk has found a candidate inside its record: every
// candidate is a header, record r is candidate r, and one thread decides at once.  Otherwise the headers are numbered (look-back over
// blocks of candidates) and the block that finishes last decides.
__global__ __launch_bounds__(256) void compact_kernel(ParseArgs pa, u32 limit)
{
    __shared__ u32 s_b;
    __shared__ u64 s_pre, lds4[4];
    CallInfo *ci = pa.ci;
    const u32 s = blockIdx.y;
    const StreamArgs &a = pa.s[s];
    StreamInfo *si = &ci->s[s];
    const bool fast = !(ci->s[0].any_inside | (pa.n_streams == 2 ? ci->s[1].any_inside : 0u));
    if (fast) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
            for (u32 q = 0; q < pa.n_streams; ++q) { ci->s[q].n_hdr = ci->s[q].n_cand; ci->s[q].why |= ci->s[q].why_all; }
            decide(pa, limit, true);
        }
        return;
    }
    const u32 n = si->n_cand;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_b = atomicAdd(&ci->ticket[2 + s], 1u);
        __syncthreads();
        const u32 b = s_b;
        if ((u64)b * REC_BLOCK >= n) break;
        const u32 j0 = b * REC_BLOCK + threadIdx.x * REC_ITEMS;
        u32 hd[REC_ITEMS], v = 0, why = 0;
#pragma unroll
        for (u32 i = 0; i < REC_ITEMS; ++i) {
            hd[i] = j0 + i < n && is_header(a, j0 + i) ? 1u : 0u;
            v += hd[i];
            if (hd[i]) why |= a.c_flags[j0 + i] & ~WALK_INCOMPLETE;
        }
        u64 total;
        const u64 ex = block_excl_scan(v, total, lds4);
        u32 r = (u32)(lookback(a.st_compact, b, total, &s_pre) + ex);
#pragma unroll
        for (u32 i = 0; i < REC_ITEMS; ++i) if (hd[i]) { a.rec_cand[r] = j0 + i; ++r; }      // (r < n_cand <= cap_rec)
        if (why) atomicOr(&si->why, why);
        if (b == (n - 1u) / REC_BLOCK && threadIdx.x == 255) si->n_hdr = r;
    }
    // the last block to get here (either stream's) decides
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&ci->ticket[5], 1u) == gridDim.x * gridDim.y - 1u) { __threadfence(); decide(pa, limit, false); }
    }
}

// records taken (mates interleaved: record r of stream s at R = r * n_streams + s) -> sequence lengths, base offsets (from off_base
// on), name offsets (from name_base on): the records go behind those the open batch holds already
struct OffsetsOut { u32 *seq_len; u64 *offsets; u64 off_base; u32 *name_off; u32 name_base; };
__global__ __launch_bounds__(256) void offsets_kernel(ParseArgs pa, OffsetsOut o)
{
    __shared__ u32 s_b;
    __shared__ u64 s_pre, lds4[4];
    CallInfo *ci = pa.ci;
    const u32 ns = pa.n_streams, n = ci->n_reads;
    const bool fast = ci->fast != 0;
    if (!n) { if (blockIdx.x == 0 && threadIdx.x == 0) { o.offsets[0] = o.off_base; o.name_off[0] = o.name_base; } return; }
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_b = atomicAdd(&ci->ticket[4], 1u);
        __syncthreads();
        const u32 b = s_b;
        if ((u64)b * REC_BLOCK >= n) return;
        const u32 R0 = b * REC_BLOCK + threadIdx.x * REC_ITEMS;
        u32 sl[REC_ITEMS], nm[REC_ITEMS], mx = 0;
        u64 v = 0;
#pragma unroll
        for (u32 i = 0; i < REC_ITEMS; ++i) {
            const u32 R = R0 + i;
            sl[i] = nm[i] = 0;
            if (R < n) {
                const StreamArgs &a = pa.s[ns == 2 ? (R & 1u) : 0u];
                const u32 j = hdr_of(a, fast, ns == 2 ? (R >> 1) : R);
                sl[i] = a.c_seq[j]; nm[i] = a.c_name[j];
            }
            v += (u64)sl[i] | ((u64)nm[i] << 31);
            mx = mx > sl[i] ? mx : sl[i];
        }
        u64 total;
        const u64 ex = block_excl_scan(v, total, lds4);
        u64 pre = lookback(pa.st_offsets, b, total, &s_pre) + ex;
#pragma unroll
        for (u32 i = 0; i < REC_ITEMS; ++i) {
            const u32 R = R0 + i;
            if (R < n) { o.seq_len[R] = sl[i]; o.offsets[R] = o.off_base + (pre & 0x7FFFFFFFULL); o.name_off[R] = o.name_base + (u32)(pre >> 31); }
            pre += (u64)sl[i] | ((u64)nm[i] << 31);
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) { const u32 x = (u32)__shfl_xor((int)mx, off); mx = mx > x ? mx : x; }
        if ((threadIdx.x & 63u) == 0 && mx) atomicMax(&ci->max_len, mx);
        if (b == (n - 1u) / REC_BLOCK && threadIdx.x == 255) {
            const u32 tb = (u32)(pre & 0x7FFFFFFFULL), tn = (u32)(pre >> 31);
            o.offsets[n] = o.off_base + tb; o.name_off[n] = o.name_base + tn;
            ci->total_bases = tb; ci->names_bytes = tn;
        }
    }
}

// ---- pack: SUB lanes per record (four records per wavefront), 64 bases a pass (4 per lane), the word layout of pack_kernel; the
// record's name and position.  (offsets start at the slice's first record, which is record R0 of the batch's packed image: word base
// (offset >> 5) + index)
struct PackOut { const u64 *offsets; u32 R0; u64 *words; u32 *nmask; const u32 *name_off; u32 name_base; char *names; u64 *pos64; u32 rel[2]; };
__global__ __launch_bounds__(256) void pack_text_kernel(ParseArgs pa, PackOut o)
{
    const u32 sl = threadIdx.x & (SUB - 1u);                    // lane within the record's group
    const CallInfo *ci = pa.ci;
    const u32 n = ci->n_reads, ns = pa.n_streams;
    const bool fast = ci->fast != 0;
    if (ci->why) return;
    constexpr u32 PER_BLOCK = 256u / SUB;
    const u32 n_groups = gridDim.x * PER_BLOCK;
    // (the wavefront's four groups run the same number of passes: shuffles below are wave-wide)
    for (u32 Rw = blockIdx.x * PER_BLOCK + (threadIdx.x >> 6) * (64u / SUB); Rw < n; Rw += n_groups) {
        const u32 R = Rw + ((threadIdx.x & 63u) / SUB);
        const bool live = R < n;
        const u32 s = ns == 2 ? (R & 1u) : 0u;
        const StreamArgs &a = pa.s[s];
        u32 j = 0, L = 0, single = 0xFFFFFFFFu, hs = 0, l0 = 0, l1 = 0, nlen = 0;
        u64 wb = 0;
        u32 noff = 0;
        if (live) {
            j = hdr_of(a, fast, ns == 2 ? (R >> 1) : R);
            L = a.c_seq[j]; single = a.c_single[j]; hs = a.c_pos[j]; nlen = a.c_name[j];
            wb = (o.offsets[R] >> 5) + o.R0 + R;
            noff = o.name_off[R];
            if (single == 0xFFFFFFFFu && L) { l0 = a.cand[j] + 1u; l1 = a.c_line1[j]; }
        }
        const u32 n_words = (L + 31u) >> 5;
        // the name (klib/kseq.h:190; trimmed in walk_kernel) and where the record starts in the caller's text
        if (live) {
            char *dst = o.names + (noff - o.name_base);
            for (u32 i = sl; i < nlen; i += SUB) dst[i] = (char)a.text[hs + 1u + i];
            if (sl == 0) o.pos64[R] = hs - o.rel[s];
        }
        u32 passes = (n_words + 1u) >> 1;                       // 64 bases = two words a pass
#pragma unroll
        for (int off = 32; off >= (int)SUB; off >>= 1) { const u32 x = (u32)__shfl_xor((int)passes, off); passes = passes > x ? passes : x; }
        for (u32 ps = 0; ps < passes; ++ps) {
            const u32 p = ps * 64u;
            const u32 bi = p + sl * 4u;
            u32 w = 0;                                          // up to four bytes of sequence, first base in the low byte
            if (bi < L) {
                const u32 nb = L - bi < 4u ? L - bi : 4u;
                if (single != 0xFFFFFFFFu) {
                    const u32 addr = single + bi, mis = addr & 3u;
                    const u32 *ap = reinterpret_cast<const u32 *>(a.text + (addr - mis));
                    const u32 lo = ap[0], hi = (mis + nb > 4u) ? ap[1] : 0u;
                    w = (u32)((((u64)hi << 32) | lo) >> (8u * mis));
                } else {
                    // the line that holds base bi: the last sequence line of the record whose offset is <= bi (blank lines -- and a line
                    // that was nothing but a stripped '\r' -- share the offset of the line behind them and come first)
                    u32 x = l0, y = l1;
                    while (y - x > 1u) { const u32 m = (x + y) >> 1; if (a.line_off[m] <= bi) x = m; else y = m; }
                    u32 l = x, at = bi - a.line_off[l], st = a.ls[l];
                    u32 len = (l + 1u < l1 ? a.line_off[l + 1u] : L) - a.line_off[l];        // (the line's bases: without the '\r' ks_getuntil2 strips)
                    for (u32 i = 0; i < nb; ++i) {
                        while (at >= len) { ++l; at = 0; st = a.ls[l]; len = (l + 1u < l1 ? a.line_off[l + 1u] : L) - a.line_off[l]; }   // (bi + i < L: a line with bases follows)
                        w |= (u32)a.text[st + at] << (8u * i);
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
            const u32 g = sl & 7u;
            u32 hi32 = g < 4 ? codes << (24 - 8 * g) : 0u;
            u32 lo32 = g >= 4 ? codes << (24 - 8 * (g - 4)) : 0u;
            u32 nm = bads << (28 - 4 * g);
            hi32 |= dpp<QP_XOR1>(hi32); lo32 |= dpp<QP_XOR1>(lo32); nm |= dpp<QP_XOR1>(nm);
            hi32 |= dpp<QP_XOR2>(hi32); lo32 |= dpp<QP_XOR2>(lo32); nm |= dpp<QP_XOR2>(nm);
            hi32 |= (u32)__shfl_xor((int)hi32, 4); lo32 |= (u32)__shfl_xor((int)lo32, 4); nm |= (u32)__shfl_xor((int)nm, 4);
            const u32 wi = (p >> 5) + (sl >> 3);
            if (g == 0 && wi < n_words) { o.words[wb + wi] = ((u64)hi32 << 32) | lo32; o.nmask[wb + wi] = nm; }
        }
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
    DevBuf ls[2], line_off[2], cand[2][10], info, offsets, words, nmask, hits;   // cand[s]: cand, c_next, c_seq, c_name, c_line1, c_single, c_flags, c_inside, c_pos, rec_cand (StreamArgs)
    // what goes back to the host, TWICE: batch b's results are copied (on the back stream) while batch b + 1 is parsed and classified into the other set
    DevBuf seq_len[2], name_off[2], names[2], pos64[2], out[2][4], runs[2][4];
    hipEvent_t ev_done[2] = {}, tc0[2] = {}, tc1[2] = {};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    CallInfo *h_info = nullptr;                         // page-locked
    unsigned long long *h_cursor = nullptr;
    struct TextCall *call = nullptr;                    // the state of the call in progress -- or, behind BNS_TEXT_DEFER, of the one that waits for bns_text_finish
};

// One bns_classify_text call: what it has accepted so far, the open batch, the state in front of the last classified batch.
struct TextCall {
    bool pending = false;                               // BNS_TEXT_DEFER: parsed and packed, the open batch waits for bns_text_finish
    bns_text_out out{};
    bool parse_only = false, want_runs = false, on_device = false;
    u32 ns = 1;
    u32 batch_no = 0;                                   // batches classified so far
    u64 runs_done = 0;                                  // runs whose copy to the host has been queued
    bool runs_overflow = false;                         // the caller's run arrays are full: the batch whose runs did not fit (and what follows) is not his
    bool rolled_back = false;
    // accepted so far (slices parsed and counted, classified or waiting in the open batch) / the open batch / where the open batch began /
    // the state in front of the last classified batch
    u64 done_reads = 0, names_done = 0, bases_done = 0;
    u32 cons[2] = {0, 0};
    u64 acc_reads = 0, acc_bases = 0, acc_names = 0; u32 acc_max_len = 0;
    u64 open_reads0 = 0, open_names0 = 0, open_bases0 = 0; u32 open_cons0[2] = {0, 0};
    u64 reads_before_prev = 0, names_before_prev = 0, bases_before_prev = 0; u32 cons_before_prev[2] = {0, 0};
    int status = BNS_TEXT_OK;
    u32 why = 0, n_launches = 0, rounds = 0;
    float ms_parse = 0, ms_classify = 0;
    u32 rel[2] = {0, 0};
    Upload *up[2] = {nullptr, nullptr};
};

void text_release_uploads(TextCall &tc) { for (u32 s = 0; s < tc.ns; ++s) if (tc.up[s]) { tc.up[s]->pending = false; tc.up[s] = nullptr; } }
// an error exit: uploads in flight are waited for and given up, the three streams drained
int text_bail(bns_ctx *ctx, TextCall &tc, int code)
{
    if (!tc.on_device && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->back_stream) (void)hipStreamSynchronize(ctx->back_stream);
    text_release_uploads(tc);
    tc.pending = false;
    return code;
}

// (after a drain of the back stream) the runs of batch batch_no - 1 -> the caller's arrays, or the context's
int text_flush_runs_of_prev(bns_ctx *ctx, TextWork &tw, TextCall &tc)
{
    if (!tc.want_runs || tc.batch_no == 0) return BNS_OK;
    hipStream_t bs = ctx->back_stream;
    const bns_text_out *out = &tc.out;
    u64 &runs_done = tc.runs_done;
    const u32 pq = (tc.batch_no - 1u) & 1u;
    const u64 n_tot = tw.h_cursor[pq];                      // runs so far, that batch's included
    if (out->run_tax) {                                     // the caller's own arrays
        if (n_tot > out->runs_cap) { tc.runs_overflow = true; return BNS_OK; }
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
}

// the last classified batch is not the caller's after all (its runs did not fit his arrays): nor is what came behind it
void text_roll_back(TextCall &tc)
{
    if (tc.rolled_back) return;
    tc.rolled_back = true;
    tc.done_reads = tc.reads_before_prev; tc.names_done = tc.names_before_prev; tc.bases_done = tc.bases_before_prev;
    for (u32 s = 0; s < tc.ns; ++s) tc.cons[s] = tc.cons_before_prev[s];
    tc.acc_reads = tc.acc_bases = tc.acc_names = 0; tc.acc_max_len = 0;
}

// ---- the open batch -> one classify launch; its results behind those of the batches in front.  Results leave on the back stream:
// batch b's arrays (set b & 1) are copied out while batch b + 1 is parsed and classified into the other set.  Before a batch is
// classified the back stream is drained -- nothing of it then reads the set that batch writes (its last user was two batches ago, and
// its parse started behind the drain of the batch in between) -- and the run count of the batch in front is known: its runs are copied now.
int text_flush_batch(bns_ctx *ctx, TextWork &tw, TextCall &tc)
{
    if (tc.rolled_back || tc.acc_reads == 0) return BNS_OK;
    hipStream_t st = ctx->stream, bs = ctx->back_stream;
    const bns_text_out *out = &tc.out;
    const bool parse_only = tc.parse_only, want_runs = tc.want_runs;
    const u32 ns = tc.ns, q = tc.batch_no & 1u;
    const u64 n_reads = tc.acc_reads, n_units = tc.acc_reads / ns, u_done = tc.open_reads0 / ns;
    // (the back stream drained: the out / runs arrays of set q are free, and the run count of the batch in front is on the host)
    HIPCHK(ctx, hipStreamSynchronize(bs));
    if (tc.batch_no && ctx->timing && !parse_only) { float ms = 0; if (hipEventElapsedTime(&ms, tw.tc0[q ^ 1u], tw.tc1[q ^ 1u]) == hipSuccess) tc.ms_classify += ms; }
    int frc = text_flush_runs_of_prev(ctx, tw, tc);
    if (frc != BNS_OK) return frc;
    if (tc.runs_overflow) { text_roll_back(tc); tc.status = BNS_TEXT_CAP; return BNS_OK; }
    u32 *o0 = (u32 *)tw.out[q][0].p, *o1 = (u32 *)tw.out[q][1].p, *o2 = (u32 *)tw.out[q][2].p, *o3 = (u32 *)tw.out[q][3].p;
    unsigned long long *d_cur = &((SmallLayout *)ctx->small.p)->runs_cursor;     // (zeroed at the start of the call: it runs on over the batches)
    if (!parse_only) {
        if (ctx->timing) HIPCHK(ctx, hipEventRecord(tw.tc0[q], st));
        frc = classify_device_impl(ctx, nullptr, (const u64 *)tw.words.p, (const u32 *)tw.nmask.p, (const u64 *)tw.offsets.p, n_reads, tc.acc_bases,
                                   std::max<u32>(tc.acc_max_len, 1u), ns == 2 ? 1 : 0, o0, out->missing || want_runs ? o1 : nullptr,
                                   out->ambig || want_runs ? o2 : nullptr, (out->n_hits || want_runs) ? o3 : nullptr, want_runs ? (u32 *)tw.hits.p : nullptr, st);
        if (frc != BNS_OK) return frc;
        ++tc.n_launches;
        if (ctx->timing) HIPCHK(ctx, hipEventRecord(tw.tc1[q], st));
        if (want_runs) {
            hipLaunchKernelGGL(hit_runs_kernel, dim3(grid_for(ctx, (n_units + HIT_RUNS_GROUP - 1) / HIT_RUNS_GROUP, 4)), dim3(256), 0, st, (const u32 *)tw.hits.p,
                               (const u64 *)tw.offsets.p, ns, (const u32 *)o3, (u64)n_units, (u64 *)tw.runs[q][0].p, (u32 *)tw.runs[q][1].p,
                               (u32 *)tw.runs[q][2].p - tc.runs_done, (u32 *)tw.runs[q][3].p - tc.runs_done, d_cur);
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
    if (out->seq_len) HIPCHK(ctx, hipMemcpyAsync(out->seq_len + tc.open_reads0, tw.seq_len[q].p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, bs));
    if (out->rec_pos) HIPCHK(ctx, hipMemcpyAsync(out->rec_pos + tc.open_reads0, tw.pos64[q].p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, bs));
    if (out->name_off) {
        HIPCHK(ctx, hipMemcpyAsync(out->name_off + tc.open_reads0, tw.name_off[q].p, (size_t)(n_reads + 1) * 4, hipMemcpyDeviceToHost, bs));
        if (tc.acc_names) HIPCHK(ctx, hipMemcpyAsync(out->names + tc.open_names0, tw.names[q].p, (size_t)tc.acc_names, hipMemcpyDeviceToHost, bs));
    }
    if (out->words) HIPCHK(ctx, hipMemcpyAsync(out->words, tw.words.p, (size_t)bns_packed_words(tc.acc_bases, n_reads) * 8, hipMemcpyDeviceToHost, bs));
    if (out->nmask) HIPCHK(ctx, hipMemcpyAsync(out->nmask, tw.nmask.p, (size_t)bns_packed_words(tc.acc_bases, n_reads) * 4, hipMemcpyDeviceToHost, bs));
    ++tc.batch_no;
    tc.reads_before_prev = tc.open_reads0; tc.names_before_prev = tc.open_names0; tc.bases_before_prev = tc.open_bases0;
    for (u32 s = 0; s < ns; ++s) tc.cons_before_prev[s] = tc.open_cons0[s];
    tc.acc_reads = tc.acc_bases = tc.acc_names = 0; tc.acc_max_len = 0;
    return BNS_OK;
}

// the caller's host buffers are his again -- those of THIS call: an upload prefetched for the next one keeps travelling
int text_give_back_uploads(bns_ctx *ctx, TextWork &tw, TextCall &tc)
{
    if (!tc.on_device && ctx->copy_stream) {
        bool other_pending = false;
        for (u32 s = 0; s < tc.ns; ++s) for (Upload &c : tw.up[s]) if (c.pending && &c != tc.up[s]) other_pending = true;
        if (!other_pending) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
        else for (u32 s = 0; s < tc.ns; ++s) if (tc.up[s]) HIPCHK(ctx, hipEventSynchronize(tc.up[s]->ev[tc.up[s]->n_pieces - 1]));
    }
    text_release_uploads(tc);
    return BNS_OK;
}

void text_fill_info(bns_ctx *ctx, const TextCall &tc, bns_text_info *info)
{
    info->n_records = tc.done_reads;
    for (u32 s = 0; s < tc.ns; ++s) info->consumed[s] = tc.cons[s] - tc.rel[s];
    info->total_bases = tc.bases_done; info->names_bytes = tc.names_done; info->n_runs_total = tc.runs_done;
    info->run_tax = tc.out.run_tax ? tc.out.run_tax : ctx->h_run_tax; info->run_len = tc.out.run_len ? tc.out.run_len : ctx->h_run_len;
    info->status = tc.status; info->why = tc.why; info->n_slices = tc.rounds + 1; info->n_launches = tc.n_launches;
    info->ms_parse = tc.ms_parse; info->ms_classify = tc.ms_classify;
}

// The end of a call (or of bns_text_finish): the open batch (whatever ended the rounds, the records in front of that are the caller's);
// then what is still on its way -- the last batch's arrays, then its runs.
int text_wrap_up(bns_ctx *ctx, TextWork &tw, TextCall &tc, bns_text_info *info)
{
    hipStream_t st = ctx->stream, bs = ctx->back_stream;
    int rc = text_flush_batch(ctx, tw, tc);
    if (rc != BNS_OK) return text_bail(ctx, tc, rc);
#define WUCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ctx->err = std::string(#expr) + ": " + hipGetErrorString(_e); return text_bail(ctx, tc, _e == hipErrorOutOfMemory ? BNS_ERR_NOMEM : BNS_ERR_HIP); } } while (0)
    WUCHK(hipStreamSynchronize(st));
    WUCHK(hipStreamSynchronize(bs));
    if (tc.batch_no && ctx->timing && !tc.parse_only) { float ms = 0; if (hipEventElapsedTime(&ms, tw.tc0[(tc.batch_no - 1u) & 1u], tw.tc1[(tc.batch_no - 1u) & 1u]) == hipSuccess) tc.ms_classify += ms; }
    if (!tc.rolled_back) {
        if ((rc = text_flush_runs_of_prev(ctx, tw, tc)) != BNS_OK) return text_bail(ctx, tc, rc);
        if (tc.runs_overflow) { text_roll_back(tc); tc.status = BNS_TEXT_CAP; }
    }
    WUCHK(hipStreamSynchronize(bs));
#undef WUCHK
    if ((rc = text_give_back_uploads(ctx, tw, tc)) != BNS_OK) return text_bail(ctx, tc, rc);
    tc.pending = false;
    text_fill_info(ctx, tc, info);
    return BNS_OK;
}

}  // namespace

struct bns_text_work : TextWork {};

void text_work_free(bns_ctx *ctx)
{
    bns_text_work *tw = ctx->text_work;
    if (!tw) return;
    std::vector<DevBuf *> bufs = {&tw->up[0][0].buf, &tw->up[0][1].buf, &tw->up[1][0].buf, &tw->up[1][1].buf, &tw->ls[0], &tw->ls[1],
                                  &tw->line_off[0], &tw->line_off[1], &tw->info, &tw->offsets, &tw->words, &tw->nmask, &tw->hits};
    for (auto &row : tw->cand) for (DevBuf &b : row) bufs.push_back(&b);
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
    delete tw->call;
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

int bns_host_register(bns_ctx *ctx, void *p, size_t bytes)
{
    if (!ctx || !p || !bytes) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(p, bytes, hipHostRegisterPortable));
    return BNS_OK;
}
int bns_host_unregister(bns_ctx *ctx, void *p)
{
    if (!ctx || !p) return BNS_ERR_ARG;
    HIPCHK(ctx, hipHostUnregister(p));
    return BNS_OK;
}

int bns_dev_copy_peer(bns_ctx *dst_ctx, void *dst, bns_ctx *src_ctx, const void *src, size_t bytes)
{
    if (!dst_ctx || !src_ctx || (bytes && (!dst || !src))) return BNS_ERR_ARG;
    if (!bytes) return BNS_OK;
    static const bool via_host = std::getenv("BNS_PEER_VIA_HOST") != nullptr;       // (tests: the two-device path between contexts of one device)
    if (dst_ctx->device == src_ctx->device && !via_host) return bns_dev_copy(dst_ctx, dst, src, bytes);
    // Two devices: through page-locked host memory of the receiving context (a few hundred bytes as a rule -- the record that straddles
    // two blocks; peer access between the devices is not assumed).  The null stream of the sending device: no stream of the sending
    // CONTEXT is waited for -- the bytes were written long ago, and its classify launch is exactly what this copy must not wait behind.
    bns_ctx *ctx = dst_ctx;
    if (ctx->peer_stage_cap < bytes) {
        if (ctx->peer_stage) { HIPCHK(ctx, hipHostFree(ctx->peer_stage)); ctx->peer_stage = nullptr; ctx->peer_stage_cap = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
        HIPCHK(ctx, hipHostMalloc(&ctx->peer_stage, want, hipHostMallocPortable));
        ctx->peer_stage_cap = want;
    }
    HIPCHK(ctx, hipSetDevice(src_ctx->device));
    HIPCHK(ctx, hipMemcpy(ctx->peer_stage, src, bytes, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(dst, ctx->peer_stage, bytes, hipMemcpyHostToDevice, ctx->stream));
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
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = now_s();
    const bool parse_only = (flags & BNS_TEXT_PARSE_ONLY) != 0, on_device = (flags & BNS_TEXT_DEVICE) != 0;
    const bool final_text = (flags & BNS_TEXT_FINAL) != 0;
    if ((flags & BNS_TEXT_DEFER) && (out->words || out->nmask)) return BNS_ERR_ARG;
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
    if (!tw.call) tw.call = new (std::nothrow) TextCall();
    if (!tw.call) return BNS_ERR_NOMEM;
    TextCall &tc = *tw.call;
    if (tc.pending) return fail(ctx, BNS_ERR_STATE, "bns_classify_text: a deferred call waits for bns_text_finish");
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
            // -- or the text is large: a slice's workspaces are sized by its bytes (4.6 bytes per byte of text), and a call of 1.35 GB -- what one
            // call on a gzip stream inflates -- allocated 6 GB of them, again whenever a call was a little larger than every one before it (a
            // hipFree each, the device drained each time: one call in eight took 4 s).  Slices of 128 MiB beyond that: the same 1.2 GB for every call
            const bool pieces_forced = (ctx->dbg & BNS_DBG_SLICE_8K) || std::getenv("BNS_TEXT_PIECE_MB");
            const u64 big_piece = 128ull << 20;
            q.piece = pieces_forced ? text_piece_bytes(ctx, text_bytes[s]) : (q.end > big_piece ? big_piece : std::max<u64>(64, ((u64)q.end + 63) & ~63ULL));
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
    auto bail = [&](int code) { for (u32 s = 0; s < ns; ++s) tc.up[s] = src[s].up; tc.on_device = on_device; tc.ns = ns; return text_bail(ctx, tc, code); };
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
    // (by the call's text rounded UP to a power of two: the calls of one input differ by a few per cent -- batches of a BGZF file, the first
    // and the later calls of a gzip stream -- and a workspace that grows by a hair is freed and allocated again, every one of them, a hipFree
    // each: with another thread's kernels queueing up all the while one such call was seen to take 4 s instead of 20 ms)
    u64 sized_for = 64u << 20;
    while (sized_for < call_text) sized_for <<= 1;
    const u64 carry_reads = n_slices == 1 ? 0 : std::min<u64>(batch_reads, sized_for / 64 + 64);
    const u64 carry_bases = n_slices == 1 ? 0 : std::min<u64>(batch_bases, sized_for);
    const u64 carry_names = n_slices == 1 ? 0 : std::min<u64>(batch_names, sized_for);
    const u64 cap_reads = slice_reads_cap + carry_reads, cap_bases = slice_bases_cap + carry_bases, cap_names = slice_bases_cap + carry_names;
    if (cap_reads >= (1ULL << 31) || cap_bases >= (1ULL << 32) - (1ULL << 20)) return bail(fail(ctx, BNS_ERR_ARG, "bns_classify_text: text too large for one batch"));
    for (u32 s = 0; s < ns; ++s) {
        if ((rc = ensure(ctx, tw.ls[s], (size_t)cap_lines * 4 + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.line_off[s], (size_t)cap_lines * 4 + 64)) != BNS_OK) return bail(rc);
        for (DevBuf &b : tw.cand[s]) if ((rc = ensure(ctx, b, (size_t)cap_rec * 4 + 64)) != BNS_OK) return bail(rc);
    }
    // CallInfo and, behind it, the look-back words of the slice's scans (tiles of either stream, candidate blocks of either stream, record
    // blocks): zeroed together in front of every parse
    const size_t st_tiles = (size_t)(range_cap / TILE + 8), st_cand = (size_t)cap_rec / REC_BLOCK + 2, st_recs = (size_t)slice_reads_cap / REC_BLOCK + 2;
    const size_t st_groups = st_tiles / 64 + 2;
    const size_t info_bytes = ((sizeof(CallInfo) + 63) & ~size_t(63)) + (2 * st_tiles + 2 * st_groups + 2 * st_cand + st_recs) * 8;
    for (int q = 0; q < 2; ++q) {
        if ((rc = ensure(ctx, tw.seq_len[q], (size_t)cap_reads * 4 + 64)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.name_off[q], (size_t)(cap_reads + 1) * 4)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.pos64[q], (size_t)cap_reads * 8)) != BNS_OK) return bail(rc);
        if ((rc = ensure(ctx, tw.names[q], (size_t)cap_names + 64)) != BNS_OK) return bail(rc);
    }
    if ((rc = ensure(ctx, tw.offsets, (size_t)(cap_reads + 1) * 8)) != BNS_OK) return bail(rc);
    if ((rc = ensure(ctx, tw.words, ((size_t)cap_bases / 32 + cap_reads + 2) * 8)) != BNS_OK) return bail(rc);
    if ((rc = ensure(ctx, tw.nmask, ((size_t)cap_bases / 32 + cap_reads + 2) * 4)) != BNS_OK) return bail(rc);
    if ((rc = ensure(ctx, tw.info, info_bytes)) != BNS_OK) return bail(rc);
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
    const double t_alloc0 = now_s();
    TXCHK(hipStreamSynchronize(st));                            // (workspaces of an earlier call on this stream are free now)
    if (getenv("BNS_CLI_TIMING") && now_s() - t_enter > 0.2)
        fprintf(stderr, "[timing] bns_classify_text: %.3f s until its workspaces stood (%.3f of it draining the stream)\n", now_s() - t_enter, now_s() - t_alloc0);
    const u8 *d_text[2] = {src[0].base, src[1].base};

    if (want_runs) TXCHK(hipMemsetAsync(&((SmallLayout *)ctx->small.p)->runs_cursor, 0, 8, st));
    CallInfo *d_ci = (CallInfo *)tw.info.p;
    if (!ctx->back_stream) TXCHK(hipStreamCreateWithFlags(&ctx->back_stream, hipStreamNonBlocking));
    // ---- the call's state (struct TextCall: it outlives the call behind BNS_TEXT_DEFER), under the names the rounds below use
    tc = TextCall{};
    tc.out = *out; tc.parse_only = parse_only; tc.want_runs = want_runs; tc.on_device = on_device; tc.ns = ns;
    for (u32 s = 0; s < ns; ++s) { tc.cons[s] = tc.open_cons0[s] = tc.cons_before_prev[s] = tc.rel[s] = src[s].rel; tc.up[s] = src[s].up; }
    u32 &batch_no = tc.batch_no, (&cons)[2] = tc.cons, &acc_max_len = tc.acc_max_len, (&open_cons0)[2] = tc.open_cons0, &why = tc.why, &rounds = tc.rounds;
    u64 &done_reads = tc.done_reads, &names_done = tc.names_done, &bases_done = tc.bases_done, &acc_reads = tc.acc_reads, &acc_bases = tc.acc_bases, &acc_names = tc.acc_names;
    u64 &open_reads0 = tc.open_reads0, &open_names0 = tc.open_names0, &open_bases0 = tc.open_bases0;
    int &status = tc.status;
    float &ms_parse = tc.ms_parse;
    const bool &rolled_back = tc.rolled_back;
    auto flush_batch = [&]() -> int { return text_flush_batch(ctx, tw, tc); };
    const u32 lim = limit >= text_bytes[0] ? 0xFFFFFFFFu : (u32)limit + src[0].rel;
    const unsigned pgrid = (unsigned)ctx->n_cu * 8;
    // One round = one parse over [cons, hi) of every stream, hi = what piece k has brought up -- cut to the window one parse may
    // cover (of a pair of files the denser one is ahead of what its mate lets it hand over: its unparsed text waits, it does not grow
    // the window) -- whose records are appended to the open batch.  k moves on with the uploads; when they are all up the rounds go on
    // until the text is used up.
    const u32 window = (u32)std::min<u64>(range_cap - 64, 0x7FFFFFFFu);
    u32 k = 0;
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
        if (ctx->timing) TXCHK(hipEventRecord(tw.t0, st));
        TXCHK(hipMemsetAsync(d_ci, 0, info_bytes, st));
        ParseArgs pa{};
        pa.ci = d_ci; pa.n_streams = ns; pa.final_text = fin; pa.trim_readno = (flags & BNS_TEXT_TRIM_READNO) ? 1 : 0;
        u64 *st_words = (u64 *)((char *)tw.info.p + ((sizeof(CallInfo) + 63) & ~size_t(63)));
        u32 max_tiles = 1;
        for (u32 s = 0; s < ns; ++s) {
            // (the piece that ends this slice -- and with it every piece in front of it: the copy stream is in order)
            if (src[s].up) TXCHK(hipStreamWaitEvent(st, src[s].up->ev[std::min(src[s].j0 + k, src[s].j0 + src[s].n - 1)], 0));
            StreamArgs &a = pa.s[s];
            a.text = d_text[s]; a.lo = cons[s]; a.hi = hi[s];
            a.tile0 = a.lo / TILE; a.n_tiles = a.hi > a.lo ? (a.hi - 1) / TILE - a.tile0 + 1 : 1;
            a.cap_lines = cap_lines; a.cap_rec = cap_rec;
            a.ls = (u32 *)tw.ls[s].p; a.line_off = (u32 *)tw.line_off[s].p;
            u32 **cp[10] = {&a.cand, &a.c_next, &a.c_seq, &a.c_name, &a.c_line1, &a.c_single, &a.c_flags, &a.c_inside, &a.c_pos, &a.rec_cand};
            for (int i = 0; i < 10; ++i) *cp[i] = (u32 *)tw.cand[s][i].p;
            a.st_lines = st_words + s * st_tiles; a.st_groups = st_words + 2 * st_tiles + s * st_groups; a.st_compact = st_words + 2 * st_tiles + 2 * st_groups + s * st_cand;
            max_tiles = std::max(max_tiles, a.n_tiles);
        }
        pa.st_offsets = st_words + 2 * st_tiles + 2 * st_groups + 2 * st_cand;
        const u32 R0 = (u32)acc_reads;
        // the slice's records go behind those the open batch holds: offsets from acc_bases on, names from names_done on
        OffsetsOut oo{(u32 *)tw.seq_len[q].p + R0, (u64 *)tw.offsets.p + R0, acc_bases, (u32 *)tw.name_off[q].p + R0, (u32)names_done};
        PackOut po{(const u64 *)tw.offsets.p + R0, R0, (u64 *)tw.words.p, (u32 *)tw.nmask.p, (const u32 *)tw.name_off[q].p + R0, (u32)(names_done - acc_names),
                   (char *)tw.names[q].p, (u64 *)tw.pos64[q].p + R0, {src[0].rel, src[1].rel}};
        const unsigned cgrid = (unsigned)std::min<u64>(pgrid, (u64)cap_rec / REC_BLOCK + 1), ogrid = (unsigned)std::min<u64>(256, slice_reads_cap / REC_BLOCK + 1);
        hipLaunchKernelGGL(count_kernel, dim3(std::min<u32>(max_tiles, pgrid * 2), ns), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(write_kernel, dim3(std::min<u32>(max_tiles, pgrid * 2), ns), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(walk_kernel, dim3(pgrid, ns), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(compact_kernel, dim3(cgrid, ns), dim3(256), 0, st, pa, lim);
        hipLaunchKernelGGL(offsets_kernel, dim3(ogrid), dim3(256), 0, st, pa, oo);
        hipLaunchKernelGGL(pack_text_kernel, dim3(pgrid), dim3(256), 0, st, pa, po);
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
    if (flags & BNS_TEXT_DEFER) {
        // parsed and packed: the caller's host buffers are his again, the records are known; classify, the hit runs and every result
        // array wait for bns_text_finish (what a full batch made necessary on the way has been classified already)
        if ((rc = text_give_back_uploads(ctx, tw, tc)) != BNS_OK) return bail(rc);
        tc.pending = true;
        text_fill_info(ctx, tc, info);
        return BNS_OK;
    }
#undef TXCHK
    return text_wrap_up(ctx, tw, tc, info);
}

int bns_text_finish(bns_ctx *ctx, bns_text_info *info)
{
    if (!ctx || !info) return BNS_ERR_ARG;
    if (!ctx->text_work || !ctx->text_work->call || !ctx->text_work->call->pending) return fail(ctx, BNS_ERR_STATE, "bns_text_finish: no deferred bns_classify_text call is waiting");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::memset(info, 0, sizeof(*info));
    return text_wrap_up(ctx, *ctx->text_work, *ctx->text_work->call, info);
}

}  // extern "C"
