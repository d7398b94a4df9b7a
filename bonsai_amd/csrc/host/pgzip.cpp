// pgzip.hpp: a DEFLATE decoder that can start in the middle of a stream (RFC 1951 restated; no zlib in here).
#include "pgzip.hpp"

#include <algorithm>
#include <cstring>
#include <memory>

namespace bns {
namespace pgz {
namespace {
using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

// LSB-first bit reader over the whole file; positions are absolute bit offsets.  peek() yields at least 57 valid bits (zeros
// behind the end of the file: the callers check positions, not the reader).
struct Bits {
    const u8 *d;
    u64 n;
    u64 pos;
    u64 total() const { return n * 8; }
    inline u64 peek() const
    {
        const u64 byte = pos >> 3;
        u64 v = 0;
        if (byte + 8 <= n) std::memcpy(&v, d + byte, 8);
        else for (u64 i = byte; i < n; ++i) v |= (u64)d[i] << (8 * (i - byte));
        return v >> (pos & 7);
    }
    inline u32 take(unsigned k) { const u32 v = (u32)(peek() & ((1ULL << k) - 1)); pos += k; return v; }
};

constexpr u16 LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr u8 LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr u16 DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr u8 DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr u8 CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Canonical Huffman code: a direct table for codes of up to FB bits ((symbol << 4) | length; 0 = longer or absent), and the
// count / symbol arrays for the rest (decoded a bit at a time: rare symbols).
template <int FB, int NSYM>
struct Huff {
    u16 fast[1 << FB];
    u32 packed[1 << FB];             // the fast loop's form of `fast` (pack_litlen / pack_dist below); 0 = take the careful path
    u16 count[16];
    u16 symbol[NSYM];
    // lens[0, n): code lengths 0..15.  `allow_empty`: a code with no symbols is accepted (a block without distances).
    // As zlib's inflate_table: over-subscribed sets are rejected, incomplete ones too unless the set is a single 1-bit code.
    bool build(const u8 *lens, int n, bool allow_empty)
    {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) ++count[lens[i]];
        std::memset(fast, 0, sizeof(fast));
        if (count[0] == n) return allow_empty;
        int left = 1, maxlen = 0;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return false;
            if (count[l]) maxlen = l;
        }
        if (left > 0 && maxlen != 1) return false;
        u16 offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (u16)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (lens[i]) symbol[offs[lens[i]]++] = (u16)i;
        // fast table: canonical codes in order, bit-reversed (the stream is LSB-first, codes are MSB-first)
        u32 code = 0, idx = 0;
        for (int l = 1; l <= FB && l <= maxlen; ++l) {
            for (u32 c = 0; c < count[l]; ++c, ++code, ++idx) {
                u32 r = 0;
                for (int b = 0; b < l; ++b) r |= ((code >> b) & 1u) << (l - 1 - b);
                const u16 e = (u16)((symbol[idx] << 4) | l);
                for (u32 i = r; i < (1u << FB); i += 1u << l) fast[i] = e;
            }
            code <<= 1;
        }
        return true;
    }
    // one symbol from the low bits of v (>= 15 of them valid): false = no such code
    inline bool slow(u64 v, u32 &sym, u32 &len) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; ++l) {
            code |= (int)(v & 1);
            v >>= 1;
            const int cnt = count[l];
            if (code - cnt < first) { sym = symbol[index + (code - first)]; len = (u32)l; return true; }
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        return false;
    }
    inline bool decode(u64 v, u32 &sym, u32 &len) const
    {
        const u32 e = fast[v & ((1u << FB) - 1)];
        if (e) { sym = e >> 4; len = e & 15u; return true; }
        return slow(v, sym, len);
    }
};
using LitHuff = Huff<11, 288>;
using DistHuff = Huff<10, 32>;
using ClHuff = Huff<7, 19>;

// lit2: the literal / length code indexed by 12 bits, TWO literals per entry where both codes fit (FASTQ is mostly literals of 2-6
// bits: the serial lookup -> shift -> lookup chain is what bounds the decoder, and this halves its length)
struct Tables { LitHuff lit; DistHuff dist; u32 lit2[1 << 12]; };

// packed entry: bits 0-3 code length, 4-7 extra bits, 8-23 base (the literal / the length's or distance's base), 24-25 kind
// lit2 entries: kind K_LIT2 = bits 8-15 first literal, 16-23 second, 0-3... the two codes' total length in bits 0-4
constexpr u32 K_LIT = 0u << 24, K_LEN = 1u << 24, K_EOB = 2u << 24, K_LIT2 = 3u << 24, K_MASK = 3u << 24;
void pack_tables(Tables &t)
{
    for (u32 i = 0; i < (1u << 11); ++i) {
        const u32 e = t.lit.fast[i], sym = e >> 4, len = e & 15u;
        u32 pk = 0;
        if (e) {
            if (sym < 256) pk = K_LIT | (sym << 8) | len;
            else if (sym == 256) pk = K_EOB | len;
            else if (sym <= 285) pk = K_LEN | ((u32)LBASE[sym - 257] << 8) | ((u32)LEXT[sym - 257] << 4) | len;
        }
        t.lit.packed[i] = pk;                                    // (symbols 286 / 287: 0 -> the careful path reports them)
    }
    for (u32 i = 0; i < (1u << 10); ++i) {
        const u32 e = t.dist.fast[i], sym = e >> 4, len = e & 15u;
        t.dist.packed[i] = (e && sym <= 29) ? (((u32)DBASE[sym] << 8) | ((u32)DEXT[sym] << 4) | len) : 0u;
    }
    for (u32 i = 0; i < (1u << 12); ++i) {
        const u32 e1 = t.lit.packed[i & 2047u];
        u32 pk = e1;                                              // (one symbol: the 11-bit entry as it is -- 0 included)
        if (e1 && (e1 & K_MASK) == K_LIT) {
            const u32 l1 = e1 & 15u;
            const u32 e2 = t.lit.packed[(i >> l1) & 2047u], l2 = e2 & 15u;
            // the second code must be decided by the bits this index has left
            if (e2 && (e2 & K_MASK) == K_LIT && l1 + l2 <= 12u) pk = K_LIT2 | (((e2 >> 8) & 0xFFu) << 16) | (e1 & 0xFF00u) | (l1 + l2);
        }
        t.lit2[i] = pk;
    }
}

const Tables &fixed_tables()
{
    static const Tables t = [] {
        Tables x;
        u8 l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        x.lit.build(l, 288, false);
        u8 d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        x.dist.build(d, 32, false);
        pack_tables(x);
        return x;
    }();
    return t;
}

// the header of a dynamic block behind its three type bits: false = not a valid header
bool read_dynamic_header(Bits &b, Tables &t)
{
    if (b.pos + 14 > b.total()) return false;
    const u32 h = b.take(14);
    const int nlen = (int)(h & 31) + 257, ndist = (int)((h >> 5) & 31) + 1, ncode = (int)((h >> 10) & 15) + 4;
    if (nlen > 286 || ndist > 30) return false;
    u8 cl[19] = {0};
    if (b.pos + 3 * (u64)ncode > b.total()) return false;
    {
        u64 v = b.peek();                                       // 57 bits: all 19 three-bit lengths
        for (int i = 0; i < ncode; ++i) { cl[CLORDER[i]] = (u8)(v & 7); v >>= 3; }
        b.pos += 3 * (u64)ncode;
    }
    ClHuff ch;
    {
        // (complete, and not the single-code special case: a code-length code of one symbol cannot describe a block)
        int left = 1, used = 0;
        for (int i = 0; i < 19; ++i) used += cl[i] != 0;
        if (used < 2) return false;
        u16 cnt[8] = {0};
        for (int i = 0; i < 19; ++i) ++cnt[cl[i]];
        for (int l = 1; l <= 7; ++l) { left <<= 1; left -= cnt[l]; if (left < 0) return false; }
        if (left != 0) return false;
        if (!ch.build(cl, 19, false)) return false;
    }
    u8 lens[286 + 30 + 138];
    int i = 0;
    const int total = nlen + ndist;
    while (i < total) {
        if (b.pos >= b.total()) return false;
        const u64 v = b.peek();
        u32 s, l;
        if (!ch.decode(v, s, l)) return false;
        b.pos += l;
        if (s < 16) { lens[i++] = (u8)s; continue; }
        u32 rep;
        u8 val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (u32)((v >> l) & 3); b.pos += 2; }
        else if (s == 17) { rep = 3 + (u32)((v >> l) & 7); b.pos += 3; }
        else { rep = 11 + (u32)((v >> l) & 127); b.pos += 7; }
        if (i + (int)rep > total) return false;
        std::memset(lens + i, val, rep);
        i += (int)rep;
    }
    if (lens[256] == 0) return false;                            // no end-of-block code
    if (!t.lit.build(lens, nlen, false)) return false;
    if (!t.dist.build(lens + nlen, ndist, true)) return false;
    return b.pos <= b.total();
}
// (the header search builds tables for candidates that mostly die a few symbols later: the packed form is made when a block's
// data is about to be decoded)


struct Out {
    std::vector<u16> &v;
    u64 n;                                                       // write index (marker prefix included)
    u64 limit;                                                   // symbols a chunk may expand to (a guard, not a format limit)
    bool room(u64 want)
    {
        if (n + want > limit) return false;
        if (n + want > v.size()) v.resize(std::min<u64>(limit, std::max<u64>(n + want, v.size() + v.size() / 2)));
        return true;
    }
};

// the symbols of one Huffman-coded block up to and including its end-of-block code.  min_src: the lowest index a back-reference
// may read (0: the marker prefix is fair game; more: the start of the current gzip member).
bool inflate_codes(Bits &b, const Tables &t, Out &o, u64 min_src)
{
    const u64 total = b.total();
    u16 *out = o.v.data();
    u64 cap = o.v.size(), n = o.n;
    constexpr u32 LMASK = (1u << 11) - 1, DMASK = (1u << 10) - 1;
    const u8 *const d = b.d;
    for (;;) {
        if (n + 280 > cap) { o.n = n; if (!o.room(1u << 20)) return false; out = o.v.data(); cap = o.v.size(); }
        // ---- fast loop: far from the end of the input and of the buffer, packed table entries, no per-symbol checks.  Leaves
        // for ONE pass through the careful code below on anything unusual (a code longer than the table's bits, an invalid
        // symbol, a reference too far back) -- which then decides.
        {
            u64 pos = b.pos;
            while (pos + 128 <= total && n + 280 <= cap) {
                u64 v;
                std::memcpy(&v, d + (pos >> 3), 8);
                v >>= (pos & 7);
                u32 e = t.lit2[v & 4095u];
                if ((e & K_MASK) == K_LEN) {
                    u32 used = e & 15u;
                    v >>= used;
                    const u32 lx = (e >> 4) & 15u;
                    const u32 L = ((e >> 8) & 0xFFFFu) + (u32)(v & ((1u << lx) - 1));
                    v >>= lx; used += lx;
                    const u32 de = t.dist.packed[v & DMASK];
                    if (!de) break;
                    const u32 dl = de & 15u, dx = (de >> 4) & 15u;
                    v >>= dl;
                    const u32 D = (de >> 8) + (u32)(v & ((1u << dx) - 1));
                    if ((u64)D + min_src > n) break;
                    pos += used + dl + dx;
                    u16 *dst = out + n;
                    const u16 *src = dst - D;
                    if (D >= 8) for (u32 i = 0; i < L; i += 8) std::memcpy(dst + i, src + i, 16);      // (may write up to 7 symbols past L: room is there)
                    else for (u32 i = 0; i < L; ++i) dst[i] = src[i];
                    n += L;
                    continue;
                }
                if (e == 0 || (e & K_MASK) == K_EOB) break;          // end of block, or a long / invalid code: the careful path
                // literals: one or two per entry, up to four entries out of one load (4 x 12 bits)
                u32 used = 0;
#define BNS_PGZ_LITS(last)                                                                                               \
                if ((e & K_MASK) == K_LIT2) { out[n] = (u16)((e >> 8) & 0xFFu); out[n + 1] = (u16)((e >> 16) & 0xFFu); n += 2; used += e & 31u; v >>= (e & 31u); } \
                else { out[n++] = (u16)((e >> 8) & 0xFFu); used += e & 15u; v >>= (e & 15u); }                          \
                if (!(last)) { e = t.lit2[v & 4095u]; if (e == 0 || (e & K_MASK) == K_LEN || (e & K_MASK) == K_EOB) { pos += used; continue; } }
                BNS_PGZ_LITS(false)
                BNS_PGZ_LITS(false)
                BNS_PGZ_LITS(false)
                BNS_PGZ_LITS(true)
#undef BNS_PGZ_LITS
                pos += used;
            }
            b.pos = pos;
        }
        // the fast loop leaves with as little as 22 symbols of room (a match of 258 from n == cap - 280); the careful path below
        // may write another 258: back to the top for room() first
        if (n + 280 > cap) continue;
        if (b.pos >= total) return false;
        u64 v = b.peek();
        u32 s, len, used;
        {
            const u32 e = t.lit.fast[v & LMASK];
            if (e) { s = e >> 4; len = e & 15u; }
            else if (!t.lit.slow(v, s, len)) return false;
        }
        v >>= len;
        used = len;
        if (s < 256) {
            out[n++] = (u16)s;
            // up to two more literals out of the same 57 bits
            u32 e = t.lit.fast[v & LMASK];
            if (e && e < (256u << 4)) {
                out[n++] = (u16)(e >> 4); v >>= (e & 15u); used += e & 15u;
                e = t.lit.fast[v & LMASK];
                if (e && e < (256u << 4)) { out[n++] = (u16)(e >> 4); used += e & 15u; }
            }
            b.pos += used;
            continue;
        }
        if (s == 256) { b.pos += used; o.n = n; return b.pos <= total; }
        if (s > 285) return false;
        s -= 257;
        const u32 L = LBASE[s] + (u32)(v & ((1u << LEXT[s]) - 1));
        v >>= LEXT[s];
        used += LEXT[s];
        u32 ds;
        if (!t.dist.decode(v, ds, len)) return false;
        if (ds > 29) return false;
        v >>= len;
        used += len;
        const u32 D = DBASE[ds] + (u32)(v & ((1u << DEXT[ds]) - 1));
        used += DEXT[ds];
        b.pos += used;
        if ((u64)D + min_src > n) return false;                  // reaches in front of what may be referenced
        const u64 src = n - D;
        if (D >= L) std::memcpy(out + n, out + src, 2 * (size_t)L);
        else for (u32 i = 0; i < L; ++i) out[n + i] = out[src + i];
        n += L;
    }
}

u32 le32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// blocks from the header at start_bit on.  max_blocks: stop after that many (header validation), 0 = until stop_bit.
bool decode_from(const u8 *data, u64 n, u64 start_bit, bool fresh, u64 stop_bit, Scan &s, unsigned max_blocks = 0)
{
    Bits b{data, n, start_bit};
    s.ok = false; s.eof = false; s.segs.clear(); s.err.clear();
    s.start_bit = start_bit;
    if (s.sym.size() < WINDOW + (1u << 20)) s.sym.resize(WINDOW + (4u << 20));
    for (u32 j = 0; j < WINDOW; ++j) s.sym[j] = (u16)(MARKER0 + j);
    // (a chunk may expand to 256x its compressed bytes + 64 Mi symbols -- FASTQ expands 3-5x, DEFLATE's limit is 1032x; the bound
    // keeps a few MB of zeros from asking for tens of GB of symbol buffers: such a file is read with BNS_NO_PGZ=1)
    const u64 span = stop_bit > start_bit ? (stop_bit - start_bit) / 8 : 0;
    Out o{s.sym, WINDOW, WINDOW + (64ull << 20) + 256 * span};
    u64 min_src = fresh ? WINDOW : 0;
    u64 seg_begin = 0;
    std::unique_ptr<Tables> dyn(new Tables);
    unsigned blocks = 0;
    for (;;) {
        if (b.pos >= stop_bit && !(max_blocks && blocks < max_blocks)) break;
        if (max_blocks && blocks >= max_blocks) break;
        if (b.pos + 3 > b.total()) { s.err = "the deflate stream ends without a final block"; return false; }
        const u32 hdr = b.take(3);
        const bool final_ = hdr & 1;
        const u32 type = hdr >> 1;
        if (type == 0) {
            b.pos = (b.pos + 7) & ~7ULL;
            const u64 at = b.pos >> 3;
            if (at + 4 > n) { s.err = "truncated stored block"; return false; }
            const u32 len = data[at] | ((u32)data[at + 1] << 8), nlen = data[at + 2] | ((u32)data[at + 3] << 8);
            if ((len ^ nlen) != 0xFFFFu) { s.err = "stored block lengths do not match"; return false; }
            if (at + 4 + len > n) { s.err = "truncated stored block"; return false; }
            if (!o.room(len + 1)) { s.err = "a chunk of the gzip input expands beyond the reader's bound (BNS_NO_PGZ=1 reads it with zlib)"; return false; }
            for (u32 i = 0; i < len; ++i) s.sym[o.n + i] = data[at + 4 + i];
            o.n += len;
            b.pos = (at + 4 + len) * 8;
        } else if (type == 1) {
            if (!inflate_codes(b, fixed_tables(), o, min_src)) { s.err = "invalid data in a fixed-Huffman block"; return false; }
        } else if (type == 2) {
            if (!read_dynamic_header(b, *dyn)) { s.err = "invalid dynamic block header"; return false; }
            pack_tables(*dyn);
            if (!inflate_codes(b, *dyn, o, min_src)) { s.err = "invalid data in a dynamic-Huffman block"; return false; }
        } else { s.err = "invalid block type"; return false; }
        ++blocks;
        if (final_) {
            b.pos = (b.pos + 7) & ~7ULL;
            const u64 at = b.pos >> 3;
            if (at + 8 > n) { s.err = "the gzip trailer is missing (truncated file)"; return false; }
            Seg g;
            g.begin = seg_begin; g.end = o.n - WINDOW; g.member_end = true; g.crc = le32(data + at); g.isize = le32(data + at + 4);
            s.segs.push_back(g);
            seg_begin = g.end;
            // (as gzread: another member only when a gzip header follows at once; anything else -- zero padding too -- is trailing
            // garbage and ends the stream)
            const u64 next = at + 8;
            const u64 he = next < n ? gzip_header_end(data, n, next) : 0;
            if (!he) { s.eof = true; b.pos = n * 8; break; }     // nothing (valid) behind the trailer: end of the stream
            b.pos = he * 8;
            min_src = o.n;                                       // a new member: nothing in front of it can be referenced
        }
    }
    if (seg_begin < o.n - WINDOW) { Seg g; g.begin = seg_begin; g.end = o.n - WINDOW; s.segs.push_back(g); }
    s.n_out = o.n - WINDOW;
    s.end_bit = b.pos;
    s.ok = true;
    return true;
}

}  // namespace

uint64_t gzip_header_end(const uint8_t *d, uint64_t n, uint64_t at)
{
    if (at + 18 > n || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8 || (d[at + 3] & 0xE0)) return 0;
    const u8 flg = d[at + 3];
    u64 p = at + 10;
    if (flg & 4) { if (p + 2 > n) return 0; p += 2 + (d[p] | ((u64)d[p + 1] << 8)); }
    if (flg & 8) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 16) { while (p < n && d[p]) ++p; ++p; }
    if (flg & 2) p += 2;
    return p < n ? p : 0;
}

bool scan_chunk(const uint8_t *data, uint64_t n, uint64_t from_bit, bool search, bool fresh, uint64_t stop_bit, Scan &s)
{
    if (!search) return decode_from(data, n, from_bit, fresh, stop_bit, s);
    // Candidates: a non-final dynamic block (type bits 0, 10) whose counts are in range -- one position in nine -- then the full
    // header (complete code-length code, complete literal/length and distance codes: one in ~10^5 of those), then the data.
    Bits b{data, n, from_bit};
    const u64 total = n * 8;
    std::unique_ptr<Tables> probe(new Tables);
    // (a chunk may lie inside one block, so the search goes on behind its end -- for 4 MiB, several times the largest block any
    // deflate writer makes; further on the caller's fallback, decoding from where the chunk in front ends, is the cheaper answer)
    const u64 limit = std::min<u64>(total, std::max(from_bit, stop_bit) + (32ull << 20));
    for (u64 pos = from_bit; pos + 17 + 12 <= limit; ++pos) {
        b.pos = pos;
        const u64 v = b.peek();
        if ((v & 7) != 4) continue;                              // BFINAL = 0, BTYPE = 10 (LSB first: 0, then 0 1)
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;
        b.pos = pos + 3;
        if (!read_dynamic_header(b, *probe)) continue;
        if (pos >= stop_bit) {
            // a chunk that lies inside one block: the header is checked on its own block, nothing is kept
            Scan tmp;
            if (!decode_from(data, n, pos, false, stop_bit, tmp, 1)) continue;
            s.ok = true; s.eof = false; s.start_bit = s.end_bit = pos; s.n_out = 0; s.segs.clear();
            if (s.sym.size() < WINDOW) s.sym.resize(WINDOW);
            for (u32 j = 0; j < WINDOW; ++j) s.sym[j] = (u16)(MARKER0 + j);
            return true;
        }
        if (decode_from(data, n, pos, false, stop_bit, s)) return true;
    }
    s.ok = false;
    s.err = "no block header found";
    return false;
}

void resolve(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out)
{
    std::unique_ptr<u8[]> lut(new u8[MARKER0 + WINDOW]);
    for (u32 i = 0; i < 256; ++i) lut[i] = (u8)i;
    std::memcpy(lut.get() + MARKER0, window, WINDOW);
    const u8 *t = lut.get();
    for (size_t i = 0; i < n; ++i) out[i] = t[sym[i]];
}

void next_window(const Scan &s, const uint8_t *window, uint8_t *out_window)
{
    const u16 *p = s.sym.data() + s.n_out;                       // the last WINDOW symbols of prefix + output
    for (u32 j = 0; j < WINDOW; ++j) { const u16 x = p[j]; out_window[j] = x < MARKER0 ? (u8)x : window[x - MARKER0]; }
}

}  // namespace pgz
}  // namespace bns
