// One plain gzip stream inflated on many threads (host ingest; the reference reads through gzFile:
// kseq_declare.h:112-145, klib/kseq.h:177-225 -- one zlib inflate at ~0.4 GB/s of text, 1-2 M reads/s).
//
// DEFLATE has no sync points, but a block header is redundant enough to be FOUND: the file is cut into chunks of compressed
// bytes, every chunk is scanned bit by bit for a dynamic-Huffman block header whose code lengths form complete codes, and decoded
// from there to the first block boundary at or behind its end -- without the 32 KiB of text in front of it.  What a
// back-reference into that unknown window would copy is written as a MARKER (symbol 256 + window position) into an output of
// 16-bit symbols; when the chunk in front has been resolved, so is this one: symbol -> byte through a table of the window
// (two passes over the text instead of one, on as many threads as there are chunks in flight).  Chunks must meet -- a scan that
// ends at bit e is followed by one that starts at bit e, else the follower is decoded again from e -- and every gzip member's
// CRC-32 and length are checked against its trailer, so a false block start cannot pass.  (The scheme of pugz / rapidgzip; this
// is an implementation of our own: inflate, header search, marker symbols, resolution.)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace bns {
namespace pgz {

constexpr uint32_t WINDOW = 32768;                 // DEFLATE's back-reference reach
constexpr uint16_t MARKER0 = 256;                  // symbol 256 + j = "byte j of the 32 KiB in front of this chunk"

// a stretch of a chunk's output that belongs to one gzip member (CRC-32 runs per member)
struct Seg {
    uint64_t begin = 0, end = 0;                   // output positions (symbols behind the marker prefix)
    bool member_end = false;                       // the member ends with this stretch: crc / isize are its trailer
    uint32_t crc = 0, isize = 0;
};

struct Scan {
    bool ok = false;
    bool eof = false;                              // the stream ended inside this chunk (last member's trailer seen, nothing valid behind)
    uint64_t start_bit = 0, end_bit = 0;           // first block header decoded; where the next chunk's first header is
    std::vector<uint16_t> sym;                     // WINDOW marker symbols, then the output
    uint64_t n_out = 0;                            // symbols behind the prefix
    std::vector<Seg> segs;
    std::string err;
};

// offset of the deflate data of the gzip member whose header starts at `at`; 0 = no gzip header there
uint64_t gzip_header_end(const uint8_t *data, uint64_t n, uint64_t at);

// Decode blocks from a header at or behind from_bit until a block ends at or behind stop_bit (or the stream ends).
//   search     : from_bit is only where to START LOOKING for a (non-final, dynamic-Huffman) block header
//   !search    : a block header starts exactly at from_bit
//   fresh      : (with !search) from_bit is the first block of a member -- nothing in front of it can be referenced
// data[0, n) is the whole file.  False: nothing decodable (s.err says why).
bool scan_chunk(const uint8_t *data, uint64_t n, uint64_t from_bit, bool search, bool fresh, uint64_t stop_bit, Scan &s);

// symbols -> bytes: window = the WINDOW bytes in front of the chunk (resolved)
void resolve(const uint16_t *sym, size_t n, const uint8_t *window, uint8_t *out);
// the resolved window behind a chunk: the last WINDOW symbols of (marker prefix + output) through the previous window
void next_window(const Scan &s, const uint8_t *window, uint8_t *out_window);

}  // namespace pgz
}  // namespace bns
