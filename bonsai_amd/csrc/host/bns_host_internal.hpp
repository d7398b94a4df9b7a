// bns_host_internal.hpp -- what the translation units of the host library share with each other and with nobody else.
#pragma once
#include "bns_host.hpp"
#include "pgzip.hpp"

#include "bns_host.hpp"
#include "pgzip.hpp"

#include <zlib.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <emmintrin.h>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <condition_variable>
#include <deque>
#include <memory>
#include <map>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sched.h>
#include <unistd.h>


namespace bns {

[[noreturn]] inline void die(const std::string &msg) { throw Error(msg); }
inline void chk(bns_ctx *ctx, int rc, const char *what)
{
    if (rc == BNS_OK) return;
    std::string m = std::string(what) + ": " + bns_strerror(rc);
    if (ctx) { m += " ("; m += bns_last_error(ctx); m += ")"; }
    die(m);
}
inline double tnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// static split of [0, n_units) over nt host threads (-p)
template <typename F>
void parallel_units(unsigned nt, unsigned n_units, F &&fn)
{
    if (nt <= 1) { fn(0u, n_units, 0u); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] { fn((unsigned)((u64)n_units * t / nt), (unsigned)((u64)n_units * (t + 1) / nt), t); });
    for (auto &x : th) x.join();
}

inline void pread_all(int fd, void *dst, size_t n, u64 at, const char *what)
{
    char *d = static_cast<char *>(dst);
    for (size_t got = 0; got < n;) {
        const ssize_t r = ::pread(fd, d + got, n - got, (off_t)(at + got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) die(std::string("short read in ") + what);
        got += (size_t)r;
    }
}

// ---- text blocks of the host reader (bns_reader.cpp) and of ChunkSource (bns_chunks.cpp)
constexpr size_t TEXT_BLOCK_HEAD = 64u << 10;                    // room in front of a raw block's text for the unparsed tail of the block before it
// A text block: [begin, end) of an uninitialised buffer; raw blocks leave HEAD bytes free in front so that the unparsed
// tail of the previous block (normally one partial record) can be put there without copying the block itself.
// Buffers of text blocks are recycled: a fresh 4-16 MiB allocation is an mmap plus a page fault per 4 KiB on first touch,
// which costs more than parsing the block.
class BlockPool {
public:
    char *get(size_t cap, size_t &got_cap)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].second >= cap && free_[i].second <= 2 * cap) {
                    char *p = free_[i].first; got_cap = free_[i].second;
                    free_[i] = free_.back(); free_.pop_back();
                    return p;
                }
        }
        got_cap = cap;
        return new char[cap];
    }
    void put(char *p, size_t cap)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (free_.size() < keep_) { free_.emplace_back(p, cap); return; }
        }
        delete[] p;
    }
    ~BlockPool() { for (auto &f : free_) delete[] f.first; }
    // how many idle buffers are kept (a reader that publishes hundreds of blocks at a time -- the GPU inflater's batches -- wants
    // that many back: a 4 MiB buffer that is freed and allocated again is an munmap, an mmap and a thousand page faults, all of them
    // under the address-space lock the other threads' faults wait for)
    void keep_at_least(size_t n) { std::lock_guard<std::mutex> lk(mu_); keep_ = std::max(keep_, n); }
private:
    size_t keep_ = 24;
    std::mutex mu_;
    std::vector<std::pair<char *, size_t>> free_;
};
BlockPool &block_pool();                                        // (bns_reader.cpp; leaked on purpose: blocks may outlive static destruction order)

// A text block: [begin, end) of an uninitialised buffer; raw blocks leave HEAD bytes free in front so that the unparsed
// tail of the previous block (normally one partial record) can be put there without copying the block itself.
struct TextBlock {
    char *buf_ = nullptr;
    size_t cap = 0, begin = 0, end = 0;
    std::deque<std::deque<std::string>> arenas;   // fields of this block's records that are not contiguous in the text (multi-line)
    explicit TextBlock(size_t capacity) { buf_ = block_pool().get(capacity, cap); }
    ~TextBlock() { block_pool().put(buf_, cap); }
    TextBlock(const TextBlock &) = delete;
    TextBlock &operator=(const TextBlock &) = delete;
    char *raw() { return buf_; }
    const char *data() const { return buf_ + begin; }
    size_t size() const { return end - begin; }
};

size_t bgzf_member(const unsigned char *p, size_t n, size_t &payload_off);     // bns_reader.cpp: size of the BGZF member at p (0: none), where its payload starts
long find_record_start(const char *b, size_t n, bool fastq);                   // bns_chunks.cpp: where a record certainly starts in b[0, n) (-1: nowhere to be sure of)
static inline void trim_readno(std::string_view &s)            // kseq_declare.h:106-110
{
    const size_t l = s.size();
    if (l > 2 && s[l - 2] == '/' && (unsigned)(s[l - 1] - '0') < 10u) s.remove_suffix(2);
}

// ---- formatting (bns_host.cpp has the std::string forms; these write through a raw pointer)
// raw-pointer twins of put_unsigned / append_counts / append_taxa_runs for the hot formatter below
inline char *wr_unsigned(char *w, u32 x)
{
    char tmp[12]; int n = 0;
    if (x == 0) tmp[n++] = '0';
    while (x) { tmp[n++] = char('0' + x % 10); x /= 10; }
    while (n) *w++ = tmp[--n];
    return w;
}
inline char *wr_counts(char *w, u32 count, char ch)
{
    if (!count) return w;
    *w++ = ch; *w++ = ':'; w = wr_unsigned(w, count); *w++ = '\t';
    return w;
}
// classifier.h:112-129, written through a raw pointer into room the caller reserved (kraken_line_bound), not byte by byte through
// push_back: the formatter was 65 ns per read, the slowest stage of the CLI.
inline size_t kraken_line_bound(const HitRuns &runs, const bseq1_t &bs) { return bs.name.size() + 64 + (size_t)runs.n * 24; }
inline char *kraken_line_raw(char *w, const HitRuns &runs, tax_t taxon, u32 ambig_count, u32 missing_count, const bseq1_t &bs)
{
    *w++ = taxon ? 'C' : 'U'; *w++ = '\t';
    std::memcpy(w, bs.name.data(), bs.name.size()); w += bs.name.size(); *w++ = '\t';
    w = wr_unsigned(w, taxon); *w++ = '\t';
    const int l = bs.l_seq();
    if (l < 0) { *w++ = '-'; w = wr_unsigned(w, (u32)(-l)); } else w = wr_unsigned(w, (u32)l);
    *w++ = '\t';
    w = wr_counts(w, missing_count, 'M');
    w = wr_counts(w, ambig_count, 'A');
    if (!taxon) { std::memcpy(w, "0:0\n", 4); w += 4; }
    else {
        for (u32 i = 0; i < runs.n; ++i) {
            if (runs.tax[i] == 0) *w++ = 'U';
            else if (runs.tax[i] == (tax_t)-1) *w++ = 'A';
            else w = wr_unsigned(w, runs.tax[i]);
            *w++ = ':'; w = wr_unsigned(w, runs.len[i]); *w++ = '\t';
        }
        w[-1] = '\n';
    }
    return w;
}

// ---- a chunk of host-parsed records on one device (bns_host.cpp): pack into the result's page-locked buffers, then the GPU call
void pack_chunk(ClassifierGeneric &c, bns_ctx *ctx, const bseq1_t *bs, unsigned n, int is_paired, ChunkResult &r, unsigned copy_threads);
void call_chunk(bns_ctx *ctx, ChunkResult &r);

// ---- text parsed on the device (bns_text_pipeline.cpp): which inputs take that path, and the pipelines.  -> how far they got: the file
// offset the host parser goes on from (one plain file), or false + the units that were printed (the host parser reads the input again
// and leaves those out)
bool text_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2);
bool pair_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2);
bool bgzf_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2);
bool bgzf_pair_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2);
u64 process_text_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out);
bool process_text_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done);
bool process_bgzf_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out, u64 &units_done);
bool process_bgzf_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done);
bool gz_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2);
bool process_gz_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out, u64 &units_done);
bool gz_pair_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2);
bool process_gz_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done);

}  // namespace bns
