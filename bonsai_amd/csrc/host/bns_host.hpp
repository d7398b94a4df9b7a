// bns_host.hpp -- C++17 host side of the classify path, above the C ABI (include/bonsai_amd.h).
//
// Mirrors the reference's host objects for this path, same names and argument meaning, new internals:
//   bns::Database            include/bonsai/database.h:16-111   bns.db = {k, w, spacing} + khash dump
//   bns::build_parent_map    include/bonsai/util.h:766-785      nodes.dmp -> parent map (flat array here)
//   bns::SeqReader/bseq_read include/bonsai/kseq_declare.h:106-175 + klib/kseq.h:177-225
//   bns::ClassifierGeneric   include/bonsai/classifier.h:131-172
//   bns::classify_seqs       include/bonsai/classifier.h:269-287 (the kt_forpool fan-out becomes ONE C-ABI call)
//   bns::process_dataset     include/bonsai/classifier.h:296-337
//   bns::Encoder             include/bonsai/encoder.h:113-638    (for_each over the GPU encoder)
#pragma once
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../../include/bonsai_amd.h"

namespace bns {

using u8 = std::uint8_t;
using u16 = std::uint16_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;
using tax_t = u32;                              // util.h:132
using spvec_t = std::vector<u16>;               // spacer.h:12

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// spacer.h:29-47: "a,b,c" or "gapxrepeat,..."; empty => k-1 zeros
spvec_t parse_spacing(const char *s, unsigned k);

// ---- bns.db -----------------------------------------------------------------------------------------
// The db's arrays are gigabytes: anonymous memory of their own (huge pages where the kernel gives them) that resize() does NOT fill with
// zeros -- the reader overwrites every byte, from several threads, and the pages are first touched there (a std::vector<u64> of 2^29
// entries spent a second zero-filling 6.6 GB on one thread before the first byte of the file was read).
void *big_alloc(size_t bytes);
void big_free(void *p, size_t bytes);
template <typename T>
struct BigAlloc {
    using value_type = T;
    BigAlloc() = default;
    template <class U> BigAlloc(const BigAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(big_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { big_free(p, n * sizeof(T)); }
    template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }              // (default-initialised: untouched)
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...); }
    template <class U> bool operator==(const BigAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const BigAlloc<U> &) const { return false; }
};
template <typename T> using BigVec = std::vector<T, BigAlloc<T>>;

// khash_t(c) as it sits on disk (util.h:280-293): header + flags/keys/vals arrays.
struct KhashC {
    u64 n_buckets = 0, n_occupied = 0, size = 0, upper_bound = 0;
    BigVec<u32> flags;
    BigVec<u64> keys;
    BigVec<u32> vals;
    bool exists(u64 i) const { return ((flags[i >> 4] >> ((i & 0xfU) << 1)) & 3U) == 0; }   // khash64.h:171
};

struct Database {
    unsigned k_ = 0, w_ = 0;
    spvec_t s_;                 // "extra gap" values, k-1 of them (database.h:22,46-48)
    KhashC db_;
    int spacing_width_ = 1;     // 1 = as database.h:46-48 reads, 2 = as the gz writer emits (database.h:89)
    Database() = default;
    explicit Database(const char *path);                       // reads plain or gzip; either spacing width
    void write(const char *path, int spacing_width = 1) const; // ".gz" suffix => gzip (database.h:81-102)
};

// util.h:766-785.  parent[id], BNS_TAX_ABSENT where id is not a key; parent[1] == 0.
std::vector<u32> build_parent_map(const char *nodes_dmp);

// ---- reads --------------------------------------------------------------------------------------------
struct bseq1_t {                // kseq_declare.h:40-44; the fields are VIEWS into memory owned by a ReadChunk (below)
    std::string_view name, comment, seq, qual;   // (no `sam`: result text goes straight into the chunk's output string)
    int l_seq() const { return (int)seq.size(); }
};

static_assert(sizeof(bseq1_t) == 64, "a record is one cache line");

// The records of a chunk: a plain array on a cache-line boundary.  The reader thread writes it, the packer and formatter threads read
// it, and the memory is recycled -- so when the reader comes back to it its lines sit in other cores' caches.  push_back_stream()
// writes a record with non-temporal stores (one full line, no read-for-ownership round trip to whichever core complex had it last:
// bseq_read 0.85 -> 0.4 s per 16 M reads on the two-socket host); publish() fences them before the chunk changes threads.
class RecVec {
public:
    RecVec() = default;
    ~RecVec() { std::free(p_); }
    RecVec(const RecVec &) = delete;
    RecVec &operator=(const RecVec &) = delete;
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    bseq1_t *data() { return p_; }
    const bseq1_t *data() const { return p_; }
    bseq1_t &operator[](size_t i) { return p_[i]; }
    const bseq1_t &operator[](size_t i) const { return p_[i]; }
    bseq1_t &back() { return p_[n_ - 1]; }
    const bseq1_t *begin() const { return p_; }
    const bseq1_t *end() const { return p_ + n_; }
    void clear() { n_ = 0; }
    void reserve(size_t cap);
    void push_back(const bseq1_t &r) { if (n_ == cap_) reserve(cap_ ? 2 * cap_ : 1024); p_[n_++] = r; }
    void push_back_stream(const bseq1_t &r);
    static void publish();
private:
    bseq1_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

struct TextBlock;                // a block of file text (bns_host.cpp)

// What the records of one bseq_read call point into: the raw text blocks they were parsed from (a single-line
// sequence / quality / name is a view straight into the file text, nothing is copied) and an arena for the fields
// that are not contiguous in the file (multi-line sequences).
struct ReadChunk {
    RecVec recs;
    std::vector<std::shared_ptr<const TextBlock>> blocks;
    std::deque<std::string> arena;
    u64 epoch = 0;              // bumped by clear(): lets a reader know whether it still has its block registered here
    void clear() { recs.clear(); blocks.clear(); arena.clear(); ++epoch; }
};

// kseq_read (klib/kseq.h:177-225) over a gz/plain file, block-wise: a background thread inflates / reads 4 MiB
// blocks, the caller's thread parses records out of them with memchr-speed line scans (35 M reads/s of 150-bp FASTQ).  Same record semantics as
// kseq: a record starts at '>' or '@', name = first whitespace-delimited token, comment = rest of the header line,
// sequence lines are joined until a line starts with '>', '@' or '+', a trailing '\r' is dropped from a line when what
// has been accumulated is longer than one character, quality lines are joined until they are as long as the sequence.
struct TextBlock;
class SeqReader {
public:
    // block_bytes: 0 = 4 MiB; the tests shrink it so that every record crosses a block boundary
    // range_begin / range_end: a plain file's bytes [range_begin, range_end) read as if they were the whole file (process_dataset
    // parses stretches of one file on two threads)
    explicit SeqReader(const char *path, size_t block_bytes = 0, u64 range_begin = 0, u64 range_end = ~0ULL);
    ~SeqReader();
    SeqReader(const SeqReader &) = delete;
    SeqReader &operator=(const SeqReader &) = delete;
    // >= 0 sequence length; -1 EOF; -2 truncated quality.  rec's views stay valid as long as `owner` does.
    int read(bseq1_t &rec, ReadChunk &owner);
    // same with an internal owner: views valid until the next call
    int read(bseq1_t &rec) { own_.clear(); return read(rec, own_); }
    // bseq_read's single-file loop: appends records (names trimmed) to out until `size` (bases so far) reaches chunk_size on an
    // even record count, the stream ends, or the next record is a truncated one (which read() then reports)
    void fill(long chunk_size, ReadChunk &out, long &size, size_t max_records = 0);   // (max_records: also stop at that many records)
    int last_status() const;          // -2 once a truncated record has been reported, else -1 at the end of the stream, 0 before
    double seconds_blocked() const;   // time read()/fill() spent waiting for the file-reading threads (plain files)
private:
    friend class ChunkSource;
    // a reader over text blocks that are already in memory, then over whatever `more` still hands out (ChunkSource: the stretches
    // of a blocked-gzip input, parsed side by side)
    SeqReader(std::deque<std::shared_ptr<TextBlock>> blocks, std::function<std::shared_ptr<TextBlock>()> more);
    std::shared_ptr<TextBlock> take_block();      // the input's next raw text block, in order (nullptr at the end)
    size_t raw_block_bytes() const;               // the size of those blocks
    bool is_bgzf() const;
    struct Impl;
    std::unique_ptr<Impl> impl_;
    ReadChunk own_;
};

// kseq_declare.h:112-145: read until >= chunk_size bases (and an even record count); mates interleaved.
// Trailing "/[0-9]" is trimmed from names (trim_readno :106-110).  Returns number of records appended.
int bseq_read(int chunk_size, SeqReader &r1, SeqReader *r2, ReadChunk &out);

// BGZF input inflated on a GPU (bns_inflate_members: one member per lane, csrc/bns_inflate.hpp) instead of on CPU threads: readers
// opened after this call hand batches of members to `device` beside their CPU inflaters; -1 (the default): CPU only.  `bonsai classify`
// turns it on for its first device on hosts with fewer than 12 usable CPUs (BNS_BGZF_GPU=0 / 1 decides by hand); `bonsai pack` and
// library users without a GPU never see it.
void set_bgzf_device(int device);
int bgzf_device();
bool is_bgzf_file(const char *path);            // a gzip file whose first member carries the 'BC' size subfield

// CPUs this process may really use: the affinity mask cut by the cgroup v2 CPU quota (a container can show 256 CPUs under
// a 16-CPU quota).  `-p -1` means this many.
int usable_cpus();

// Narrow this thread's CPU affinity (inherited by the threads it starts) to the CPUs next to the given devices -- the union of
// their /sys/bus/pci/devices/<address>/local_cpulist, cut to the current mask.  On a two-socket host the reader, packer and
// formatter otherwise land on either socket: text parsed on one and packed on the other crosses the socket link twice (measured:
// reader 0.53 -> 0.38 s, formatter 0.33 -> 0.23 s per 16 M reads).  Returns the number of CPUs left, 0 when nothing was changed.
int bind_near_devices(const std::vector<int> &devices);

// ---- classifier ------------------------------------------------------------------------------------------
enum output_format : int { KRAKEN = 1, FASTQ = 2, EMIT_ALL = 4 };   // classifier.h:24-28

// A growable page-locked host buffer (bns_host_alloc); falls back to ordinary memory if pinning fails.
struct PinnedBuf {
    bns_ctx *ctx = nullptr;
    char *p = nullptr;
    size_t cap = 0;
    bool pinned = false, mapped = false;
    char *reserve(bns_ctx *c, size_t bytes);
    // the same from memory of our own -- anonymous pages, huge ones where the kernel gives them, touched here (on the caller's thread,
    // outside the runtime's lock) and then registered: 2 ms per 96 MiB instead of 15-45 (bns_host_register); for buffers whose copies need
    // not run at the last 15 % of the link rate
    char *reserve_registered(bns_ctx *c, size_t bytes);
    void release();
    ~PinnedBuf();
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
};

// An array in page-locked memory: what the GPU call copies to and from (into pageable memory the runtime stages the copy through
// its own bounce buffer at a fraction of the link rate: 17 ns per read of the CLI's 22 were that).
template <typename T>
struct PinArr {
    PinnedBuf buf;
    size_t n = 0;
    T *resize(bns_ctx *ctx, size_t count) { buf.reserve(ctx, count * sizeof(T) + 8); n = count; return data(); }
    T *data() { return reinterpret_cast<T *>(buf.p); }
    const T *data() const { return reinterpret_cast<const T *>(buf.p); }
    T &operator[](size_t i) { return data()[i]; }
    const T &operator[](size_t i) const { return data()[i]; }
    size_t size() const { return n; }
    void release() { buf.release(); n = 0; }
};

// What the GPU call leaves for the formatter: per-unit results and, when the output prints them, the hit runs.  (Holds page-locked
// memory of the classifier's context: release it, or let it go out of scope, before the classifier does.)
struct ChunkResult {
    unsigned n = 0;
    int is_paired = 0;
    bool want_runs = false;
    bool taxon_only = false;     // nothing prints missing / ambig / hit counts (-K -F): the call brings back the taxon alone
    PinArr<u32> taxon, missing, ambig, n_hits, n_runs;
    PinArr<u64> run_start;
    std::vector<u32> run_tax, run_len;
    // the packed input of the GPU call (kept with the result: a chunk is packed on one thread while the previous one is on the GPU)
    PinnedBuf words;
    PinArr<u64> offsets;
    u64 n_bad = 0;
    void release() { taxon.release(); missing.release(); ambig.release(); n_hits.release(); n_runs.release(); run_start.release(); words.release(); offsets.release(); }
    // scratch of the GPU call (kept with the result so that it is recycled with it): where the chunk's sequences lie, and the
    // packer's list of words that hold a base other than A/C/G/T
    std::vector<const char *> seq_ptrs;
    std::vector<u32> seq_lens, bad_mask;
    std::vector<u64> bad_word;
    double t_pack = 0, t_call = 0, t_copy = 0;     // stage seconds of this chunk's GPU call (BNS_CLI_TIMING)
};

struct ClassifierGeneric {
    bns_ctx *ctx_ = nullptr;                 // == ctxs_[0]
    // one context per device (north_star / SURVEY 8e: reads shard, the db is replicated).  process_dataset's chunk is split
    // into contiguous unit ranges, one per device (mates stay together), classified concurrently, and the results are
    // concatenated in input order -- no exchange step inside classification.
    std::vector<bns_ctx *> ctxs_;
    std::vector<int> devices_;               // ctxs_[i] lives on device devices_[i]
    unsigned k_ = 0, c_ = 0;
    u32 output_flag_ = 0;
    int nt_ = 1;
    u64 classified_[2] = {0, 0};
    // per-chunk work buffers, kept between calls (a fresh 70 MB vector per chunk is mostly page faults)
    struct Shard { ChunkResult res; };                                                  // per extra device (devices 1..)
    std::vector<std::unique_ptr<Shard>> shards_;
    struct Work {
        struct alignas(128) Part {                                                     // one formatting thread's output
            char *p = nullptr; size_t n = 0, cap = 0;                                  // Kraken lines: a raw buffer (no zero-fill, no per-record resize)
            std::string s;                                                             // fastq-style records
            void ensure(size_t want) { if (want > cap) { char *q = static_cast<char *>(std::realloc(p, want)); if (!q) throw std::bad_alloc(); p = q; cap = want; } }
            Part() = default;
            Part(Part &&o) noexcept : p(o.p), n(o.n), cap(o.cap), s(std::move(o.s)) { o.p = nullptr; o.n = o.cap = 0; }
            Part(const Part &) = delete;
            Part &operator=(const Part &) = delete;
            ~Part() { std::free(p); }
        };
        std::vector<Part> parts;
        ChunkResult res, first;                                                        // classify_seqs' own result buffers (first: device 0's part of a split chunk)
        double t_assemble = 0, t_gpu = 0, t_format = 0, t_wait = 0, t_write = 0, t_pack = 0, t_call = 0, t_copy = 0;       // stage seconds (BNS_CLI_TIMING=1 prints them)
    } work_;
    // mirrors classifier.h:155-166: (db, spaces, k, wsz, num_threads, emit_all, emit_fastq, emit_kraken, canonicalize)
    ClassifierGeneric(const Database &db, const std::vector<u32> &parent, int device = 0, int num_threads = 1,
                      bool emit_all = true, bool emit_fastq = true, bool emit_kraken = false, bool canonicalize = true,
                      int layout = BNS_LAYOUT_MINBUCKET);
    // several devices (`bonsai classify -g 0-7`): the db is uploaded once and RCCL-broadcast (bns_load_table_multi)
    ClassifierGeneric(const Database &db, const std::vector<u32> &parent, const std::vector<int> &devices, int num_threads = 1,
                      bool emit_all = true, bool emit_fastq = true, bool emit_kraken = false, bool canonicalize = true,
                      int layout = BNS_LAYOUT_MINBUCKET);
    ~ClassifierGeneric();
    ClassifierGeneric(const ClassifierGeneric &) = delete;
    ClassifierGeneric &operator=(const ClassifierGeneric &) = delete;
    int get_emit_all() const { return output_flag_ & EMIT_ALL; }
    int get_emit_kraken() const { return output_flag_ & KRAKEN; }
    int get_emit_fastq() const { return output_flag_ & FASTQ; }
    std::FILE *taxon_out_ = nullptr;         // `bonsai classify -b`: the taxon of every unit, in input order, as raw little-endian u32
    bool nseq_printed_ = false;              // process_dataset's "nseq:" line on stderr has been printed (by the device text path or the host one)
    u64 n_classified() const { return classified_[0]; }
    u64 n_unclassified() const { return classified_[1]; }
};
using Classifier = ClassifierGeneric;

// classifier.h:112-129 / 72-108 / 45-61: byte-for-byte formatters.  The hit stream comes either as the reference's `taxa`
// vector or already run-length encoded (bns_classify_batch_runs): same text.
struct HitRuns { const u32 *tax; const u32 *len; u32 n; };
void append_kraken_classification(const HitRuns &runs, tax_t taxon, u32 ambig_count, u32 missing_count,
                                  const bseq1_t &bs, std::string &bks);
void append_fastq_classification(const HitRuns &runs, tax_t taxon, u32 ambig_count, u32 missing_count,
                                 const bseq1_t *bs, std::string &bks, int verbose, int is_paired);
void append_kraken_classification(const std::vector<tax_t> &taxa, tax_t taxon, u32 ambig_count, u32 missing_count,
                                  const bseq1_t &bs, std::string &bks);
void append_fastq_classification(const std::vector<tax_t> &taxa, tax_t taxon, u32 ambig_count, u32 missing_count,
                                 const bseq1_t *bs, std::string &bks, int verbose, int is_paired);

// classifier.h:269-287: classify bs[0..n) (mates adjacent when is_paired) and append the result text to cks.
void classify_seqs(ClassifierGeneric &c, bseq1_t *bs, std::string &cks, unsigned n, int is_paired);
// its two halves, which process_dataset runs on different threads (GPU call of chunk i+1 || text of chunk i)
void classify_chunk(ClassifierGeneric &c, const bseq1_t *bs, unsigned n, int is_paired, ChunkResult &r);
void format_chunk(ClassifierGeneric &c, const bseq1_t *bs, const ChunkResult &r, std::string &cks);
// text left in (*into)[0 .. result) (into == nullptr: c.work_.parts)
unsigned format_chunk_parts(ClassifierGeneric &c, const bseq1_t *bs, const ChunkResult &r, std::vector<ClassifierGeneric::Work::Part> *into = nullptr,
                            unsigned skip_first = 0);   // skip_first: leave out (text and tally) the chunk's first units

// "0-3", "0,2,5", "all" (every visible device) -> device list; throws bns::Error on anything else
std::vector<int> parse_devices(const char *spec);
// classifier.h:296-337
// bseq_read chunks of one file (or two, mates interleaved) in input order.  With parser_threads > 1 one plain (not gzip, not piped)
// file is parsed in stretches of segment_bytes (0: ~4 chunks) on that many threads -- find_cut_points says where a stretch may begin
// -- and handed out in file order; the records and their order are those of one sequential parse (a chunk never spans two
// stretches, so chunk boundaries differ).  cuts_override: the cut offsets to use instead (tests).
class ChunkSource {
public:
    // range_begin: one plain file read from that offset on (a record boundary) as if it were the file's start
    ChunkSource(const char *fq1, const char *fq2, unsigned chunk_size, unsigned parser_threads = 1, u64 segment_bytes = 0,
                const std::vector<u64> *cuts_override = nullptr, u64 range_begin = 0);
    ~ChunkSource();
    std::unique_ptr<ReadChunk> next();                 // nullptr at the end of the input
    void recycle(std::unique_ptr<ReadChunk> c);        // a chunk the caller is done with (its memory is used again)
    size_t stretches() const;
    bool fell_back() const;                            // a stretch did not end between two records: the rest was parsed sequentially
    double parse_seconds() const, blocked_seconds() const;
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// parser_threads > 1: one plain (not piped) file is parsed in stretches of segment_bytes (0: ~4 chunks) on that many threads, a
// BGZF file in stretches of its inflated text blocks; results and their order do not depend on it.
void process_dataset(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, unsigned chunk_size, unsigned parser_threads = 1,
                     u64 segment_bytes = 0);
std::vector<u64> find_cut_points(const char *path, u64 seg_bytes);

// ---- pre-packed read container (SURVEY 8f-2 "pre-packed input format") ---------------------------------------------------------
// `bonsai pack` writes the reads of a FASTA / FASTQ input (plain, gzip, or a pair of files) as the 2-bit image the GPU call takes
// -- bns_pack_reads' words, the sparse invalid-base list, the read lengths -- plus the read names, chunk by chunk; `bonsai
// classify` recognises the file by its magic and hands every chunk straight to bns_classify_batch_packed*: no parser, no packer,
// 40 bytes per 150-bp read off the disk.  Results are those of the original input (Kraken lines byte for byte; FASTQ-style
// output needs the bases and qualities and is refused).  Little-endian:
//   file header  32 B: "BNSPACK\1" | u32 version (1) | u32 flags (bit 0: mates interleaved, bit 1: names present) | 16 B reserved
//   chunk header 64 B: u32 'CHNK' | u32 n_reads | u64 total_bases | u64 n_words | u64 n_bad | u64 names_bytes | u64 payload_bytes | 16 B reserved
//   payload          : u32 lens[n_reads] (+pad to 8) | u64 words[n_words] | u64 bad_word[n_bad] | u32 bad_mask[n_bad] (+pad to 8) |
//                      names: n_reads NUL-terminated strings (+pad to 8)
constexpr char PACK_MAGIC[8] = {'B', 'N', 'S', 'P', 'A', 'C', 'K', 1};
constexpr u32 PACK_CHUNK_MAGIC = 0x4B4E4843u;   // "CHNK"
struct PackFileHeader { char magic[8]; u32 version, flags; u64 reserved[2]; };
struct PackChunkHeader { u32 magic, n_reads; u64 total_bases, n_words, n_bad, names_bytes, payload_bytes, reserved[2]; };
static_assert(sizeof(PackFileHeader) == 32 && sizeof(PackChunkHeader) == 64, "container headers are fixed-size");
bool is_pack_container(const char *path);
// reads -> container; chunk_bases = bases per chunk (0: 2^27: a chunk is one GPU call); returns (reads, bases) written
std::pair<u64, u64> pack_dataset(const char *fq1, const char *fq2, const char *out_path, unsigned chunk_bases, unsigned parser_threads, int threads,
                                 bool with_names = true);

// ---- db construction -------------------------------------------------------------------------
// build_name_hash (util.h:693-722): "name<TAB>taxid" per line, later lines win
std::vector<std::pair<std::string, tax_t>> build_name_hash(const char *seq2tax_path);
// get_taxid (util.h:898-929): name = field between the last two '|' of the first header line, or its first token;
// unknown names map to 1
std::string genome_name(const std::string &first_header_line_without_gt);
tax_t get_taxid(const char *genome_path, const std::vector<std::pair<std::string, tax_t>> &sorted_names);

struct BuildOptions {
    unsigned k = 31;
    int wsz = -1;                // bin/bonsai.cpp:170,217: < k means k
    spvec_t spacing;             // k-1 extra gaps (empty = contiguous)
    bool canon = true;
    bool entropy = false;        // -e: score::Entropy (path-overload rule) instead of score::Lex
    int device = 0;
};
// lca_map (feature_min.h:178-183 -> make_map :93-171 -> update_lca_map :205-228) on the GPU: every k-mer (or windowed
// minimizer) of every sequence of genome g maps to get_taxid(g); shared ones to the lca.  Returns a Database whose
// khash arrays are valid for kh_get and sized like khash would size them (smallest power of two with load <= 0.77).
Database lca_map(const std::vector<std::string> &paths, const std::vector<u32> &parent, const char *seq2tax_path,
                 const BuildOptions &opt);

// ---- Encoder API surface (encoder.h:415-442, 448-530) -----------------------------------------------------
// Synchronous, in sequence order, on the calling thread -- like the reference; the k-mers come from the GPU encoder.
// Sequences of a FILE are read by SeqReader (kseq_read's records: klib/kseq.h:177-225) and handed to the device in batches of
// whole records; the functor still sees one value at a time, record after record.
class Encoder {
public:
    // (k, gaps, w) is the reference's Spacer(k, w, spaces) (spacer.h:58-71), `score` its ScoreType template argument
    // (BNS_SCORE_LEX = Encoder<score::Lex>, BNS_SCORE_ENTROPY_STRING / _PATH = Encoder<score::Entropy>: which of the two rules runs
    // is decided per call, as in the reference: the string overload computes the string rule, the path overloads the path rule,
    // SURVEY F8); w <= comb size = unwindowed
    Encoder(unsigned k, const spvec_t &gaps = {}, bool canonicalize = true, int device = 0, unsigned w = 0, int score = 0);
    ~Encoder();
    Encoder(const Encoder &) = delete;
    Encoder &operator=(const Encoder &) = delete;
    // encoder.h:415-442
    template <typename Functor>
    void for_each(const Functor &func, const char *str, u64 l)
    {
        fetch(str, l);
        for (u64 km : kmers_) func(km);
    }
    // for_each_hash(func, str, l, k = 0) (encoder.h:355-394): the ntHash stream; a spaced or windowed Spacer is an
    // UNRECOVERABLE_ERROR there, a bns::Error here.  table256 = seeds in make_nthash_lut's geometry, nullptr = ntHash's own.
    template <typename Functor>
    void for_each_hash(const Functor &func, const char *str, u64 l, unsigned k = 0, const u64 *table256 = nullptr)
    {
        fetch_hash(str, l, k, table256);
        for (u64 h : kmers_) func(h);
    }
    // for_each(func, path, kseq_t * = nullptr) (encoder.h:511-530; std::string form :507-509): every record of a FASTA / FASTQ
    // file -- plain or gzip, or .xz / .bz2 / .zst through `xz|bzip2|zstd -dc` as there --, dispatched as the reference's path
    // overloads dispatch (encoder.h:448-463 through for_each_canon / for_each_uncanon): a spaced seed goes through
    // for_each_uncanon_spaced (the string overload emits nothing), Encoder<score::Entropy> scores by the path rule
    // (F8).  The kseq_t * argument of the reference (a caller-owned parse buffer) is accepted and ignored: the reader is SeqReader.
    // Could not open the file: UNRECOVERABLE_ERROR there, bns::Error here.
    template <typename Functor>
    void for_each(const Functor &func, const char *path) { each_path(path, PATH_AUTO, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each(const Functor &func, const char *path, KS *) { for_each(func, path); }
    template <typename Functor>
    void for_each(const Functor &func, const std::string &path) { for_each(func, path.c_str()); }
    // for_each_canon / for_each_uncanon(func, path) (encoder.h:479-494): the dispatch of one side whatever canonicalize() says
    template <typename Functor>
    void for_each_canon(const Functor &func, const char *path) { each_path(path, PATH_CANON, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each_canon(const Functor &func, const char *path, KS *) { for_each_canon(func, path); }
    template <typename Functor>
    void for_each_uncanon(const Functor &func, const char *path) { each_path(path, PATH_UNCANON, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each_uncanon(const Functor &func, const char *path, KS *) { for_each_uncanon(func, path); }
    // for_each_hash(func, path, kseq_t * = nullptr) (encoder.h:408-414): the ntHash stream of every record
    template <typename Functor>
    void for_each_hash(const Functor &func, const char *path) { each_path(path, PATH_HASH, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each_hash(const Functor &func, const char *path, KS *) { for_each_hash(func, path); }
    // for_each(func, container of paths) (encoder.h:531-540)
    template <typename Functor>
    void for_each(const Functor &func, const std::vector<std::string> &paths) { for (const auto &p : paths) for_each(func, p.c_str()); }
    // python/bns.cpp:112-129 from_str equivalent
    const std::vector<u64> &from_str(const char *str, u64 l) { fetch(str, l); return kmers_; }
    bool canonicalize() const { return canon_; }
    unsigned k() const { return k_; }
private:
    enum PathMode { PATH_AUTO, PATH_CANON, PATH_UNCANON, PATH_HASH };
    using Sink = std::function<void(const u64 *, size_t)>;
    template <typename Functor>
    static Sink sink_of(const Functor &func) { return [&func](const u64 *v, size_t n) { for (size_t i = 0; i < n; ++i) func(v[i]); }; }
    void configure(bool path_rules, bool canon);
    void each_path(const char *path, PathMode mode, const Sink &sink);
    void fetch(const char *str, u64 l);
    void fetch_hash(const char *str, u64 l, unsigned k, const u64 *table256);
    bns_ctx *ctx_ = nullptr;
    unsigned k_, w_;
    int score_;
    bool canon_, spaced_;
    spvec_t gaps_;
    int configured_ = -1;                       // (path_rules << 1 | canon) the context is set up for
    std::vector<u64> kmers_;
};

// ---- RollingHasher API surface (encoder.h:644-865) ---------------------------------------------------------
// RollingHasher<IntType, CyclicHash<IntType>> for IntType = u64 and unsigned __int128 (the instantiations the reference uses:
// python/bns.cpp:42-86, bin/setsketcher.cpp:18-27, test/encoding.cpp:152) over bns_rolling_hash*_batch.  Same constructor
// arguments and defaults; `enc` must be DNA (0) -- protein alphabets are outside SURVEY 8.  The character tables are the restated
// generator's for (seed1, seed2) (bns_rolling_tables*, parity unpinned: SURVEY F10).
namespace detail {
class RollingCore {                              // the part that does not depend on the word type
public:
    RollingCore(unsigned bits, unsigned k, bool canon, int enc, long long wsz, u64 seed1, u64 seed2, int device);
    ~RollingCore();
    RollingCore(const RollingCore &) = delete;
    RollingCore &operator=(const RollingCore &) = delete;
    using Sink = std::function<void(const u64 *, size_t)>;     // u64: values; u128: (lo, hi) pairs, n = number of VALUES
    void each_str(const char *s, size_t l, bool canon, const Sink &sink);
    void each_path(const char *path, bool canon, const Sink &sink);
    void window(long long w) { w_ = w <= (long long)k_ ? -1 : w; }
    long long window() const { return w_; }
    unsigned bits_, k_;
    bool canon_;
    long long w_ = -1;
    u64 seed1_, seed2_;
private:
    void run(const char *bases, const u64 *offsets, u64 n, bool canon, const Sink &sink);
    bns_ctx *ctx_ = nullptr;
    std::vector<u64> fwd_, rc_, out_;
    std::vector<u32> cnt_;
};
}  // namespace detail

enum InputType : int { DNA = 0 };                // rhtraits.h's InputType, the member this path supports

template <typename IntType>
class RollingHasher {
    static_assert(sizeof(IntType) == 8 || sizeof(IntType) == 16, "RollingHasher<u64> or RollingHasher<unsigned __int128>");
public:
    RollingHasher(unsigned k = 21, bool canon = false, InputType enc = DNA, long long wsz = -1, u64 seed1 = 1337, u64 seed2 = 137,
                  int device = 0)
        : core_(sizeof(IntType) * 8, k, canon, (int)enc, wsz, seed1, seed2, device) {}
    long long window() const { return core_.window(); }
    void window(long long w) { core_.window(w); }
    InputType hashtype() const { return DNA; }
    bool canonicalize() const { return core_.canon_; }
    void canonicalize(bool v) { core_.canon_ = v; }
    void reset() {}                               // (no state survives a call: every call starts its hashers afresh, as reset() would)
    // encoder.h:810-814, 692-796
    template <typename Functor>
    void for_each_hash(const Functor &func, const char *s, size_t l) { core_.each_str(s, l, core_.canon_, sink_of(func)); }
    template <typename Functor>
    void for_each_canon(const Functor &func, const char *s, size_t l) { core_.each_str(s, l, true, sink_of(func)); }
    template <typename Functor>
    void for_each_uncanon(const Functor &func, const char *s, size_t l) { core_.each_str(s, l, false, sink_of(func)); }
    // encoder.h:821-839 (path; .xz / .bz2 / .zst through their decompressors), :841-856
    template <typename Functor>
    void for_each_hash(const Functor &func, const char *path) { core_.each_path(path, core_.canon_, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each_hash(const Functor &func, const char *path, KS *) { for_each_hash(func, path); }
    template <typename Functor>
    void for_each_canon(const Functor &func, const char *path) { core_.each_path(path, true, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each_canon(const Functor &func, const char *path, KS *) { for_each_canon(func, path); }
    template <typename Functor>
    void for_each_uncanon(const Functor &func, const char *path) { core_.each_path(path, false, sink_of(func)); }
    template <typename Functor, typename KS>
    void for_each_uncanon(const Functor &func, const char *path, KS *) { for_each_uncanon(func, path); }
    // encoder.h:857-860
    template <typename... Args>
    void for_each(Args &&...args) { for_each_hash(std::forward<Args>(args)...); }
private:
    template <typename Functor>
    static detail::RollingCore::Sink sink_of(const Functor &func)
    {
        return [&func](const u64 *v, size_t n) {
            if constexpr (sizeof(IntType) == 8) { for (size_t i = 0; i < n; ++i) func((IntType)v[i]); }
            else { for (size_t i = 0; i < n; ++i) func((IntType)(((unsigned __int128)v[2 * i + 1] << 64) | v[2 * i])); }
        };
    }
    detail::RollingCore core_;
};

}  // namespace bns
