// bns_host.cpp -- host side of the classify path (see bns_host.hpp for the reference map).
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

namespace {

[[noreturn]] void die(const std::string &msg) { throw Error(msg); }

void chk(bns_ctx *ctx, int rc, const char *what)
{
    if (rc == BNS_OK) return;
    std::string m = std::string(what) + ": " + bns_strerror(rc);
    if (ctx) { m += " ("; m += bns_last_error(ctx); m += ")"; }
    die(m);
}

bool ends_with(const std::string &s, const char *suf)
{
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

bool gz_read_all(gzFile fp, void *dst, u64 n)
{
    char *d = static_cast<char *>(dst);
    while (n) {
        const unsigned want = n > (1u << 30) ? (1u << 30) : (unsigned)n;
        const int got = gzread(fp, d, want);
        if (got <= 0) return false;
        d += got; n -= (u64)got;
    }
    return true;
}

void put_unsigned(std::string &s, u32 x)            // kspp/ks.h:337-354 putuw_
{
    char tmp[12]; int n = 0;
    if (x == 0) tmp[n++] = '0';
    while (x) { tmp[n++] = char('0' + x % 10); x /= 10; }
    while (n) s.push_back(tmp[--n]);
}

void put_signed(std::string &s, long c)             // kspp/ks.h:318-336 / 355-373 putw_ / putl_
{
    if (c < 0) { s.push_back('-'); put_unsigned(s, (u32)(-c)); }
    else put_unsigned(s, (u32)c);
}

void append_counts(u32 count, char ch, std::string &s)      // classifier.h:63-70
{
    if (!count) return;
    s.push_back(ch); s.push_back(':'); put_unsigned(s, count); s.push_back('\t');
}

// classifier.h:45-61 (+30-42): "taxid:count" per run of equal consecutive hits, 'U' for taxid 0, 'A' for (tax_t)-1; the
// hit stream arrives run-length encoded (bns_classify_batch_runs, or OwnedRuns over a `taxa` vector)
void append_taxa_runs(tax_t taxon, const u32 *run_tax, const u32 *run_len, u32 n_runs, std::string &s)
{
    if (!taxon) { s += "0:0\n"; return; }
    for (u32 i = 0; i < n_runs; ++i) {
        if (run_tax[i] == 0) s.push_back('U');
        else if (run_tax[i] == (tax_t)-1) s.push_back('A');
        else put_unsigned(s, run_tax[i]);
        s.push_back(':'); put_unsigned(s, run_len[i]); s.push_back('\t');
    }
    s.back() = '\n';
}

}  // namespace

// ---------------------------------------------------------------------------------------------- spacing
spvec_t parse_spacing(const char *ss, unsigned k)
{
    if (!ss || !*ss) return spvec_t(k ? k - 1 : 0, 0);
    spvec_t ret;
    const char *p = ss;
    while (*p) {
        char *e;
        const unsigned long j = std::strtoul(p, &e, 10);
        ret.push_back((u16)j);
        p = e;
        if (*p == 'x') {
            const long rep = (long)std::strtoul(p + 1, &e, 10) - 1;
            for (long r = 0; r < rep; ++r) ret.push_back((u16)j);
            p = e;
        }
        const char *comma = std::strchr(p, ',');
        if (!comma) break;
        p = comma + 1;
    }
    return ret;
}

// ---------------------------------------------------------------------------------------------- bns.db
Database::Database(const char *path)
{
    // database.h:33-56.  The reference reader expects u8 spacing entries (:46-48) while its gz writer emits
    // u16 (:89); the file does not say which, so try both and keep the one whose khash header is
    // self-consistent and whose payload ends exactly at EOF.
    for (int width = 1; width <= 2; ++width) {
        gzFile fp = gzopen(path, "rb");
        if (!fp) die(std::string("Could not open ") + path + " for reading.");
        u32 k = 0, w = 0;
        bool ok = gz_read_all(fp, &k, 4) && gz_read_all(fp, &w, 4) && k >= 1 && k <= 32;
        spvec_t sp(ok ? k - 1 : 0);
        for (u32 i = 0; ok && i + 1 < k; ++i) {
            if (width == 1) { u8 b; ok = gz_read_all(fp, &b, 1); sp[i] = b; }
            else ok = gz_read_all(fp, &sp[i], 2);
        }
        u64 hdr[4];
        ok = ok && gz_read_all(fp, hdr, sizeof(hdr));
        if (ok) {
            const u64 nb = hdr[0];
            ok = nb && !(nb & (nb - 1)) && hdr[2] <= hdr[1] && hdr[1] <= nb && hdr[3] == (u64)(nb * 0.77 + 0.5);
            if (ok) {
                KhashC t;
                t.n_buckets = nb; t.n_occupied = hdr[1]; t.size = hdr[2]; t.upper_bound = hdr[3];
                t.flags.resize(nb < 16 ? 1 : nb >> 4); t.keys.resize(nb); t.vals.resize(nb);
                char extra;
                ok = gz_read_all(fp, t.flags.data(), t.flags.size() * 4) && gz_read_all(fp, t.keys.data(), nb * 8) &&
                     gz_read_all(fp, t.vals.data(), nb * 4) && gzread(fp, &extra, 1) != 1;
                if (ok) { k_ = k; w_ = w; s_ = sp; db_ = std::move(t); spacing_width_ = width; }
            }
        }
        gzclose(fp);
        if (ok) return;
    }
    die(std::string("Error: Could not read a bns.db database from ") + path);
}

void Database::write(const char *path, int spacing_width) const
{
    gzFile fp = gzopen(path, ends_with(path, ".gz") ? "wb" : "wbT");
    if (!fp) die(std::string("Could not open ") + path + " for writing.");
    auto put = [&](const void *p, u64 n) {
        const char *c = static_cast<const char *>(p);
        while (n) {
            const unsigned chunk = n > (1u << 30) ? (1u << 30) : (unsigned)n;
            if (gzwrite(fp, c, chunk) != (int)chunk) { gzclose(fp); die("Error writing database"); }
            c += chunk; n -= chunk;
        }
    };
    const u32 k = k_, w = w_;
    put(&k, 4); put(&w, 4);
    for (u32 i = 0; i + 1 < k_; ++i) {
        const u16 g = i < s_.size() ? s_[i] : 0;
        if (spacing_width == 1) { const u8 b = (u8)g; put(&b, 1); } else put(&g, 2);
    }
    const u64 hdr[4] = {db_.n_buckets, db_.n_occupied, db_.size, db_.upper_bound};
    put(hdr, sizeof(hdr));
    // empty / deleted slots are written as zeros (util.h:282-284) -- streamed through one 4 Mi-slot scratch block per array
    // rather than through full copies of keys and vals (12 bytes per bucket: ~100 GB for a 2^33-bucket table)
    put(db_.flags.data(), db_.flags.size() * 4);
    const u64 BLK = 1ull << 22;
    {
        std::vector<u64> kb(std::min<u64>(BLK, db_.n_buckets));
        for (u64 i0 = 0; i0 < db_.n_buckets; i0 += BLK) {
            const u64 n = std::min<u64>(BLK, db_.n_buckets - i0);
            for (u64 j = 0; j < n; ++j) kb[j] = db_.exists(i0 + j) ? db_.keys[i0 + j] : 0;
            put(kb.data(), n * 8);
        }
    }
    {
        std::vector<u32> vb(std::min<u64>(BLK, db_.n_buckets));
        for (u64 i0 = 0; i0 < db_.n_buckets; i0 += BLK) {
            const u64 n = std::min<u64>(BLK, db_.n_buckets - i0);
            for (u64 j = 0; j < n; ++j) vb[j] = db_.exists(i0 + j) ? db_.vals[i0 + j] : 0;
            put(vb.data(), n * 4);
        }
    }
    gzclose(fp);
}

// ---------------------------------------------------------------------------------------------- taxonomy
std::vector<u32> build_parent_map(const char *fn)
{
    std::ifstream is(fn);
    if (!is) die(std::string("Failed to create taxmap from ") + fn);
    std::vector<u32> parent;
    auto reserve = [&](u32 id) {
        if (id >= (1u << 28)) die("taxid >= 2^28 is not supported by the flat parent array");
        if (id >= parent.size()) parent.resize((size_t)id + 1, BNS_TAX_ABSENT);
    };
    std::string line;
    size_t n_keys = 0;
    while (std::getline(is, line)) {
        if (line.empty() || line[0] == '#') continue;                       // util.h:775
        const u32 child = (u32)std::atoi(line.c_str());
        const char *bar = std::strchr(line.c_str(), '|');
        if (!bar) die("Malformed line in " + std::string(fn) + ": " + line); // the reference warns and stores -1 (util.h:776-778)
        const u32 par = (u32)std::atoi(bar + 2);
        reserve(child);
        if (par != BNS_TAX_ABSENT) reserve(par);
        else die("Malformed parent in " + std::string(fn) + ": " + line);
        if (parent[child] == BNS_TAX_ABSENT) ++n_keys;
        parent[child] = par;
    }
    reserve(1);
    if (parent[1] == BNS_TAX_ABSENT) ++n_keys;
    parent[1] = 0;                                                           // util.h:780-781
    if (n_keys < 2) die(std::string("Failed to create taxmap from ") + fn);  // util.h:782
    return parent;
}

// ---------------------------------------------------------------------------------------------- FASTA/FASTQ
namespace {
constexpr size_t RAW_BLOCK = 4u << 20;

inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace() in the C locale
}  // namespace

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
static BlockPool &block_pool() { static BlockPool *p = new BlockPool; return *p; }   // (leaked on purpose: blocks may outlive static destruction order)

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

// ---- BGZF (blocked gzip, what bgzip / htslib and many sequencing pipelines write): every gzip member is at most 64 KiB of text and
// carries its own compressed size in a 'BC' extra subfield, so members can be found without inflating and inflated side by side.
// (One plain gzip stream cannot: DEFLATE has no sync points -- that input keeps its one inflate thread, decoupled from the parser.)
namespace {
double tnow();                                                   // (defined with the pipeline's other helpers, below)
// raw-DEFLATE decoder for one member: libdeflate when the system has it (dlopen -- ~3x zlib's inflate), else zlib
struct LibDeflate {
    void *lib = nullptr;
    void *(*alloc)() = nullptr;
    int (*dec)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*free_)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
};
const LibDeflate *libdeflate()
{
    static const LibDeflate d = [] {
        LibDeflate x;
        if (std::getenv("BNS_NO_LIBDEFLATE")) return x;
        x.lib = ::dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!x.lib) return x;
        x.alloc = reinterpret_cast<void *(*)()>(::dlsym(x.lib, "libdeflate_alloc_decompressor"));
        x.dec = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(::dlsym(x.lib, "libdeflate_deflate_decompress"));
        x.free_ = reinterpret_cast<void (*)(void *)>(::dlsym(x.lib, "libdeflate_free_decompressor"));
        x.crc = reinterpret_cast<uint32_t (*)(uint32_t, const void *, size_t)>(::dlsym(x.lib, "libdeflate_crc32"));
        if (!x.alloc || !x.dec || !x.free_ || !x.crc) x.lib = nullptr;
        return x;
    }();
    return d.lib ? &d : nullptr;
}
struct MemberInflater {
    const LibDeflate *ld = libdeflate();
    void *dctx = nullptr;
    z_stream zs{};
    bool z_init = false;
    MemberInflater() { if (ld) dctx = ld->alloc(); if (!dctx) ld = nullptr; }
    ~MemberInflater() { if (dctx) ld->free_(dctx); if (z_init) inflateEnd(&zs); }
    // in: the member's deflate payload; out: exactly out_n bytes expected; crc_want: the member's CRC32 field
    bool run(const unsigned char *in, size_t in_n, char *out, size_t out_n, uint32_t crc_want)
    {
        if (ld) {
            size_t got = 0;
            if (ld->dec(dctx, in, in_n, out, out_n, &got) != 0 || got != out_n) return false;
            return ld->crc(0, out, out_n) == crc_want;
        }
        if (!z_init) { if (inflateInit2(&zs, -15) != Z_OK) return false; z_init = true; }
        else inflateReset(&zs);
        zs.next_in = const_cast<unsigned char *>(in); zs.avail_in = (uInt)in_n;
        zs.next_out = reinterpret_cast<unsigned char *>(out); zs.avail_out = (uInt)out_n;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.avail_out != 0) return false;
        return (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const unsigned char *>(out), (uInt)out_n) == crc_want;
    }
};
// the member that starts at p (n bytes available): its total size from the 'BC' subfield, the offset of its deflate payload; 0 when
// p does not start a BGZF member
size_t bgzf_member(const unsigned char *p, size_t n, size_t &payload_off)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = p[10] | ((size_t)p[11] << 8);
    if (12 + xlen > n) return 0;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
        const size_t slen = p[q + 2] | ((size_t)p[q + 3] << 8);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
            payload_off = 12 + xlen;
            return (size_t)(p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
        }
        q += 4 + slen;
    }
    return 0;
}
}  // namespace

// ---- BGZF on the GPU: which device (set_bgzf_device)
namespace {
std::atomic<int> g_bgzf_device{-1};
}  // namespace
void set_bgzf_device(int device) { g_bgzf_device = device; }
bool is_bgzf_file(const char *path)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    unsigned char head[64];
    const ssize_t n = ::pread(fd, head, sizeof(head), 0);
    ::close(fd);
    size_t pay = 0;
    return n >= 18 && bgzf_member(head, (size_t)n, pay) != 0;
}
int bgzf_device() { return g_bgzf_device; }

struct SeqReader::Impl {
    using Block = TextBlock;
    static constexpr size_t HEAD = 64u << 10;
    size_t raw_block = RAW_BLOCK;   // 4 MiB
    gzFile fp = nullptr;
    int fd = -1;                  // plain files are read with read(2), not through zlib
    // producer side: raw blocks
    std::thread producer;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Block>> ready;
    bool producer_done = false, stop = false;
    // consumer side: the block being parsed
    std::shared_ptr<Block> cur;
    size_t pos = 0;
    bool final_ = false;          // no more data will arrive: what is in cur is the end of the stream
    bool at_header = false;       // cur[pos] is the '>' / '@' that starts the next record (kseq's last_char)
    const ReadChunk *reg_owner = nullptr;   // where cur was last registered (owner, its epoch, the block)
    u64 reg_epoch = 0;
    const Block *reg_block = nullptr;

    // A FAILED read is not the end of the file: it is recorded here (under mu) and pop_raw() turns it into an error, so that a
    // short input never passes as a clean one (exit 0 with part of the output).  EINTR is retried.
    std::string io_error;
    void set_io_error(const std::string &what)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (io_error.empty()) io_error = what;
    }
    size_t read_some(char *dst, size_t n)
    {
        size_t got = 0;
        while (got < n) {                                        // short counts are normal
            long r;
            if (fd >= 0) {
                r = (long)::read(fd, dst + got, n - got);
                if (r < 0 && errno == EINTR) continue;
                if (r < 0) set_io_error(std::string("read error on the input: ") + std::strerror(errno));
            } else {
                r = gzread(fp, dst + got, (unsigned)std::min<size_t>(n - got, 1u << 30));
                if (r < 0) { int ec = 0; const char *m = gzerror(fp, &ec); set_io_error(std::string("read error on the gzip input: ") + (m ? m : "?")); }
            }
            if (r <= 0) break;
            got += (size_t)r;
        }
        return got;
    }
    // Plain files: N_PRODUCERS threads pread() alternate blocks (block i = bytes [i, i + 1) * raw_block; one read(2) stream copies
    // out of the page cache at ~6 GB/s, below what one parser thread takes) and hand them over in file order; a .gz file is one
    // zlib stream and keeps one producer.
    static constexpr unsigned N_PRODUCERS = 3;
    std::vector<std::thread> producers;
    std::map<u64, std::shared_ptr<Block>> ready_at;          // plain files: finished blocks by index
    u64 next_block = 0, end_block = ~0ULL;                      // next index the consumer takes; first index past the end of the file
    u64 range_begin = 0, range_end = ~0ULL;                     // plain files: the byte range this reader covers (a whole file: 0 .. end)
    int last_rc = 0;                                            // what read() last ended on: -1 end of stream, -2 truncated record
    bool saw_truncated = false;                                 // a truncated record was reported at some point
    bool use_pread = false;                                     // (a pipe cannot be pread: one producer, read(2))
    // BGZF input: a splitter thread walks the member headers and cuts the file into tasks of consecutive members (<= raw_block of
    // text each); inflater threads turn tasks into text blocks, handed to the parser in file order through ready_at
    bool bgzf = false;
    int bfd = -1;
    struct BgzfMember { u32 in_off, in_len, out_off, out_len, crc; };
    struct BgzfTask { u64 index = 0, file_off = 0; size_t in_bytes = 0, out_bytes = 0; std::vector<BgzfMember> members; };
    std::deque<BgzfTask> btasks;
    size_t bq_cap = 64;                                         // tasks the splitter may run ahead of the inflaters
    bool split_done = false;
    // GPU inflaters (set_bgzf_device): what they spent, summed over the threads (BNS_CLI_TIMING)
    double gz_t_read = 0, gz_t_call = 0, gz_t_kernel = 0, gz_t_copy = 0;
    u64 gz_batches = 0, gz_members = 0, gz_text = 0;
    unsigned gz_threads = 0;
    std::thread splitter;
    void start_bgzf()
    {
        splitter = std::thread([this] {
            const size_t W = 8u << 20;
            // The member headers are 18 bytes in every ~30 KB of the file: the walk goes over a read-only MAPPING of it and touches one
            // page per member (a copy of every window through pread was 5 GB/s -- the whole reader's ceiling once the device inflates
            // beside the CPU threads); files that cannot be mapped go through pread windows.
            const off_t fsz = ::lseek(bfd, 0, SEEK_END);
            const unsigned char *map = nullptr;
            if (fsz > 0 && !std::getenv("BNS_BGZF_NO_MMAP")) {
                void *mp = ::mmap(nullptr, (size_t)fsz, PROT_READ, MAP_SHARED, bfd, 0);
                if (mp != MAP_FAILED) { map = static_cast<const unsigned char *>(mp); (void)::madvise(mp, (size_t)fsz, MADV_RANDOM); }
            }
            struct Unmap { const unsigned char *&m; size_t n; ~Unmap() { if (m) ::munmap(const_cast<unsigned char *>(m), n); } } unmap{map, (size_t)(fsz > 0 ? fsz : 0)};
            std::vector<unsigned char> win(map ? 0 : W + (1u << 16));
            const size_t wcap = W + (1u << 16);
            u64 at = 0, index = 0;
            BgzfTask cur_task;
            auto flush = [&](bool last) {
                if (!cur_task.members.empty()) {
                    cur_task.index = index++;
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return btasks.size() < bq_cap || stop; });
                    if (stop) return false;
                    btasks.push_back(std::move(cur_task));
                    cv.notify_all();
                    cur_task = BgzfTask();
                }
                if (last) { std::lock_guard<std::mutex> lk(mu); split_done = true; end_block = index; cv.notify_all(); }
                return true;
            };
            for (;;) {
                size_t got = 0;
                const unsigned char *wp = nullptr;
                if (map) {
                    got = at < (u64)fsz ? (size_t)std::min<u64>(wcap, (u64)fsz - at) : 0;
                    wp = map + at;
                } else {
                    while (got < win.size()) {
                        const ssize_t r = ::pread(bfd, win.data() + got, win.size() - got, (off_t)(at + got));
                        if (r < 0 && errno == EINTR) continue;
                        if (r < 0) { set_io_error(std::string("read error on the BGZF input: ") + std::strerror(errno)); flush(true); return; }
                        if (r == 0) break;
                        got += (size_t)r;
                    }
                    wp = win.data();
                }
                if (got == 0) { flush(true); return; }
                size_t p = 0;
                while (p < got) {
                    size_t pay = 0;
                    const size_t msz = bgzf_member(wp + p, got - p, pay);
                    if (!msz) {
                        // a header cut by the window (fewer than 18 bytes, or an extra field -- any XLEN -- that runs over its end):
                        // the next window starts here
                        const size_t left = got - p;
                        const bool magic = left < 4 || (wp[p] == 0x1f && wp[p + 1] == 0x8b && wp[p + 2] == 8 && (wp[p + 3] & 4));
                        const bool cut = left < 18 || (magic && 12 + (wp[p + 10] | ((size_t)wp[p + 11] << 8)) > left);
                        if (cut && got == wcap && p > 0) break;
                        set_io_error("damaged BGZF member header (or gzip members without the BC field after BGZF ones)"); flush(true); return;
                    }
                    if (p + msz > got) { if (got < wcap) { set_io_error("truncated BGZF member"); flush(true); return; } break; }
                    if (msz < pay + 8) { set_io_error("damaged BGZF member"); flush(true); return; }
                    const unsigned char *t = wp + p + msz - 8;
                    const u32 crc = t[0] | ((u32)t[1] << 8) | ((u32)t[2] << 16) | ((u32)t[3] << 24);
                    const u32 isize = t[4] | ((u32)t[5] << 8) | ((u32)t[6] << 16) | ((u32)t[7] << 24);
                    // (the format caps a member's text at 64 KiB; an unchecked trailer would size a task -- and a GPU stage -- by any u32)
                    if (isize > 65536u) { set_io_error("damaged BGZF member (recorded text size above 64 KiB)"); flush(true); return; }
                    if (isize) {
                        if (!cur_task.members.empty() && cur_task.out_bytes + isize > raw_block) { if (!flush(false)) return; }
                        if (cur_task.members.empty()) cur_task.file_off = at + p;
                        const u32 rel = (u32)(at + p - cur_task.file_off);
                        cur_task.members.push_back(BgzfMember{rel + (u32)pay, (u32)(msz - pay - 8), (u32)cur_task.out_bytes, isize, crc});
                        cur_task.out_bytes += isize;
                        cur_task.in_bytes = rel + msz;
                    }
                    p += msz;
                }
                if (p == 0) { set_io_error("damaged BGZF input (a member larger than the read window)"); flush(true); return; }
                at += p;
            }
        });
        unsigned n_inf = 6;
        if (const char *e = std::getenv("BNS_GZ_THREADS")) n_inf = (unsigned)std::max(0, std::atoi(e));
        else n_inf = (unsigned)std::max(2, std::min(32, usable_cpus() - 4));     // (the parser, packer and formatter threads want the rest; inflate scales linearly: profiles/r04_gz_scaling.txt)
        // with a device to inflate on (set_bgzf_device), GPU threads take batches of tasks off the same queue BESIDE the CPU inflaters:
        // the CPU threads are what the host's quota allows, the device adds its share on top
        const int gdev = g_bgzf_device.load();
        // (on a host of up to six CPUs the device inflates alone: two CPU inflaters there take from the parser and the packer more
        // than they add -- 4 CPUs: 8 M reads/s beside them, 11-12 M without, 3.9 M on the CPUs alone; profiles/r04_bgzf_cpus.txt)
        if (gdev >= 0 && !std::getenv("BNS_GZ_THREADS") && usable_cpus() <= 6) n_inf = 0;
        u64 ahead = 2 * n_inf;                                   // tasks inflated ahead of the parser
        if (gdev >= 0) ahead += start_bgzf_gpu(gdev, ahead);
        else if (n_inf == 0) n_inf = 1;
        for (unsigned t = 0; t < n_inf; ++t)
            producers.emplace_back([this, ahead] {
                MemberInflater inf;
                std::vector<unsigned char> in;
                for (;;) {
                    BgzfTask task;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || (!btasks.empty() && btasks.front().index < next_block + ahead) || (btasks.empty() && split_done); });
                        if (stop || btasks.empty()) return;
                        task = std::move(btasks.front());
                        btasks.pop_front();
                        cv.notify_all();
                    }
                    std::shared_ptr<Block> b;
                    bool ok = true;
                    try { b = std::make_shared<Block>(HEAD + task.out_bytes); in.resize(task.in_bytes); }
                    catch (const std::bad_alloc &) {
                        set_io_error("BGZF input: out of memory for a text block");
                        std::lock_guard<std::mutex> lk(mu);
                        end_block = std::min(end_block, next_block);
                        cv.notify_all();
                        return;
                    }
                    b->begin = HEAD;
                    for (size_t got = 0; got < task.in_bytes;) {
                        const ssize_t r = ::pread(bfd, in.data() + got, task.in_bytes - got, (off_t)(task.file_off + got));
                        if (r < 0 && errno == EINTR) continue;
                        if (r <= 0) { ok = false; break; }
                        got += (size_t)r;
                    }
                    for (const BgzfMember &m : task.members)
                        if (ok) ok = inf.run(in.data() + m.in_off, m.in_len, b->raw() + HEAD + m.out_off, m.out_len, m.crc);
                    if (!ok) set_io_error("BGZF member does not inflate to its recorded size and checksum");
                    b->end = HEAD + (ok ? task.out_bytes : 0);
                    std::lock_guard<std::mutex> lk(mu);
                    ready_at[task.index] = std::move(b);
                    cv.notify_all();
                }
            });
    }
    // BGZF members inflated on the GPU, beside the CPU inflaters.  The kernel's time hardly depends on the batch (it is ONE member's
    // serial decode, ~40 ms for 64 KiB: csrc/bns_inflate.hip), so the device wants thousands of members per call and answers late:
    // its threads therefore take their batches from the BACK of the task queue -- text the parser will not ask for until the CPU
    // inflaters, which serve the front task by task, have worked their way there.  Per batch: the compressed bytes into a page-locked
    // buffer (pread), one bns_inflate_members call, the text out of a page-locked staging buffer into ordinary pooled blocks.
    // (returns how many tasks its threads may hold: the caller adds them to the window inflated ahead of the parser)
    u64 start_bgzf_gpu(int device, u64 cpu_ahead)
    {
        unsigned BATCH = 128;                                    // tasks (of <= raw_block of text, ~64 members each) per call
        if (const char *e = std::getenv("BNS_BGZF_GPU_BATCH")) BATCH = (unsigned)std::max(1, std::min(1024, std::atoi(e)));
        unsigned n_thr = 2;
        if (const char *e = std::getenv("BNS_BGZF_GPU_THREADS")) n_thr = (unsigned)std::max(1, std::min(8, std::atoi(e)));
        gz_threads = n_thr;
        block_pool().keep_at_least((size_t)BATCH * (n_thr + 1) + 64);
        // tasks at the front of the queue that are the CPU inflaters': what they get through while the device works on a round of
        // batches -- a CPU thread inflates ~22 tasks (of 4 MiB) in the ~0.15 s a batch takes, so 11 x their look-ahead of two tasks
        // each, and no more than the device's own share.  (Too few and a dozen CPU threads wait for the device, which then has two
        // thirds of the file: 19 M reads/s either way on 16 CPUs; too many -- 256 for the two inflaters of a 4-CPU host -- and the
        // device waits for them: 7.5 M reads/s against 12 M with the device alone.)
        const size_t reserve = cpu_ahead ? std::max<size_t>((size_t)cpu_ahead, std::min<size_t>((size_t)BATCH * n_thr, 11u * (size_t)cpu_ahead)) : 0;
        bq_cap = reserve + (size_t)BATCH * (n_thr + 1);
        const u64 window = 2 * (u64)bq_cap;                      // how far ahead of the parser a batch may lie
        for (unsigned t = 0; t < n_thr; ++t)
            producers.emplace_back([this, device, reserve, BATCH, window] {
                bns_inflater *h = nullptr;
                if (bns_inflater_create(device, &h) != BNS_OK) {
                    // beside CPU inflaters the device is a help, not a need: they carry on alone; without them it is the reader
                    if (reserve) { std::fprintf(stderr, "[W] BGZF input: no inflater on GPU %d; inflating on the CPU threads only\n", device); return; }
                    set_io_error("BGZF input: could not open an inflater on the GPU (BNS_BGZF_GPU=0 inflates on the CPU)");
                    std::lock_guard<std::mutex> lk(mu);
                    end_block = std::min(end_block, next_block);
                    cv.notify_all();
                    return;
                }
                const bool trace = std::getenv("BNS_BGZF_TRACE") != nullptr;
                if (trace) std::fprintf(stderr, "[bgzf-gpu] inflater open\n");
                char *comp = nullptr, *stage = nullptr;
                size_t comp_cap = 0, stage_cap = 0;
                auto grow = [&](char *&p, size_t &cap, size_t want) {
                    if (want <= cap) return true;
                    if (p) bns_inflater_host_free(h, p);
                    p = nullptr; cap = 0;
                    void *q = nullptr;
                    if (bns_inflater_host_alloc(h, want, &q) != BNS_OK) return false;
                    p = static_cast<char *>(q); cap = want;
                    return true;
                };
                std::vector<u64> in_off, out_off;
                std::vector<u32> in_len, out_len, crc, status, want_crc;
                std::vector<BgzfTask> batch;
                std::vector<size_t> comp_at;
                double t_read = 0, t_call = 0, t_kernel = 0, t_copy = 0;
                u64 n_batches = 0, n_members = 0, n_text = 0;
                // (without CPU inflaters nobody else serves the front of the queue: the batches are then taken there, in file order, and
                // at the end of the file whatever is left is a batch)
                const bool from_front = reserve == 0;
                const size_t min_batch = from_front ? 1 : std::max<size_t>(1, BATCH / 4);
                for (;;) {
                    batch.clear();
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        // a batch worth the call's latency behind the CPU inflaters' share -- or, once the file has been split to its
                        // end, whatever is left there (the CPU threads finish the front)
                        // (and not further ahead of the parser than the window: the blocks it produces are held until they are parsed)
                        cv.wait(lk, [&] {
                            if (stop) return true;
                            const bool enough = btasks.size() >= reserve + BATCH || (split_done && btasks.size() >= reserve + min_batch);
                            if (enough && (from_front ? btasks.front().index : btasks.back().index) < next_block + window) return true;
                            return split_done && btasks.size() <= reserve;
                        });
                        if (stop || btasks.size() <= reserve) break;
                        const size_t k = std::min<size_t>(BATCH, btasks.size() - reserve);
                        if (from_front) {
                            for (size_t q = 0; q < k; ++q) { batch.push_back(std::move(btasks.front())); btasks.pop_front(); }
                        } else {
                            for (size_t q = 0; q < k; ++q) { batch.push_back(std::move(btasks.back())); btasks.pop_back(); }
                            std::reverse(batch.begin(), batch.end());
                        }
                        cv.notify_all();
                    }
                    // the batch's compressed bytes, task after task (16-byte aligned), + the decoder's read-ahead behind the last one
                    size_t comp_bytes = 0, slot_text = 0, members = 0;
                    comp_at.resize(batch.size());
                    for (size_t j = 0; j < batch.size(); ++j) {
                        comp_at[j] = comp_bytes;
                        comp_bytes += (batch[j].in_bytes + 15u) & ~size_t(15);
                        slot_text = std::max(slot_text, batch[j].out_bytes);
                        members += batch[j].members.size();
                    }
                    const size_t SLOT = (slot_text + 4095u) & ~size_t(4095);
                    if (trace) std::fprintf(stderr, "[bgzf-gpu] batch of %zu tasks (first index %llu), %zu members, slot %zu\n", batch.size(), (unsigned long long)batch[0].index, members, SLOT);
                    bool ok = grow(comp, comp_cap, std::max(comp_bytes + 64, (size_t)BATCH * (raw_block / 2))) && grow(stage, stage_cap, std::max(batch.size(), (size_t)BATCH) * SLOT);
                    double t0 = tnow();
                    for (size_t j = 0; ok && j < batch.size(); ++j)
                        for (size_t got = 0; got < batch[j].in_bytes;) {
                            const ssize_t r = ::pread(bfd, comp + comp_at[j] + got, batch[j].in_bytes - got, (off_t)(batch[j].file_off + got));
                            if (r < 0 && errno == EINTR) continue;
                            if (r <= 0) { ok = false; break; }
                            got += (size_t)r;
                        }
                    t_read += tnow() - t0;
                    if (trace) std::fprintf(stderr, "[bgzf-gpu] buffers and pread done (ok %d)\n", (int)ok);
                    in_off.resize(members); out_off.resize(members); in_len.resize(members); out_len.resize(members);
                    crc.resize(members); status.resize(members); want_crc.resize(members);
                    size_t i = 0;
                    for (size_t j = 0; j < batch.size(); ++j)
                        for (const BgzfMember &m : batch[j].members) {
                            in_off[i] = comp_at[j] + m.in_off; in_len[i] = m.in_len;
                            out_off[i] = j * SLOT + m.out_off; out_len[i] = m.out_len;
                            want_crc[i] = m.crc;
                            ++i;
                        }
                    t0 = tnow();
                    if (ok && members) {
                        const int rc = bns_inflate_members(h, reinterpret_cast<const uint8_t *>(comp), comp_bytes, in_off.data(), in_len.data(), out_off.data(), out_len.data(),
                                                           members, reinterpret_cast<uint8_t *>(stage), batch.size() * SLOT, crc.data(), status.data());
                        if (rc != BNS_OK) { set_io_error(std::string("BGZF input: the GPU inflater failed: ") + bns_inflater_error(h)); ok = false; }
                        else {
                            t_kernel += bns_inflater_last_kernel_ms(h) * 1e-3;
                            for (size_t q = 0; q < members; ++q)
                                if (status[q] != 0 || crc[q] != want_crc[q]) { ok = false; break; }
                            if (!ok) set_io_error("BGZF member does not inflate to its recorded size and checksum");
                        }
                    } else if (!ok) set_io_error("BGZF input: read error, or no page-locked memory for the GPU inflater");
                    t_call += tnow() - t0;
                    if (trace) std::fprintf(stderr, "[bgzf-gpu] call done (ok %d)\n", (int)ok);
                    ++n_batches; n_members += members;
                    t0 = tnow();
                    for (size_t j = 0; j < batch.size(); ++j) {
                        auto b = std::make_shared<Block>(HEAD + (ok ? batch[j].out_bytes : 0) + 8);
                        b->begin = HEAD;
                        if (ok) std::memcpy(b->raw() + HEAD, stage + j * SLOT, batch[j].out_bytes);
                        b->end = HEAD + (ok ? batch[j].out_bytes : 0);
                        n_text += ok ? batch[j].out_bytes : 0;
                        std::lock_guard<std::mutex> lk(mu);
                        ready_at[batch[j].index] = std::move(b);
                        cv.notify_all();
                    }
                    t_copy += tnow() - t0;
                }
                if (trace) std::fprintf(stderr, "[bgzf-gpu] thread leaves\n");
                if (comp) bns_inflater_host_free(h, comp);
                if (stage) bns_inflater_host_free(h, stage);
                bns_inflater_destroy(h);
                if (trace) std::fprintf(stderr, "[bgzf-gpu] inflater closed\n");
                std::lock_guard<std::mutex> lk(mu);
                gz_t_read += t_read; gz_t_call += t_call; gz_t_kernel += t_kernel; gz_t_copy += t_copy;
                gz_batches += n_batches; gz_members += n_members; gz_text += n_text;
            });
        return (u64)(bq_cap - cpu_ahead);
    }
    // One plain gzip stream on many threads (pgzip.hpp): scan tasks decode chunks of compressed bytes into marker symbols from a
    // block header they find themselves; the coordinator takes them in file order, checks that they meet (else decodes the chunk
    // again from where the one in front ended), hands every chunk its 32 KiB window and cuts it into resolve tasks -- one text
    // block each, CRC-32 per gzip member on the way; blocks reach the parser in file order through ready_at.
    bool pgz = false;
    const unsigned char *pgz_data = nullptr;
    size_t pgz_n = 0;
    struct PgzChunk {
        u64 index = 0;
        pgz::Scan scan;
        bool scanned = false;
        std::shared_ptr<std::vector<unsigned char>> window;      // the resolved 32 KiB in front of it
        u64 block_base = 0;
        u32 n_pieces = 0, pieces_done = 0;
        struct PieceCrc { u32 seg, crc; u64 len; };
        std::vector<std::vector<PieceCrc>> piece_crc;            // per piece: its share of every member stretch it overlaps
    };
    struct PgzPiece { std::shared_ptr<PgzChunk> c; u32 piece; u64 begin, end; };
    std::map<u64, std::shared_ptr<PgzChunk>> pgz_scanned;
    std::deque<PgzPiece> pgz_pieces;
    std::vector<std::vector<uint16_t>> pgz_sym_pool;
    u64 pgz_next_scan = 0, pgz_stitched = 0, pgz_n_chunks = 0, pgz_first = 0, pgz_chunk_bytes = 2u << 20;
    unsigned pgz_threads = 2;
    bool pgz_all_dispatched = false;
    double pgz_t_scan = 0, pgz_t_alloc = 0, pgz_t_resolve = 0, pgz_t_crc = 0, pgz_t_coord = 0;   // seconds of work, summed over the threads (BNS_CLI_TIMING)
    bool pgz_no_search = false;          // four chunks in a row found no block header (a stream of stored blocks?): the coordinator decodes the rest itself
    std::thread pgz_coord;
    static uint32_t crc32_of(const unsigned char *p, size_t n)
    {
        if (const LibDeflate *ld = libdeflate()) return ld->crc(0, p, n);
        uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
        while (n) { const size_t k = std::min<size_t>(n, 1u << 30); c = (uint32_t)crc32(c, p, (uInt)k); p += k; n -= k; }
        return c;
    }
    // a regular gzip file of some size whose first member has a deflate payload: map it (false: the zlib reader takes it)
    bool map_pgz(int fd_)
    {
        if (std::getenv("BNS_NO_PGZ")) return false;
        struct stat st;
        if (::fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) return false;
        const size_t min_bytes = std::getenv("BNS_PGZ_CHUNK") ? 64 : (4u << 20);      // (small files: one zlib stream is as fast)
        if ((size_t)st.st_size < min_bytes) return false;
        void *m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return false;
        const u64 he = pgz::gzip_header_end(static_cast<const unsigned char *>(m), (u64)st.st_size, 0);
        if (!he) { ::munmap(m, (size_t)st.st_size); return false; }
        (void)::madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        pgz_data = static_cast<const unsigned char *>(m); pgz_n = (size_t)st.st_size; pgz_first = he;
        return true;
    }
    void pgz_scan_one(PgzChunk &c, bool search, u64 from_bit, bool fresh)
    {
        const u64 c1 = std::min<u64>(pgz_first + (c.index + 1) * pgz_chunk_bytes, pgz_n);
        const u64 stop = c.index + 1 >= pgz_n_chunks ? (u64)pgz_n * 8 : c1 * 8;
        pgz::scan_chunk(pgz_data, pgz_n, from_bit, search, fresh, stop, c.scan);
    }
    void pgz_worker()
    {
        for (;;) {
            PgzPiece piece;
            std::shared_ptr<PgzChunk> sc;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] {
                    return stop || !pgz_pieces.empty() || (!pgz_no_search && pgz_next_scan < pgz_n_chunks && pgz_next_scan < pgz_stitched + 2 * pgz_threads) ||
                           pgz_all_dispatched;
                });
                if (stop) return;
                if (!pgz_pieces.empty()) { piece = std::move(pgz_pieces.front()); pgz_pieces.pop_front(); }
                else if (!pgz_no_search && pgz_next_scan < pgz_n_chunks && pgz_next_scan < pgz_stitched + 2 * pgz_threads && !pgz_all_dispatched) {
                    sc = std::make_shared<PgzChunk>();
                    sc->index = pgz_next_scan++;
                    if (!pgz_sym_pool.empty()) { sc->scan.sym = std::move(pgz_sym_pool.back()); pgz_sym_pool.pop_back(); }
                } else if (pgz_all_dispatched) return;
                else continue;
            }
            if (sc) {
                // (chunk 0 starts at the member's first block; the others look for a header from their first byte on)
                const double ts = tnow();
                if (sc->index == 0) pgz_scan_one(*sc, false, (u64)pgz_first * 8, true);
                else pgz_scan_one(*sc, true, (pgz_first + sc->index * pgz_chunk_bytes) * 8, false);
                std::lock_guard<std::mutex> lk(mu);
                pgz_t_scan += tnow() - ts;
                sc->scanned = true;
                pgz_scanned[sc->index] = sc;
                cv.notify_all();
                continue;
            }
            // resolve one text block
            PgzChunk &c = *piece.c;
            const size_t len = (size_t)(piece.end - piece.begin);
            const double tr0 = tnow();
            auto b = std::make_shared<Block>(HEAD + len);
            b->begin = HEAD; b->end = HEAD + len;
            const double tr1 = tnow();
            pgz::resolve(c.scan.sym.data() + pgz::WINDOW + piece.begin, len, c.window->data(), reinterpret_cast<unsigned char *>(b->raw()) + HEAD);
            const double tr2 = tnow();
            std::vector<PgzChunk::PieceCrc> crcs;
            for (u32 g = 0; g < c.scan.segs.size(); ++g) {
                const u64 a = std::max(piece.begin, c.scan.segs[g].begin), e = std::min(piece.end, c.scan.segs[g].end);
                if (a < e) crcs.push_back({g, crc32_of(reinterpret_cast<const unsigned char *>(b->raw()) + HEAD + (a - piece.begin), (size_t)(e - a)), e - a});
            }
            const double tr3 = tnow();
            std::lock_guard<std::mutex> lk(mu);
            pgz_t_alloc += tr1 - tr0; pgz_t_resolve += tr2 - tr1; pgz_t_crc += tr3 - tr2;
            c.piece_crc[piece.piece] = std::move(crcs);
            ++c.pieces_done;
            ready_at[c.block_base + piece.piece] = std::move(b);
            cv.notify_all();
        }
    }
    void start_pgz()
    {
        if (const char *e = std::getenv("BNS_GZ_THREADS")) pgz_threads = (unsigned)std::max(1, std::atoi(e));
        else pgz_threads = (unsigned)std::max(2, std::min(32, usable_cpus() - 4));
        if (const char *e = std::getenv("BNS_PGZ_CHUNK")) pgz_chunk_bytes = (u64)std::max(4096, std::atoi(e));
        pgz_n_chunks = ((u64)pgz_n - pgz_first + pgz_chunk_bytes - 1) / pgz_chunk_bytes;
        for (unsigned t = 0; t < pgz_threads; ++t) producers.emplace_back([this] { pgz_worker(); });
        pgz_coord = std::thread([this] {
            auto window = std::make_shared<std::vector<unsigned char>>(pgz::WINDOW, 0);
            u64 expect = (u64)pgz_first * 8, blocks = 0;
            std::deque<std::shared_ptr<PgzChunk>> unverified;
            uint32_t run_crc = 0; u64 run_len = 0; bool run_any = false;
            bool failed = false;
            unsigned search_failures = 0;
            // fold the CRCs of finished chunks, in order; at a member's end compare with its trailer
            auto verify = [&](bool wait_all) {
                for (;;) {
                    std::shared_ptr<PgzChunk> c;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        if (unverified.empty()) return;
                        if (wait_all) cv.wait(lk, [&] { return stop || unverified.front()->pieces_done == unverified.front()->n_pieces; });
                        if (stop || unverified.front()->pieces_done != unverified.front()->n_pieces) return;
                        c = unverified.front(); unverified.pop_front();
                    }
                    // per member stretch: its pieces' CRCs in order
                    for (u32 g = 0; g < c->scan.segs.size(); ++g) {
                        for (const auto &pc : c->piece_crc)
                            for (const auto &x : pc)
                                if (x.seg == g) {
                                    run_crc = run_any ? (uint32_t)crc32_combine(run_crc, x.crc, (z_off_t)x.len) : x.crc;
                                    run_any = true; run_len += x.len;
                                }
                        if (c->scan.segs[g].member_end) {
                            const uint32_t have = run_any ? run_crc : (uint32_t)crc32(0L, Z_NULL, 0);
                            if (have != c->scan.segs[g].crc || (uint32_t)run_len != c->scan.segs[g].isize)
                                set_io_error("the gzip input does not match its checksum (CRC-32 / length of a member)");
                            run_any = false; run_crc = 0; run_len = 0;
                        }
                    }
                    // (symbol buffers are recycled: tens of MB each, and fresh memory costs a page fault per 4 KiB)
                    std::lock_guard<std::mutex> lk(mu);
                    pgz_sym_pool.push_back(std::move(c->scan.sym));
                    c->scan.sym = std::vector<uint16_t>();
                }
            };
            for (u64 i = 0; i < pgz_n_chunks && !failed; ++i) {
                std::shared_ptr<PgzChunk> c;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    if (pgz_no_search && i >= pgz_next_scan) {               // nobody was handed this chunk: it is decoded here, from where the last one ended
                        c = std::make_shared<PgzChunk>();
                        c->index = i;
                        pgz_next_scan = i + 1;
                        if (!pgz_sym_pool.empty()) { c->scan.sym = std::move(pgz_sym_pool.back()); pgz_sym_pool.pop_back(); }
                    } else {
                        cv.wait(lk, [&] { return stop || pgz_scanned.count(i); });
                        if (stop) return;
                        c = pgz_scanned[i]; pgz_scanned.erase(i);
                    }
                }
                if (c->scanned && !c->scan.ok && i > 0) {
                    if (++search_failures >= 4) { std::lock_guard<std::mutex> lk(mu); pgz_no_search = true; }
                } else if (c->scanned) search_failures = 0;
                if (!c->scan.ok || c->scan.start_bit != expect) {
                    // the chunks do not meet (the true first block was a stored / fixed / final one, a false header, or nothing found): again, from where
                    // the chunk in front ended
                    if (i == 0) { set_io_error(std::string("damaged gzip input: ") + c->scan.err); failed = true; break; }
                    pgz_scan_one(*c, false, expect, false);
                    if (!c->scan.ok) { set_io_error(std::string("damaged gzip input: ") + c->scan.err); failed = true; break; }
                }
                expect = c->scan.end_bit;
                c->window = window;
                const double tc0 = tnow();
                auto nw = std::make_shared<std::vector<unsigned char>>(pgz::WINDOW);
                pgz::next_window(c->scan, window->data(), nw->data());
                window = nw;
                pgz_t_coord += tnow() - tc0;
                const u64 n_out = c->scan.n_out;
                c->n_pieces = (u32)((n_out + raw_block - 1) / raw_block);
                c->piece_crc.resize(c->n_pieces);
                c->block_base = blocks;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    // (text blocks not yet taken by the parser are bounded: the scans run ahead, the resolves wait here)
                    cv.wait(lk, [&] { return stop || blocks < next_block + 8 + 4 * pgz_threads; });
                    if (stop) return;
                    for (u32 p = 0; p < c->n_pieces; ++p)
                        pgz_pieces.push_back(PgzPiece{c, p, (u64)p * raw_block, std::min<u64>(n_out, (u64)(p + 1) * raw_block)});
                    blocks += c->n_pieces;
                    ++pgz_stitched;
                    unverified.push_back(c);
                    cv.notify_all();
                }
                verify(false);
                if (c->scan.eof) break;
                if (i + 1 == pgz_n_chunks && !c->scan.eof) { set_io_error("the gzip input ends inside a member (truncated file)"); failed = true; }
            }
            verify(true);
            std::lock_guard<std::mutex> lk(mu);
            end_block = failed ? std::min(end_block, blocks) : blocks;
            pgz_all_dispatched = true;
            cv.notify_all();
        });
    }
    void start()
    {
        if (pgz) { start_pgz(); return; }
        if (bgzf) { start_bgzf(); return; }
        use_pread = fd >= 0 && ::lseek(fd, 0, SEEK_CUR) != (off_t)-1;
        if (use_pread) {
            for (unsigned t = 0; t < N_PRODUCERS; ++t)
                producers.emplace_back([this, t] {
                    for (u64 i = t;; i += N_PRODUCERS) {
                        {
                            std::unique_lock<std::mutex> lk(mu);
                            cv.wait(lk, [&] { return i < next_block + 2 * N_PRODUCERS || stop || i >= end_block; });
                            if (stop || i >= end_block) return;
                        }
                        auto b = std::make_shared<Block>(HEAD + raw_block);
                        b->begin = HEAD;
                        const u64 at = range_begin + i * raw_block;
                        const size_t want = at >= range_end ? 0 : (size_t)std::min<u64>(raw_block, range_end - at);
                        size_t got = 0;
                        while (got < want) {                             // short counts are normal
                            const ssize_t r = ::pread(fd, b->raw() + HEAD + got, want - got, (off_t)(at + got));
                            if (r < 0 && errno == EINTR) continue;
                            if (r < 0) set_io_error(std::string("read error on the input: ") + std::strerror(errno));
                            if (r <= 0) break;
                            got += (size_t)r;
                        }
                        b->end = HEAD + got;
                        std::lock_guard<std::mutex> lk(mu);
                        if (got) ready_at[i] = std::move(b);
                        if (got < raw_block) end_block = std::min(end_block, got ? i + 1 : i);
                        cv.notify_all();
                        if (got < raw_block) return;
                    }
                });
            return;
        }
        producer = std::thread([this] {
            for (;;) {
                auto b = std::make_shared<Block>(HEAD + raw_block);
                b->begin = HEAD;
                const size_t got = read_some(b->raw() + HEAD, raw_block);
                b->end = HEAD + got;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready.size() < 3 || stop; });
                if (stop) return;
                const bool last = got < raw_block;
                if (got) ready.push_back(std::move(b));
                if (last) { producer_done = true; cv.notify_all(); return; }
                cv.notify_all();
            }
        });
    }
    // a reader over blocks already in memory (ChunkSource's stretches of a BGZF input): no threads, no file
    bool mem = false;
    std::deque<std::shared_ptr<Block>> mem_blocks;
    std::function<std::shared_ptr<Block>()> mem_more;
    // next raw block or nullptr at end of stream
    double t_blocked = 0;                                       // time the parser spent waiting for a block
    std::shared_ptr<Block> pop_raw()
    {
        auto b = pop_raw_unchecked();
        std::string e;
        { std::lock_guard<std::mutex> lk(mu); e = io_error; }
        if (!e.empty()) die(e);                                  // (a producer's read failed: not an end of file)
        return b;
    }
    std::shared_ptr<Block> pop_raw_unchecked()
    {
        if (mem) {
            if (!mem_blocks.empty()) { auto b = std::move(mem_blocks.front()); mem_blocks.pop_front(); return b; }
            return mem_more ? mem_more() : nullptr;
        }
        std::unique_lock<std::mutex> lk(mu);
        if (use_pread || bgzf || pgz) {
            if (!(ready_at.count(next_block) || next_block >= end_block)) {
                const auto t0 = std::chrono::steady_clock::now();
                cv.wait(lk, [&] { return ready_at.count(next_block) || next_block >= end_block; });
                t_blocked += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            auto it = ready_at.find(next_block);
            if (it == ready_at.end()) return nullptr;
            auto b = std::move(it->second);
            ready_at.erase(it);
            ++next_block;
            cv.notify_all();
            return b;
        }
        cv.wait(lk, [&] { return !ready.empty() || producer_done; });
        if (ready.empty()) return nullptr;
        auto b = std::move(ready.front());
        ready.pop_front();
        cv.notify_all();
        return b;
    }
    // Make cur = [unparsed tail of cur from offset `from`] + fresh data; sets final_ when nothing more can arrive.
    void refill(size_t from)
    {
        if (final_) return;
        const size_t tail = cur ? cur->size() - from : 0;
        const char *tail_p = cur ? cur->data() + from : nullptr;
        auto raw = pop_raw();
        if (!raw) { final_ = true; if (cur) pos = from; return; }
        if (tail <= HEAD) {                                      // the usual case: the tail goes into the block's headroom
            if (tail) std::memcpy(raw->raw() + raw->begin - tail, tail_p, tail);
            raw->begin -= tail;
            cur = std::move(raw);
            pos = 0;
            return;
        }
        // a record larger than the headroom (a genome): concatenate, asking for as much again as is already there so
        // that re-parsing it stays O(n)
        const size_t want = std::max<size_t>(raw_block, tail);
        std::vector<std::shared_ptr<Block>> more{raw};
        size_t added = raw->size();
        while (added < want) {
            auto r = pop_raw();
            if (!r) { final_ = true; break; }
            added += r->size();
            more.push_back(std::move(r));
        }
        auto nb = std::make_shared<Block>(tail + added);
        std::memcpy(nb->raw(), tail_p, tail);
        size_t at = tail;
        for (auto &r : more) { std::memcpy(nb->raw() + at, r->data(), r->size()); at += r->size(); }
        nb->end = at;
        cur = std::move(nb);
        pos = 0;
    }

    bool need_refill = false;     // cur has been parsed as far as its data goes

    enum { OK = 0, NEED_MORE = 1 };
    // One kseq_read step over base[pos..end).  rc receives kseq's return value when the result is OK.
    static int parse_one(const char *base, size_t end, bool final_, size_t &pos, bool &at_header, bseq1_t &rec,
                         std::deque<std::string> &arena, int &rc);
    // cur is a block with something to parse in it (refilled as needed), or false at the end of the stream
    bool have_block()
    {
        if (need_refill) {
            if (final_) return false;
            refill(pos);                                             // carries the unparsed tail over
            need_refill = false;
            if (cur) cur->arenas.emplace_back();                     // (records handed out earlier may point into the older arenas)
        }
        if (!cur) {
            refill(0);
            if (!cur) { final_ = true; return false; }
            cur->arenas.emplace_back();
        }
        return true;
    }
    void register_with(ReadChunk &owner)                         // the views handed out point into cur
    {
        if (!(reg_owner == &owner && reg_epoch == owner.epoch && reg_block == cur.get())) {
            owner.blocks.push_back(cur);
            reg_owner = &owner; reg_epoch = owner.epoch; reg_block = cur.get();
        }
    }
};

int SeqReader::Impl::parse_one(const char *base, size_t end, bool final_, size_t &pos, bool &at_header, bseq1_t &rec,
                               std::deque<std::string> &arena, int &rc)
{
    size_t p = pos;
    if (!at_header) {                                            // jump to the next '>' / '@', wherever it is
        while (p < end && base[p] != '>' && base[p] != '@') ++p;
        if (p == end) {
            pos = end;
            if (!final_) return NEED_MORE;
            rc = -1; return OK;
        }
    }
    const size_t rec_start = p;                                  // on NEED_MORE everything from here is kept
    auto need_more = [&] { pos = rec_start; at_header = true; return (int)NEED_MORE; };
    ++p;
    // name = first whitespace-delimited token; comment = rest of the header line
    size_t q = p;
    while (q < end && !is_space((unsigned char)base[q])) ++q;
    if (q == end && !final_) return need_more();
    rec.name = std::string_view(base + p, q - p);
    rec.comment = rec.seq = rec.qual = std::string_view();
    if (q == end && rec.name.empty()) { pos = end; at_header = false; rc = -1; return OK; }
    p = q;
    bool stream_ended = (q == end);
    if (!stream_ended) {
        const char delim = base[p++];
        if (delim != '\n') {
            const void *nl = std::memchr(base + p, '\n', end - p);
            if (!nl && !final_) return need_more();
            size_t e = nl ? (size_t)((const char *)nl - base) : end;
            size_t len = e - p;
            if (len > 1 && base[p + len - 1] == '\r') --len;
            rec.comment = std::string_view(base + p, len);
            p = nl ? e + 1 : end;
        }
    }
    // sequence lines until a line starts with '>', '@' or '+'
    std::string *acc = nullptr;                                  // set once the sequence is not one contiguous line
    std::string_view seq;
    int c = -1;
    for (;;) {
        if (p == end) { if (!final_) return need_more(); c = -1; break; }
        c = (unsigned char)base[p];
        if (c == '>' || c == '+' || c == '@') break;
        if (c == '\n') { ++p; continue; }
        const void *nl = std::memchr(base + p, '\n', end - p);
        if (!nl && !final_) return need_more();
        const size_t e = nl ? (size_t)((const char *)nl - base) : end;
        if (!acc && seq.empty()) {
            size_t len = e - p;
            if (len > 1 && base[p + len - 1] == '\r') --len;
            seq = std::string_view(base + p, len);
        } else {
            if (!acc) { arena.emplace_back(seq); acc = &arena.back(); }
            acc->append(base + p, e - p);
            if (acc->size() > 1 && acc->back() == '\r') acc->pop_back();
        }
        p = nl ? e + 1 : end;
    }
    if (acc) seq = *acc;
    rec.seq = seq;
    if (c != '+') {                                              // FASTA
        pos = p; at_header = (c == '>' || c == '@');
        rc = (int)seq.size(); return OK;
    }
    // the rest of the '+' line; the stream ending here means no quality
    {
        const void *nl = std::memchr(base + p, '\n', end - p);
        if (!nl) {
            if (!final_) return need_more();
            pos = end; at_header = false; rc = -2; return OK;
        }
        p = (size_t)((const char *)nl - base) + 1;
    }
    std::string *qacc = nullptr;
    std::string_view qual;
    while (qual.size() < seq.size()) {
        if (p == end) { if (!final_) return need_more(); break; }
        const void *nl = std::memchr(base + p, '\n', end - p);
        if (!nl && !final_) return need_more();
        const size_t e = nl ? (size_t)((const char *)nl - base) : end;
        if (!qacc && qual.empty()) {
            size_t len = e - p;
            if (len > 1 && base[p + len - 1] == '\r') --len;
            qual = std::string_view(base + p, len);
            if (qual.empty()) { arena.emplace_back(); qacc = &arena.back(); }   // an empty first line: keep accumulating
        } else {
            if (!qacc) { arena.emplace_back(qual); qacc = &arena.back(); }
            qacc->append(base + p, e - p);
            if (qacc->size() > 1 && qacc->back() == '\r') qacc->pop_back();
            qual = *qacc;
        }
        p = nl ? e + 1 : end;
    }
    rec.qual = qual;
    pos = p; at_header = false;
    rc = qual.size() != seq.size() ? -2 : (int)seq.size();
    return OK;
}

SeqReader::SeqReader(const char *path, size_t block_bytes, u64 range_begin, u64 range_end) : impl_(new Impl)
{
    if (block_bytes) impl_->raw_block = block_bytes;
    else if (const char *e = std::getenv("BNS_READER_BLOCK")) { const long v = std::atol(e); if (v >= 256) impl_->raw_block = (size_t)v; }   // (tests: small text blocks on small inputs)
    impl_->range_begin = range_begin; impl_->range_end = range_end;
    // gzip magic -> zlib; anything else is read as is (gzread would do the same, through two more copies)
    unsigned char magic[2] = {0, 0};
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) die(std::string("Could not open ") + path + " for reading.");
    const ssize_t got = ::pread(fd, magic, 2, 0);
    unsigned char head[64];
    size_t pay = 0;
    const ssize_t hgot = ::pread(fd, head, sizeof(head), 0);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b && hgot >= 18 && bgzf_member(head, (size_t)hgot, pay) && !std::getenv("BNS_NO_BGZF")
        && ::lseek(fd, 0, SEEK_CUR) != (off_t)-1) {
        if (range_begin != 0 || range_end != ~0ULL) die(std::string("a byte range of a gzip file was asked for: ") + path);
        impl_->bgzf = true;                                      // blocked gzip: members inflated side by side
        impl_->bfd = fd;
    } else if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b && impl_->map_pgz(fd)) {
        if (range_begin != 0 || range_end != ~0ULL) die(std::string("a byte range of a gzip file was asked for: ") + path);
        impl_->pgz = true;                                       // one gzip stream, inflated on many threads (pgzip.hpp)
        impl_->bfd = fd;
    } else if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        ::close(fd);
        impl_->fp = gzopen(path, "rb");
        if (!impl_->fp) die(std::string("Could not open ") + path + " for reading.");
        gzbuffer(impl_->fp, 1 << 20);
        if (range_begin != 0 || range_end != ~0ULL) die(std::string("a byte range of a gzip file was asked for: ") + path);
    } else {
        impl_->fd = fd;
    }
    impl_->start();
    if (!impl_->use_pread && !impl_->bgzf && !impl_->pgz && (range_begin != 0 || range_end != ~0ULL)) die(std::string("a byte range of a pipe was asked for: ") + path);
}

SeqReader::SeqReader(std::deque<std::shared_ptr<TextBlock>> blocks, std::function<std::shared_ptr<TextBlock>()> more) : impl_(new Impl)
{
    impl_->mem = true;
    impl_->mem_blocks = std::move(blocks);
    impl_->mem_more = std::move(more);
}
std::shared_ptr<TextBlock> SeqReader::take_block() { return impl_->pop_raw(); }
bool SeqReader::is_bgzf() const { return impl_->bgzf; }

double SeqReader::seconds_blocked() const { return impl_->t_blocked; }
int SeqReader::last_status() const { return impl_->saw_truncated ? -2 : impl_->last_rc; }

SeqReader::~SeqReader()
{
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->stop = true;
    }
    impl_->cv.notify_all();
    if (impl_->producer.joinable()) impl_->producer.join();
    if (impl_->splitter.joinable()) impl_->splitter.join();
    if (impl_->pgz_coord.joinable()) impl_->pgz_coord.join();
    if (impl_->pgz && std::getenv("BNS_CLI_TIMING"))
        std::fprintf(stderr, "[timing] gzip reader (%u threads): scan %.3f s, block alloc %.3f, resolve %.3f, crc %.3f, coordinator %.3f (summed over the threads)\n",
                     impl_->pgz_threads, impl_->pgz_t_scan, impl_->pgz_t_alloc, impl_->pgz_t_resolve, impl_->pgz_t_crc, impl_->pgz_t_coord);
    for (auto &t : impl_->producers) t.join();
    if (impl_->gz_threads && std::getenv("BNS_CLI_TIMING"))
        std::fprintf(stderr, "[timing] BGZF on the GPU (%u threads): %llu batches, %llu members, %.2f GB of text; copy-out %.3f s, pread %.3f, calls %.3f of which kernel %.3f (summed over the threads)\n",
                     impl_->gz_threads, (unsigned long long)impl_->gz_batches, (unsigned long long)impl_->gz_members, impl_->gz_text / 1e9, impl_->gz_t_copy, impl_->gz_t_read,
                     impl_->gz_t_call, impl_->gz_t_kernel);
    if (impl_->pgz_data) ::munmap(const_cast<unsigned char *>(impl_->pgz_data), impl_->pgz_n);
    if (impl_->bfd >= 0) ::close(impl_->bfd);
    if (impl_->fp) gzclose(impl_->fp);
    if (impl_->fd >= 0) ::close(impl_->fd);
}

// The usual record -- '@' header, one sequence line, a '+' line, one quality line of the sequence's length, no '\r' -- parsed with
// three line scans and one bounded one, all of it inside the block.  Anything else (a FASTA record, wrapped lines, CRLF, a record
// that touches the end of the block) returns false with nothing changed and goes through parse_one, whose result for a record
// this function accepts is the same: name / comment as kseq splits the header, pos just past the quality line's newline.
static inline bool fast_fastq(const char *base, size_t end, size_t &pos, bseq1_t &rec, int &rc)
{
    if (base[pos] != '@') return false;
    const char *const e = base + end;
    const char *h = base + pos + 1;
    const char *nl = static_cast<const char *>(std::memchr(h, '\n', (size_t)(e - h)));
    if (!nl) return false;
    const char *q = h;
    while (!is_space((unsigned char)*q)) ++q;                    // stops at nl at the latest
    const char *s = nl + 1;
    if (s >= e) return false;
    const char c0 = *s;
    if (c0 == '>' || c0 == '+' || c0 == '@' || c0 == '\n') return false;
    const char *snl = static_cast<const char *>(std::memchr(s, '\n', (size_t)(e - s)));
    if (!snl || snl + 1 >= e || snl[1] != '+' || snl[-1] == '\r') return false;
    const size_t len = (size_t)(snl - s);
    const char *pnl = static_cast<const char *>(std::memchr(snl + 1, '\n', (size_t)(e - snl - 1)));
    if (!pnl) return false;
    const char *ql = pnl + 1;
    if ((size_t)(e - ql) <= len) return false;
    if (static_cast<const char *>(std::memchr(ql, '\n', len + 1)) != ql + len || ql[len - 1] == '\r') return false;
    rec.name = std::string_view(h, (size_t)(q - h));
    if (q == nl) rec.comment = std::string_view();
    else {
        size_t cl = (size_t)(nl - q - 1);
        if (cl > 1 && nl[-1] == '\r') --cl;
        rec.comment = std::string_view(q + 1, cl);
    }
    rec.seq = std::string_view(s, len);
    rec.qual = std::string_view(ql, len);
    rc = (int)len;
    pos = (size_t)(ql + len + 1 - base);
    return true;
}

// (A multi-threaded parser -- one stretch of a block per thread, record starts guessed from "@...\n...\n+" and every seam checked
// -- was measured and dropped: one thread parses 50 M reads/s = 16 GB/s of FASTQ on the box's host.)
int SeqReader::read(bseq1_t &rec, ReadChunk &owner)
{
    Impl &m = *impl_;
    for (;;) {
        if (!m.have_block()) return m.last_rc = -1;
        const char *base = m.cur->data();
        const size_t end = m.cur->size();
        if (!m.at_header) {                                          // what kseq does first: skip to the next '>' / '@'
            while (m.pos < end && base[m.pos] != '>' && base[m.pos] != '@') ++m.pos;
            if (m.pos == end) { m.need_refill = true; continue; }
            m.at_header = true;
        }
        int rc;
        if (fast_fastq(base, end, m.pos, rec, rc)) m.at_header = false;
        else {
            std::deque<std::string> &arena = m.cur->arenas.back();
            const size_t mark = arena.size();
            if (Impl::parse_one(base, end, m.final_, m.pos, m.at_header, rec, arena, rc) == Impl::NEED_MORE) {
                while (arena.size() > mark) arena.pop_back();        // the partial record is parsed again after the refill
                m.need_refill = true;
                continue;
            }
            if (rc == -1) { m.need_refill = true; return m.last_rc = -1; }       // (only when nothing more can arrive)
        }
        m.register_with(owner);
        if (rc < 0) { m.last_rc = rc; m.saw_truncated = true; }
        return rc;
    }
}

void RecVec::reserve(size_t cap)
{
    if (cap <= cap_) return;
    void *q = nullptr;
    if (posix_memalign(&q, 64, cap * sizeof(bseq1_t)) != 0 || !q) die("out of host memory");
    if (n_) std::memcpy(q, static_cast<const void *>(p_), n_ * sizeof(bseq1_t));
    std::free(p_);
    p_ = static_cast<bseq1_t *>(q); cap_ = cap;
}

void RecVec::push_back_stream(const bseq1_t &r)
{
    if (n_ == cap_) reserve(cap_ ? 2 * cap_ : 1024);
    const __m128i *s = reinterpret_cast<const __m128i *>(&r);
    __m128i *d = reinterpret_cast<__m128i *>(p_ + n_);
    _mm_stream_si128(d, _mm_loadu_si128(s));
    _mm_stream_si128(d + 1, _mm_loadu_si128(s + 1));
    _mm_stream_si128(d + 2, _mm_loadu_si128(s + 2));
    _mm_stream_si128(d + 3, _mm_loadu_si128(s + 3));
    ++n_;
}

void RecVec::publish() { _mm_sfence(); }

static inline void trim_readno(std::string_view &s)            // kseq_declare.h:106-110
{
    const size_t l = s.size();
    if (l > 2 && s[l - 2] == '/' && (unsigned)(s[l - 1] - '0') < 10u) s.remove_suffix(2);
}

// bseq_read's loop for one file: read()'s loop with the records going straight into out.recs (no call and no copy per record).
// A truncated record is left unread for the caller's read() to report.
void SeqReader::fill(long chunk_size, ReadChunk &out, long &size, size_t max_records)
{
    Impl &m = *impl_;
    auto enough = [&] { return (size >= chunk_size && (out.recs.size() & 1) == 0) || (max_records && out.recs.size() >= max_records); };
    while (m.have_block()) {
        std::deque<std::string> &arena = m.cur->arenas.back();
        const char *base = m.cur->data();
        const size_t end = m.cur->size();
        const bool final_ = m.final_;
        size_t pos = m.pos;
        bool at_header = m.at_header, registered = false, stop = false;
        for (;;) {
            if (!at_header) {
                while (pos < end && base[pos] != '>' && base[pos] != '@') ++pos;
                if (pos == end) { m.need_refill = true; break; }
                at_header = true;
            }
            bseq1_t rec;
            int rc;
            if (fast_fastq(base, end, pos, rec, rc)) at_header = false;
            else {
                const size_t mark = arena.size(), rec_start = pos;
                if (Impl::parse_one(base, end, final_, pos, at_header, rec, arena, rc) == Impl::NEED_MORE) {
                    while (arena.size() > mark) arena.pop_back();
                    m.need_refill = true;
                    break;
                }
                if (rc == -1) { m.need_refill = true; break; }
                if (rc < 0) {                                        // truncated: not consumed here
                    while (arena.size() > mark) arena.pop_back();
                    pos = rec_start; at_header = true; stop = true;
                    break;
                }
            }
            if (!registered) { m.register_with(out); registered = true; }
            trim_readno(rec.name);
            out.recs.push_back_stream(rec);
            size += (long)rec.seq.size();
            if (enough()) { stop = true; break; }
        }
        m.pos = pos; m.at_header = at_header;
        if (stop) return;
    }
}

int bseq_read(int chunk_size, SeqReader &r1, SeqReader *r2, ReadChunk &out)
{
    out.clear();
    out.recs.reserve((size_t)chunk_size / 64 + 16);             // ~ records of >= 64 bases; avoids regrowth copies
    long size = 0;
    bseq1_t a, b;
    if (!r2) {
        r1.fill(chunk_size, out, size);
        if (!(size >= chunk_size && (out.recs.size() & 1) == 0)) {     // the stream ended, or a truncated record is next
            while (r1.read(a, out) >= 0) {
                trim_readno(a.name);
                size += a.l_seq();
                out.recs.push_back(a);
                if (size >= chunk_size && (out.recs.size() & 1) == 0) break;
            }
        }
        RecVec::publish();
        return (int)out.recs.size();
    }
    while (r1.read(a, out) >= 0) {
        if (r2 && r2->read(b, out) < 0) { std::fprintf(stderr, "[W::bseq_read] the 2nd file has fewer sequences.\n"); break; }
        trim_readno(a.name);
        size += a.l_seq();
        out.recs.push_back_stream(a);
        if (r2) { trim_readno(b.name); size += b.l_seq(); out.recs.push_back_stream(b); }
        if (size >= chunk_size && (out.recs.size() & 1) == 0) break;
    }
    if (size == 0 && r2 && r2->read(b, out) >= 0) std::fprintf(stderr, "[W::bseq_read] the 1st file has fewer sequences.\n");
    RecVec::publish();
    return (int)out.recs.size();
}

// ---------------------------------------------------------------------------------------------- formatting
namespace {
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
}  // namespace

// classifier.h:112-129, written through a raw pointer into room the caller reserved (kraken_line_bound), not byte by byte through
// push_back: the formatter was 65 ns per read, the slowest stage of the CLI.
inline size_t kraken_line_bound(const HitRuns &runs, const bseq1_t &bs) { return bs.name.size() + 64 + (size_t)runs.n * 24; }
static inline char *kraken_line_raw(char *w, const HitRuns &runs, tax_t taxon, u32 ambig_count, u32 missing_count, const bseq1_t &bs)
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

void append_kraken_classification(const HitRuns &runs, tax_t taxon, u32 ambig_count, u32 missing_count,
                                  const bseq1_t &bs, std::string &bks)
{
    const size_t at = bks.size();
    bks.resize(at + kraken_line_bound(runs, bs));
    char *w = kraken_line_raw(&bks[at], runs, taxon, ambig_count, missing_count, bs);
    bks.resize((size_t)(w - bks.data()));
}

void append_fastq_classification(const HitRuns &runs, tax_t taxon, u32 ambig_count, u32 missing_count,
                                 const bseq1_t *bs, std::string &bks, int verbose, int is_paired)
{
    // classifier.h:72-108, reproduced as written (the record name carries no '@'; with verbose == 0 the
    // trailing tab of the comment becomes the newline; mate 2 repeats the comment and adds its own newline).
    bks += bs->name; bks.push_back(' ');
    const size_t cms = bks.size();
    bks.push_back(taxon == 0 ? 'U' : 'C'); bks.push_back('\t');
    put_unsigned(bks, taxon); bks.push_back('\t');
    put_signed(bks, bs->l_seq()); bks.push_back('\t');
    append_counts(missing_count, 'M', bks);
    append_counts(ambig_count, 'A', bks);
    if (verbose) append_taxa_runs(taxon, runs.tax, runs.len, runs.n, bks); else bks.back() = '\n';
    const size_t cme = bks.size();
    bks += bs->seq; bks += "\n+\n"; bks += bs->qual.empty() ? bs->seq : bs->qual; bks.push_back('\n');
    if (is_paired) {
        const bseq1_t *m2 = bs + 1;
        bks += m2->name; bks.push_back(' ');
        bks.append(bks, cms, cme - cms); bks.push_back('\n');
        bks += m2->seq; bks += "\n+\n"; bks += m2->qual.empty() ? m2->seq : m2->qual; bks.push_back('\n');
    }
}

namespace {
struct OwnedRuns {                                               // the reference's `taxa` vector -> runs
    std::vector<u32> tax, len;
    explicit OwnedRuns(const std::vector<tax_t> &taxa)
    {
        for (size_t i = 0; i < taxa.size();) {
            size_t j = i;
            while (j < taxa.size() && taxa[j] == taxa[i]) ++j;
            tax.push_back(taxa[i]); len.push_back((u32)(j - i));
            i = j;
        }
    }
    HitRuns view() const { return HitRuns{tax.data(), len.data(), (u32)tax.size()}; }
};
}  // namespace

void append_kraken_classification(const std::vector<tax_t> &taxa, tax_t taxon, u32 ambig_count, u32 missing_count,
                                  const bseq1_t &bs, std::string &bks)
{
    append_kraken_classification(OwnedRuns(taxa).view(), taxon, ambig_count, missing_count, bs, bks);
}

void append_fastq_classification(const std::vector<tax_t> &taxa, tax_t taxon, u32 ambig_count, u32 missing_count,
                                 const bseq1_t *bs, std::string &bks, int verbose, int is_paired)
{
    append_fastq_classification(OwnedRuns(taxa).view(), taxon, ambig_count, missing_count, bs, bks, verbose, is_paired);
}

int usable_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (std::FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
            const long q = std::atol(quota);
            if (q > 0) n = (int)std::min<long>(n, (q + period - 1) / period);
        }
        std::fclose(f);
    }
    return std::max(1, n);
}

// ---------------------------------------------------------------------------------------------- classifier
ClassifierGeneric::ClassifierGeneric(const Database &db, const std::vector<u32> &parent, int device, int num_threads,
                                     bool emit_all, bool emit_fastq, bool emit_kraken, bool canonicalize, int layout)
    : ClassifierGeneric(db, parent, std::vector<int>{device}, num_threads, emit_all, emit_fastq, emit_kraken, canonicalize, layout) {}

ClassifierGeneric::ClassifierGeneric(const Database &db, const std::vector<u32> &parent, const std::vector<int> &devices, int num_threads,
                                     bool emit_all, bool emit_fastq, bool emit_kraken, bool canonicalize, int layout)
    : k_(db.k_), nt_(num_threads > 0 ? num_threads : 1)
{
    if (devices.empty()) die("no device given");
    if (emit_all) output_flag_ |= EMIT_ALL;
    if (emit_fastq) output_flag_ |= FASTQ;
    if (emit_kraken) output_flag_ |= KRAKEN;
    c_ = k_;
    for (u16 g : db.s_) c_ += g;
    try {
        for (int d : devices) {
            bns_ctx *c = nullptr;
            chk(nullptr, bns_create(d, &c), "bns_create");
            ctxs_.push_back(c);
            devices_.push_back(d);
            // bin/bonsai.cpp:152: Spacer(db.k_, wsz = db.k_, db.s_): classify looks up every k-mer.
            // A spaced seed takes the intended for_each_uncanon_spaced path (deviation from SURVEY F7, see README).
            chk(c, bns_set_encoder(c, db.k_, db.s_.empty() ? nullptr : db.s_.data(), canonicalize ? 1 : 0, 1), "bns_set_encoder");
        }
        ctx_ = ctxs_[0];
        // one PCIe upload of the db, RCCL broadcast over xGMI to the other devices, one re-hash per device
        chk(ctx_, bns_load_table_multi(ctxs_.data(), (int)ctxs_.size(), db.db_.n_buckets, db.db_.flags.data(), db.db_.keys.data(),
                                       db.db_.vals.data(), layout), "bns_load_table_multi");
        for (bns_ctx *c : ctxs_) chk(c, bns_load_taxonomy(c, parent.data(), (u32)parent.size()), "bns_load_taxonomy");
        for (size_t i = 1; i < ctxs_.size(); ++i) shards_.emplace_back(new Shard());
    } catch (...) {
        for (bns_ctx *c : ctxs_) bns_destroy(c);
        ctxs_.clear(); ctx_ = nullptr;
        throw;
    }
}

ClassifierGeneric::~ClassifierGeneric()
{
    work_.res.release(); work_.first.release();              // (page-locked memory goes back while the contexts still exist)
    for (auto &sh : shards_) sh->res.release();
    for (bns_ctx *c : ctxs_) bns_destroy(c);
}

int bind_near_devices(const std::vector<int> &devices)
{
    cpu_set_t cur, want;
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return 0;
    CPU_ZERO(&want);
    for (int d : devices) {
        char id[64] = {0};
        if (bns_device_pci_bus_id(d, id, (int)sizeof(id)) != BNS_OK) return 0;
        for (char *p = id; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
        std::ifstream f(std::string("/sys/bus/pci/devices/") + id + "/local_cpulist");
        std::string list;
        if (!f || !std::getline(f, list) || list.empty()) return 0;
        for (size_t at = 0; at < list.size();) {                  // "0-63,128-191"
            char *end = nullptr;
            const long lo = std::strtol(list.c_str() + at, &end, 10);
            long hi = lo;
            if (end == list.c_str() + at) return 0;
            if (*end == '-') hi = std::strtol(end + 1, &end, 10);
            for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) if (c >= 0) CPU_SET((int)c, &want);
            at = (size_t)(end - list.c_str());
            if (at < list.size() && list[at] == ',') ++at; else if (at < list.size()) return 0;
        }
    }
    CPU_AND(&want, &want, &cur);
    const int n = CPU_COUNT(&want);
    if (n == 0 || n == CPU_COUNT(&cur)) return 0;
    return sched_setaffinity(0, sizeof(want), &want) == 0 ? n : 0;
}

std::vector<int> parse_devices(const char *spec)
{
    std::vector<int> out;
    const std::string s = spec ? spec : "";
    if (s == "all") {
        int n = bns_device_count();
        if (n < 1) die("no usable GPU");
        for (int i = 0; i < n; ++i) out.push_back(i);
        return out;
    }
    size_t i = 0;
    auto num = [&]() -> int {
        if (i >= s.size() || !std::isdigit((unsigned char)s[i])) die("bad device list '" + s + "' (expected e.g. 0, 0-7, 0,2,5 or all)");
        int v = 0;
        while (i < s.size() && std::isdigit((unsigned char)s[i])) v = v * 10 + (s[i++] - '0');
        return v;
    };
    while (i < s.size()) {
        const int a = num();
        int b = a;
        if (i < s.size() && s[i] == '-') { ++i; b = num(); }
        if (b < a) die("bad device range in '" + s + "'");
        for (int d = a; d <= b; ++d) out.push_back(d);
        if (i < s.size()) { if (s[i] != ',') die("bad device list '" + s + "'"); ++i; if (i == s.size()) die("bad device list '" + s + "'"); }
    }
    if (out.empty()) die("empty device list");
    return out;
}

char *PinnedBuf::reserve(bns_ctx *c, size_t bytes)
{
    if (bytes <= cap) return p;
    const size_t want = std::max(bytes, 2 * cap);              // (page-locking is 0.45 ms per MiB and freeing drains the device: grow in few steps)
    release();
    ctx = c;
    void *q = nullptr;
    pinned = bns_host_alloc(c, want, &q) == BNS_OK && q;
    if (!pinned) q = std::malloc(want);
    if (!q) die("out of host memory");
    p = static_cast<char *>(q); cap = want;
    return p;
}

void PinnedBuf::release()
{
    if (p) { if (pinned) bns_host_free(ctx, p); else std::free(p); }
    p = nullptr; cap = 0; pinned = false;
}

PinnedBuf::~PinnedBuf() { release(); }

namespace {
double tnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
}  // namespace

// First half of classify_seqs: gather the chunk's sequences into one buffer and make the ONE C-ABI call that replaces the
// kt_forpool fan-out of classifier.h:275 (with the hit stream already run-length encoded on the device when the output
// prints it).  Everything the formatter needs ends up in r.
namespace {
// one device's share of a chunk: reads [first, first + n) -> results into r (whose vectors are sized here).
// The sequences are views scattered over the file text; instead of gathering them into one ASCII buffer (what round 2 did: a
// copy of every base, then 150 bytes per read over PCIe) they are PACKED where they lie into the page-locked staging buffer --
// 2 bits per base, bns_pack_reads_ptrs on `copy_threads` threads -- and handed to the packed entry point: 40 bytes per read up.
// pack_chunk is the host half (packs into r's own page-locked buffers), call_chunk the GPU call; process_dataset runs them on two
// threads per device so that chunk i + 1 is packed while chunk i is on the GPU.
void pack_chunk(ClassifierGeneric &c, bns_ctx *ctx, const bseq1_t *bs, unsigned n, int is_paired, ChunkResult &r, unsigned copy_threads)
{
    const unsigned inc = is_paired ? 2 : 1;
    r.n = n; r.is_paired = is_paired;
    r.want_runs = c.get_emit_kraken() != 0;
    r.taxon_only = false;
    const unsigned n_units = n / inc;
    r.taxon.resize(ctx, n_units); r.missing.resize(ctx, n_units); r.ambig.resize(ctx, n_units); r.n_hits.resize(ctx, n_units);
    r.run_tax.clear(); r.run_len.clear();
    if (r.want_runs) { r.run_start.resize(ctx, n_units); r.n_runs.resize(ctx, n_units); }
    r.n_bad = 0; r.t_pack = r.t_call = r.t_copy = 0;
    if (!n) return;
    r.offsets.resize(ctx, n + 1);
    std::vector<const char *> &ptrs = r.seq_ptrs;
    std::vector<u32> &lens = r.seq_lens;
    ptrs.resize(n); lens.resize(n);
    u64 total = 0;
    for (unsigned i = 0; i < n; ++i) { ptrs[i] = bs[i].seq.data(); lens[i] = (u32)bs[i].seq.size(); total += bs[i].seq.size(); }
    const u64 n_words = bns_packed_words(total, n);
    u64 *words = reinterpret_cast<u64 *>(r.words.reserve(ctx, (size_t)n_words * 8 + 8));
    u64 n_bad = 0;
    if (r.bad_word.size() < 4096) { r.bad_word.resize(4096); r.bad_mask.resize(4096); }
    const double t_p0 = tnow();
    int rc = bns_pack_reads_ptrs(ptrs.data(), lens.data(), n, r.offsets.data(), words, r.bad_word.data(), r.bad_mask.data(), r.bad_word.size(), &n_bad,
                                 (int)std::max(1u, copy_threads));
    if (rc != BNS_OK && n_bad > r.bad_word.size()) {           // more words with an invalid base than there was room for: once more
        r.bad_word.resize((size_t)n_bad); r.bad_mask.resize((size_t)n_bad);
        rc = bns_pack_reads_ptrs(ptrs.data(), lens.data(), n, r.offsets.data(), words, r.bad_word.data(), r.bad_mask.data(), r.bad_word.size(), &n_bad,
                                 (int)std::max(1u, copy_threads));
    }
    chk(ctx, rc, "bns_pack_reads_ptrs");
    r.n_bad = n_bad;
    r.t_pack = tnow() - t_p0;
}

void call_chunk(bns_ctx *ctx, ChunkResult &r)
{
    const unsigned n = r.n;
    if (!n) return;
    const u64 *words = reinterpret_cast<const u64 *>(r.words.p);
    const double t_p1 = tnow();
    if (r.want_runs) {
        const u32 *run_tax = nullptr, *run_len = nullptr;
        u64 n_runs_total = 0;
        chk(ctx, bns_classify_batch_packed_runs(ctx, words, r.bad_word.data(), r.bad_mask.data(), r.n_bad, r.offsets.data(), n, r.is_paired, r.taxon.data(),
                                                r.missing.data(), r.ambig.data(), r.n_hits.data(), r.run_start.data(), r.n_runs.data(), &run_tax, &run_len,
                                                &n_runs_total), "bns_classify_batch_packed_runs");
        const double t_c = tnow();
        r.run_tax.assign(run_tax, run_tax + n_runs_total);           // the context's buffers only live until its next call
        r.run_len.assign(run_len, run_len + n_runs_total);
        r.t_copy = tnow() - t_c;
        r.t_call = t_c - t_p1;
    } else if (r.taxon_only) {                                   // (-K -F: only the tally and the -b file read the results: the taxon alone comes back)
        chk(ctx, bns_classify_batch_packed(ctx, words, r.bad_word.data(), r.bad_mask.data(), r.n_bad, r.offsets.data(), n, r.is_paired, r.taxon.data(),
                                           nullptr, nullptr, nullptr, nullptr), "bns_classify_batch_packed");
        r.t_call = tnow() - t_p1; r.t_copy = 0;
    } else {
        chk(ctx, bns_classify_batch_packed(ctx, words, r.bad_word.data(), r.bad_mask.data(), r.n_bad, r.offsets.data(), n, r.is_paired, r.taxon.data(),
                                           r.missing.data(), r.ambig.data(), r.n_hits.data(), nullptr), "bns_classify_batch_packed");
        r.t_call = tnow() - t_p1; r.t_copy = 0;
    }
}

void classify_on(ClassifierGeneric &c, bns_ctx *ctx, const bseq1_t *bs, unsigned n, int is_paired, ChunkResult &r, unsigned copy_threads)
{
    pack_chunk(c, ctx, bs, n, is_paired, r, copy_threads);
    call_chunk(ctx, r);
}
}  // namespace

void classify_chunk(ClassifierGeneric &c, const bseq1_t *bs, unsigned n, int is_paired, ChunkResult &r)
{
    const unsigned inc = is_paired ? 2 : 1;
    n -= n % inc;
    r.n = n; r.is_paired = is_paired;
    r.want_runs = c.get_emit_kraken() != 0;                  // run strings are only printed in Kraken / verbose FASTQ mode
    if (!n) return;
    if (c.ctxs_.size() > 1) {
        // reads shard: contiguous unit ranges, one per device, classified concurrently (one host thread per device:
        // calls on a context are serialised, contexts are independent); results concatenated in input order
        const double t0 = tnow();
        const unsigned G = (unsigned)c.ctxs_.size(), n_units = n / inc;
        std::vector<ChunkResult *> part(G);
        std::vector<unsigned> lo(G + 1);
        for (unsigned g = 0; g <= G; ++g) lo[g] = (unsigned)((u64)n_units * g / G);
        std::vector<std::thread> th;
        std::vector<std::string> errs(G);
        ChunkResult &first = c.work_.first;
        for (unsigned g = 0; g < G; ++g) {
            part[g] = g == 0 ? &first : &c.shards_[g - 1]->res;
            th.emplace_back([&, g] {
                try {
                    classify_on(c, c.ctxs_[g], bs + (size_t)lo[g] * inc, (lo[g + 1] - lo[g]) * inc, is_paired, *part[g], 1u);
                } catch (const std::exception &e) { errs[g] = e.what(); }
            });
        }
        for (auto &t : th) t.join();
        for (auto &e : errs) if (!e.empty()) die(e);
        r.taxon.resize(c.ctx_, n_units); r.missing.resize(c.ctx_, n_units); r.ambig.resize(c.ctx_, n_units); r.n_hits.resize(c.ctx_, n_units);
        if (r.want_runs) { r.run_start.resize(c.ctx_, n_units); r.n_runs.resize(c.ctx_, n_units); }
        r.run_tax.clear(); r.run_len.clear();
        for (unsigned g = 0; g < G; ++g) {
            const ChunkResult &p = *part[g];
            const u64 base = r.run_tax.size();
            const size_t at = lo[g], cnt = lo[g + 1] - lo[g];
            std::memcpy(r.taxon.data() + at, p.taxon.data(), cnt * 4);
            std::memcpy(r.missing.data() + at, p.missing.data(), cnt * 4);
            std::memcpy(r.ambig.data() + at, p.ambig.data(), cnt * 4);
            std::memcpy(r.n_hits.data() + at, p.n_hits.data(), cnt * 4);
            if (r.want_runs) {
                for (size_t i = 0; i < cnt; ++i) r.run_start[at + i] = p.run_start[i] + base;
                std::memcpy(r.n_runs.data() + at, p.n_runs.data(), cnt * 4);
                r.run_tax.insert(r.run_tax.end(), p.run_tax.begin(), p.run_tax.end());
                r.run_len.insert(r.run_len.end(), p.run_len.begin(), p.run_len.end());
            }
        }
        c.work_.t_gpu += tnow() - t0;
        return;
    }
    const double t0 = tnow();
    classify_on(c, c.ctx_, bs, n, is_paired, r, (unsigned)std::max(1, c.nt_));
    c.work_.t_gpu += tnow() - t0;
}

// Second half: the result text of the chunk (classifier.h:277-286) and the classified / unclassified tally.  The text is left in
// c.work_.parts[0 .. return value), one piece per formatting thread, in input order: process_dataset writes the pieces as they
// are (appending them to one string first was a quarter of the formatter's time); format_chunk() is the appending form.
unsigned format_chunk_parts(ClassifierGeneric &c, const bseq1_t *bs, const ChunkResult &r, std::vector<ClassifierGeneric::Work::Part> *into, unsigned skip_first)
{
    if (!r.n) return 0;
    const double t0 = tnow();
    const unsigned inc = r.is_paired ? 2 : 1, n_units = r.n / inc;
    const unsigned nt = (unsigned)std::max(1, std::min<int>(c.nt_, (int)(n_units / 4096 + 1)));
    // (one output buffer per thread, each header on a cache line of its own: with the headers packed in a vector every append
    // of one thread -- it updates the length -- invalidated its neighbours' lines, and -p 4 formatted SLOWER than -p 1)
    std::vector<ClassifierGeneric::Work::Part> &parts = into ? *into : c.work_.parts;
    if (parts.size() < nt) parts.resize(nt);
    std::vector<u64> ncls(nt * 2, 0);
    const bool kraken_only = !c.get_emit_fastq() && c.get_emit_kraken();
    parallel_units(nt, n_units, [&](unsigned lo, unsigned hi, unsigned t) {
        ClassifierGeneric::Work::Part &part = parts[t];
        part.n = 0;
        part.s.clear();
        lo = std::max(lo, skip_first); hi = std::max(hi, lo);    // (units another path has printed already)
        u64 n_cls[2] = {0, 0};                                   // (thread-local: ncls' entries share cache lines)
        if (kraken_only) {                                           // the usual output: raw buffer, one capacity check per record
            part.ensure((size_t)(hi - lo) * 48 + 4096);
            for (unsigned u = lo; u < hi; ++u) {
                const bseq1_t &b = bs[u * inc];
                if (u + 8 < hi) __builtin_prefetch(bs[(size_t)(u + 8) * inc].name.data());   // (the name is in file text last touched by the parser)
                ++n_cls[r.taxon[u] == 0];
                if (!(c.get_emit_all() || r.taxon[u])) continue;
                const HitRuns runs = r.want_runs ? HitRuns{r.run_tax.data() + r.run_start[u], r.run_len.data() + r.run_start[u], r.n_runs[u]}
                                                 : HitRuns{nullptr, nullptr, 0};
                const size_t bound = kraken_line_bound(runs, b);
                if (part.n + bound > part.cap) part.ensure(std::max(part.n + bound, part.cap * 2));
                part.n = (size_t)(kraken_line_raw(part.p + part.n, runs, r.taxon[u], r.ambig[u], r.missing[u], b) - part.p);
            }
        } else {
            std::string &out = part.s;
            for (unsigned u = lo; u < hi; ++u) {
                ++n_cls[r.taxon[u] == 0];
                if (!c.get_emit_fastq() || !(c.get_emit_all() || r.taxon[u])) continue;    // (no text: the records are not touched -- a container chunk has none)
                const HitRuns runs = r.want_runs ? HitRuns{r.run_tax.data() + r.run_start[u], r.run_len.data() + r.run_start[u], r.n_runs[u]}
                                                 : HitRuns{nullptr, nullptr, 0};
                append_fastq_classification(runs, r.taxon[u], r.ambig[u], r.missing[u], &bs[u * inc], out, c.get_emit_kraken(), r.is_paired);
            }
        }
        ncls[t * 2] = n_cls[0]; ncls[t * 2 + 1] = n_cls[1];
    });
    {
        static std::mutex tally_mu;                              // (process_dataset formats on two threads)
        std::lock_guard<std::mutex> lk(tally_mu);
        for (unsigned t = 0; t < nt; ++t) { c.classified_[0] += ncls[t * 2]; c.classified_[1] += ncls[t * 2 + 1]; }
        c.work_.t_format += tnow() - t0;
    }
    return nt;
}

void format_chunk(ClassifierGeneric &c, const bseq1_t *bs, const ChunkResult &r, std::string &cks)
{
    const unsigned nt = format_chunk_parts(c, bs, r, nullptr);
    for (unsigned t = 0; t < nt; ++t) {
        const ClassifierGeneric::Work::Part &part = c.work_.parts[t];
        cks.append(part.p, part.n);
        cks += part.s;
    }
}

void classify_seqs(ClassifierGeneric &c, bseq1_t *bs, std::string &cks, unsigned n, int is_paired)
{
    classify_chunk(c, bs, n, is_paired, c.work_.res);
    format_chunk(c, bs, c.work_.res, cks);
}

// Where a plain FASTA / FASTQ file can be cut so that every stretch, parsed on its own, gives exactly the records the whole file
// gives there.  A cut is the start of a line that (a) begins a record of the file's kind -- for FASTQ two consecutive records in the
// strict four-line form (header, one sequence line, '+' line, a quality line of the sequence's length, then another '@' header: a
// quality line that merely starts with '@' is followed by a header, not by a sequence, and fails), for FASTA a '>' line followed by
// a sequence line in a neighbourhood without '+' lines -- and (b) is where the parser of the stretch before it arrives between
// two records, which process_dataset checks after the fact (that stretch must end cleanly on a complete record; if it does, its
// parser read every byte before the cut exactly as the sequential parser would have, and that one would have started its next
// record at the cut).  Nothing is cut when the file is gzip, a pipe, of another kind, or no such line is found near a target.
std::vector<u64> find_cut_points(const char *path, u64 seg_bytes)
{
    std::vector<u64> cuts;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return cuts;
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    const off_t sz = ::lseek(fd, 0, SEEK_END);
    if (sz <= 0 || seg_bytes == 0 || (u64)sz < 2 * seg_bytes) return cuts;
    unsigned char first[2] = {0, 0};
    if (::pread(fd, first, 2, 0) != 2 || (first[0] == 0x1f && first[1] == 0x8b)) return cuts;
    const bool fastq = first[0] == '@';
    if (!fastq && first[0] != '>') return cuts;
    const size_t W = 1u << 20;
    std::vector<char> buf(W);
    for (u64 target = seg_bytes; target + seg_bytes / 2 < (u64)sz; target += seg_bytes) {
        const u64 at = std::max<u64>(target, cuts.empty() ? 0 : cuts.back() + 1);
        const ssize_t n = ::pread(fd, buf.data(), W, (off_t)at);
        if (n <= 0) break;
        const char *b = buf.data(), *e = b + n;
        const bool to_eof = at + (u64)n == (u64)sz;
        auto line_end = [&](const char *p) -> const char * { return static_cast<const char *>(std::memchr(p, '\n', (size_t)(e - p))); };
        // FASTQ: a strict four-line record at p, followed by a header (or the end of the file); returns the start of what follows
        auto strict_record = [&](const char *p) -> const char * {
            if (p >= e || *p != '@') return nullptr;
            const char *h = line_end(p); if (!h) return nullptr;
            const char *s = h + 1; if (s >= e || *s == '@' || *s == '>' || *s == '+' || *s == '\n' || *s == '\r') return nullptr;
            const char *sn = line_end(s); if (!sn) return nullptr;
            const char *pl = sn + 1; if (pl >= e || *pl != '+') return nullptr;
            const char *pn = line_end(pl); if (!pn) return nullptr;
            const char *q = pn + 1;
            const char *qn = line_end(q); if (!qn) return nullptr;
            if (qn - q != sn - s) return nullptr;
            return qn + 1;
        };
        bool plus_line = false;                                      // FASTA: any line of the window that starts with '+'
        if (!fastq) for (const char *p = b; p < e; ) { const char *nl = line_end(p); if (!nl) break; p = nl + 1; if (p < e && *p == '+') { plus_line = true; break; } }
        if (!fastq && plus_line) continue;
        for (const char *p = line_end(b); p && p + 1 < e; p = line_end(p + 1)) {
            const char *c0 = p + 1;                                  // first character of a line
            if (fastq) {
                const char *r2 = strict_record(c0);
                if (!r2) continue;
                const char *r3 = (r2 == e && to_eof) ? r2 : strict_record(r2);
                if (!r3 || !(r3 < e ? *r3 == '@' : to_eof)) continue;
            } else {
                if (*c0 != '>') continue;
                const char *h = line_end(c0);
                if (!h || h + 1 >= e) continue;
                const char s0 = h[1];
                if (s0 == '>' || s0 == '@' || s0 == '+' || s0 == '\n' || s0 == '\r') continue;
            }
            cuts.push_back(at + (u64)(c0 - b));
            break;
        }
    }
    return cuts;
}

// The same test on text that is already in memory (a BGZF input's inflated blocks): the offset of a line in [b, b + n) that begins
// a record the way find_cut_points wants it, with everything the test looks at inside the window; -1 when there is none.
static long find_record_start(const char *b, size_t n, bool fastq)
{
    const char *e = b + n;
    auto line_end = [&](const char *p) -> const char * { return p < e ? static_cast<const char *>(std::memchr(p, '\n', (size_t)(e - p))) : nullptr; };
    auto strict_record = [&](const char *p) -> const char * {
        if (p >= e || *p != '@') return nullptr;
        const char *h = line_end(p); if (!h) return nullptr;
        const char *s = h + 1; if (s >= e || *s == '@' || *s == '>' || *s == '+' || *s == '\n' || *s == '\r') return nullptr;
        const char *sn = line_end(s); if (!sn) return nullptr;
        const char *pl = sn + 1; if (pl >= e || *pl != '+') return nullptr;
        const char *pn = line_end(pl); if (!pn) return nullptr;
        const char *q = pn + 1;
        const char *qn = line_end(q); if (!qn) return nullptr;
        if (qn - q != sn - s) return nullptr;
        return qn + 1;
    };
    if (!fastq) {                                                    // FASTA: no '+' line anywhere near (a FASTQ quality line may start with '>')
        const char *lim = n > (1u << 20) ? b + (1u << 20) : e;
        for (const char *p = b; p < lim; ) { const char *nl = line_end(p); if (!nl) break; p = nl + 1; if (p < e && *p == '+') return -1; }
    }
    for (const char *p = line_end(b); p && p + 1 < e; p = line_end(p + 1)) {
        const char *c0 = p + 1;
        if (fastq) {
            const char *r2 = strict_record(c0);
            if (!r2) continue;
            const char *r3 = strict_record(r2);
            if (!r3 || r3 >= e || *r3 != '@') continue;
        } else {
            if (*c0 != '>') continue;
            const char *h = line_end(c0);
            if (!h || h + 1 >= e) continue;
            const char s0 = h[1];
            if (s0 == '>' || s0 == '@' || s0 == '+' || s0 == '\n' || s0 == '\r') continue;
        }
        return (long)(c0 - b);
    }
    return -1;
}

// ---- ChunkSource: bseq_read chunks of one or two files, in input order ---------------------------------------------------------
struct ChunkSource::Impl {
    std::string fq1;
    unsigned chunk_size = 0, P = 1;
    std::unique_ptr<SeqReader> r1, r2;                         // sequential mode (and the fallback's reader)
    // parallel mode
    struct Segment { u64 begin = 0, end = ~0ULL; std::deque<std::unique_ptr<ReadChunk>> chunks; bool done = false, clean = false; };
    std::vector<Segment> segs;
    bool fastq_file = false;
    size_t cur_seg = 0;                                        // the stretch next() hands out
    size_t n_stretches = 1;                                    // (as planned: a fallback does not change it)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> parsers;
    std::vector<std::unique_ptr<ReadChunk>> spare;
    bool stop = false, fell_back = false;
    std::string error;
    double t_parse = 0, t_blocked = 0;
    // ---- a BGZF file on several parser threads.  A gzip file has no byte ranges to hand to readers of their own, but its text
    // arrives as blocks in file order (the reader's inflaters, CPU and GPU): a distributor thread takes them from ONE reader (the
    // feeder), closes a stretch every `stretch_blocks` blocks at a record start found in the next block's text (find_record_start:
    // the bytes in front of it go to the closing stretch as a small block of their own), and parser threads parse whole stretches
    // through readers over in-memory blocks.  Checked and handed out like the stretches of a plain file; a stretch that does not
    // end between two records is parsed again, with everything behind it, by one reader from where it began.
    bool bgz_par = false;
    std::unique_ptr<SeqReader> feeder;
    size_t stretch_blocks = 16;
    struct MemSeg {
        std::vector<std::shared_ptr<TextBlock>> blocks;
        std::vector<std::pair<size_t, size_t>> span;           // begin / end of every block as the stretch got it (parsing moves them)
        std::deque<std::unique_ptr<ReadChunk>> chunks;
        bool done = false, clean = false, last = false;
        void push(std::shared_ptr<TextBlock> b) { span.emplace_back(b->begin, b->end); blocks.push_back(std::move(b)); }
    };
    std::vector<std::unique_ptr<MemSeg>> msegs;                // complete stretches, by index (under mu)
    std::unique_ptr<MemSeg> filling;                           // the distributor's (only the distributor touches it while it runs)
    size_t next_parse = 0;
    bool dist_done = false, feeder_ended = false;
    int file_kind = -1;                                        // 1 FASTQ, 0 FASTA, 2 neither (no cuts), -1 not seen yet
    std::thread distributor;
    void distribute()
    {
        try {
            filling.reset(new MemSeg);
            const long force_bad = std::getenv("BNS_BGZF_FORCE_BAD_CUT") ? std::atol(std::getenv("BNS_BGZF_FORCE_BAD_CUT")) : -1;   // (tests: a cut inside a record at that stretch)
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return msegs.size() < cur_seg + 2 * (size_t)P + 1 || stop; });
                    if (stop) return;
                }
                auto b = feeder->take_block();
                if (!b) { feeder_ended = true; break; }
                if (file_kind < 0 && b->size()) file_kind = b->data()[0] == '@' ? 1 : b->data()[0] == '>' ? 0 : 2;
                if (filling->blocks.size() >= stretch_blocks && file_kind != 2 && file_kind >= 0) {
                    long c = find_record_start(b->data(), b->size(), file_kind == 1);
                    if (force_bad >= 0 && (long)msegs.size() == force_bad && b->size() > 200) c = 100;
                    if (c > 0) {
                        auto tail = std::make_shared<TextBlock>(SeqReader::Impl::HEAD + (size_t)c + 8);
                        tail->begin = SeqReader::Impl::HEAD;
                        std::memcpy(tail->raw() + tail->begin, b->data(), (size_t)c);
                        tail->end = tail->begin + (size_t)c;
                        filling->push(std::move(tail));
                        b->begin += (size_t)c;
                        std::lock_guard<std::mutex> lk(mu);
                        msegs.push_back(std::move(filling));
                        filling.reset(new MemSeg);
                        cv.notify_all();
                    }
                }
                filling->push(std::move(b));
            }
            std::lock_guard<std::mutex> lk(mu);
            filling->last = true;
            msegs.push_back(std::move(filling));
            dist_done = true;
            cv.notify_all();
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            if (error.empty()) error = e.what();
            stop = true;
            cv.notify_all();
        }
    }
    void parse_mem_stretches()
    {
        try {
            for (;;) {
                MemSeg *sg = nullptr;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return next_parse < msegs.size() || dist_done || stop; });
                    if (stop || next_parse >= msegs.size()) return;
                    sg = msegs[next_parse++].get();
                }
                SeqReader rd(std::deque<std::shared_ptr<TextBlock>>(sg->blocks.begin(), sg->blocks.end()), nullptr);
                bool any = false, last_has_qual = false, last_empty = false;
                double tp = 0;
                for (;;) {
                    auto c = take_spare();
                    const double t0 = tnow();
                    const int got = bseq_read((int)chunk_size, rd, nullptr, *c);
                    tp += tnow() - t0;
                    if (got <= 0) break;
                    const bseq1_t &last = c->recs[c->recs.size() - 1];
                    any = true; last_has_qual = !last.qual.empty(); last_empty = last.seq.empty();
                    std::lock_guard<std::mutex> lk(mu);
                    if (stop) return;
                    sg->chunks.push_back(std::move(c));
                }
                // (as for a plain file's stretches: parse_stretches)
                const bool clean = sg->last || (rd.last_status() == -1 && any && (file_kind == 1 ? last_has_qual : (!last_has_qual && !last_empty)));
                std::lock_guard<std::mutex> lk(mu);
                t_parse += tp;
                sg->clean = clean;
                sg->done = true;
                cv.notify_all();
                if (!clean) return;
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            if (error.empty()) error = e.what();
            stop = true;
            cv.notify_all();
        }
    }
    // two files, two parser threads: each file's records in batches of n_per_half, interleaved by next()
    bool paired_par = false, first_done = false;
    size_t n_per_half = 0;
    // trunc: the batch at the BACK of q ended early because a truncated record followed it (the record is consumed); the half's
    // parser thread has then stopped, and next() goes over to merge mode: one thread, records taken from what the parsers left
    // queued and then straight from the two readers, paired as bseq_read pairs them
    struct Half { std::deque<std::unique_ptr<ReadChunk>> q; bool done = false, trunc = false; size_t cursor = 0; } half[2];
    bool merge_mode = false;
    // next record of file t in merge mode: >= 0 its length, -1 end of file, -2 a truncated record (dropped)
    int half_next(unsigned t, bseq1_t &rec, ReadChunk &out)
    {
        Half &h = half[t];
        while (!h.q.empty()) {
            ReadChunk &b = *h.q.front();
            if (h.cursor < b.recs.size()) {
                if (h.cursor == 0) out.blocks.insert(out.blocks.end(), b.blocks.begin(), b.blocks.end());   // (the views point into the batch's text)
                rec = b.recs[h.cursor++];
                return (int)rec.seq.size();
            }
            const bool last = h.q.size() == 1;
            spare.push_back(std::move(h.q.front()));
            h.q.pop_front();
            h.cursor = 0;
            if (last && h.trunc) { h.trunc = false; return -2; }
        }
        if (h.trunc) { h.trunc = false; return -2; }
        if (h.done) return -1;
        SeqReader &rd = t == 0 ? *r1 : *r2;
        const int rc = rd.read(rec, out);
        if (rc >= 0) trim_readno(rec.name);
        if (rc == -1) h.done = true;
        return rc;
    }

    void parse_half(unsigned t)
    {
        SeqReader &rd = t == 0 ? *r1 : *r2;
        try {
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return half[t].q.size() < 3 || stop; });
                    if (stop) return;
                }
                auto c = take_spare();
                c->clear();
                c->recs.reserve(n_per_half);
                long size = 0;
                const double t0 = tnow();
                rd.fill(std::numeric_limits<long>::max(), *c, size, n_per_half);
                bool ended = false, truncated = false;
                if (c->recs.size() < n_per_half) {                   // the end of the file, or a truncated record
                    bseq1_t tmp;
                    const int rc = rd.read(tmp, *c);
                    // a truncated record (rc == -2): the reference drops it and carries on with the mates shifted
                    // (kseq_declare.h:112-145); side by side the two files cannot reproduce that, so this thread stops here and
                    // next() pairs the rest on one thread (merge mode)
                    truncated = rc != -1;
                    ended = true;
                }
                RecVec::publish();
                const double dt = tnow() - t0;
                std::lock_guard<std::mutex> lk(mu);
                t_parse += dt;
                if (!c->recs.empty() || truncated) half[t].q.push_back(std::move(c));
                if (truncated) half[t].trunc = true; else if (ended) half[t].done = true;
                cv.notify_all();
                if (ended) return;
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            if (error.empty()) error = e.what();
            stop = true;
            cv.notify_all();
        }
    }

    std::unique_ptr<ReadChunk> take_spare()
    {
        std::unique_ptr<ReadChunk> c;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!spare.empty()) { c = std::move(spare.back()); spare.pop_back(); }
        }
        if (!c) c = std::make_unique<ReadChunk>();
        return c;
    }
    void parse_stretches(unsigned t)
    {
        try {
            for (size_t i = t; i < segs.size(); i += P) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return i < cur_seg + 2 * (size_t)P || stop; });     // at most 2 P stretches parsed ahead of the consumer
                    if (stop) return;
                }
                SeqReader rd(fq1.c_str(), 0, segs[i].begin, segs[i].end);
                bool any = false, last_has_qual = false, last_empty = false;
                double tp = 0;
                for (;;) {
                    auto c = take_spare();
                    const double t0 = tnow();
                    const int got = bseq_read((int)chunk_size, rd, nullptr, *c);
                    tp += tnow() - t0;
                    if (got <= 0) break;
                    const bseq1_t &last = c->recs[c->recs.size() - 1];
                    any = true; last_has_qual = !last.qual.empty(); last_empty = last.seq.empty();
                    std::lock_guard<std::mutex> lk(mu);
                    if (stop) return;
                    segs[i].chunks.push_back(std::move(c));
                }
                // The stretch must have ended between two records for the next one to begin where the sequential parser would.  FASTQ: its
                // last record is complete (a cut inside a header, sequence or quality line leaves one without quality, or truncated).
                // FASTA: a line that starts with '>' ends the record before it whatever that was, so the cut itself is the guarantee;
                // what can be seen here is a header cut short (a record without sequence).
                const bool clean = i + 1 == segs.size() ||
                                   (rd.last_status() == -1 && any && (fastq_file ? last_has_qual : (!last_has_qual && !last_empty)));
                std::lock_guard<std::mutex> lk(mu);
                t_parse += tp; t_blocked += rd.seconds_blocked();
                segs[i].clean = clean;
                segs[i].done = true;
                cv.notify_all();
                if (!clean) return;
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            if (error.empty()) error = e.what();
            stop = true;
            cv.notify_all();
        }
    }
    void join_parsers()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : parsers) t.join();
        parsers.clear();
        if (distributor.joinable()) distributor.join();
    }
};

ChunkSource::ChunkSource(const char *fq1, const char *fq2, unsigned chunk_size, unsigned parser_threads, u64 segment_bytes,
                         const std::vector<u64> *cuts_override, u64 range_begin)
    : impl_(new Impl)
{
    Impl &m = *impl_;
    m.fq1 = fq1; m.chunk_size = chunk_size;
    std::vector<u64> cuts;
    if (!fq2 && parser_threads > 1) {
        if (!segment_bytes) segment_bytes = std::max<u64>(64ull << 20, 9ull * chunk_size);      // ~4 chunks of 150-bp FASTQ
        cuts = cuts_override ? *cuts_override : find_cut_points(fq1, segment_bytes);
        // (a plain file read from range_begin on -- the part of it the device's text parser handed back: a record boundary)
        cuts.erase(std::remove_if(cuts.begin(), cuts.end(), [&](u64 x) { return x <= range_begin; }), cuts.end());
    }
    if (cuts.empty()) {
        m.r1.reset(new SeqReader(fq1, 0, range_begin));        // (each file has its own read / inflate thread)
        if (fq2) m.r2.reset(new SeqReader(fq2));
        m.paired_par = fq2 && parser_threads > 1;              // (the parser threads start after the first chunk: it says how many pairs a chunk holds)
        if (!fq2 && parser_threads > 1 && !cuts_override && m.r1->is_bgzf() && !std::getenv("BNS_BGZF_ONE_PARSER")) {
            // a BGZF file: stretches of its inflated text blocks on the parser threads (Impl::distribute)
            m.bgz_par = true;
            m.feeder = std::move(m.r1);
            m.P = parser_threads;
            m.stretch_blocks = (size_t)std::max<u64>(1, segment_bytes / m.feeder->impl_->raw_block);
            m.distributor = std::thread([this] { impl_->distribute(); });
            for (unsigned t = 0; t < m.P; ++t) m.parsers.emplace_back([this] { impl_->parse_mem_stretches(); });
        }
        return;
    }
    { std::vector<Impl::Segment> fresh(cuts.size() + 1); m.segs.swap(fresh); }
    m.n_stretches = m.segs.size();
    for (size_t i = 0; i < m.segs.size(); ++i) { m.segs[i].begin = i ? cuts[i - 1] : range_begin; m.segs[i].end = i + 1 < m.segs.size() ? cuts[i] : ~0ULL; }
    {
        char ch = 0;
        const int f = ::open(fq1, O_RDONLY);
        if (f >= 0) { m.fastq_file = ::pread(f, &ch, 1, 0) == 1 && ch == '@'; ::close(f); }
    }
    m.P = (unsigned)std::min<size_t>(parser_threads, m.segs.size());
    for (unsigned t = 0; t < m.P; ++t) m.parsers.emplace_back([this, t] { impl_->parse_stretches(t); });
}

ChunkSource::~ChunkSource() { impl_->join_parsers(); }

size_t ChunkSource::stretches() const { return impl_->bgz_par || !impl_->msegs.empty() ? std::max<size_t>(1, impl_->msegs.size()) : impl_->n_stretches; }
bool ChunkSource::fell_back() const { return impl_->fell_back; }
double ChunkSource::parse_seconds() const { return impl_->t_parse; }
double ChunkSource::blocked_seconds() const
{
    const Impl &m = *impl_;
    return m.t_blocked + (m.r1 ? m.r1->seconds_blocked() : 0.0) + (m.r2 ? m.r2->seconds_blocked() : 0.0) + (m.feeder ? m.feeder->seconds_blocked() : 0.0);
}

void ChunkSource::recycle(std::unique_ptr<ReadChunk> c)
{
    c->clear();
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->spare.push_back(std::move(c));
}

std::unique_ptr<ReadChunk> ChunkSource::next()
{
    Impl &m = *impl_;
    if (m.paired_par && m.first_done) {
        // mates i of batch k of either file -> records 2 i and 2 i + 1 of chunk k.  A file that ends first ends the input (with
        // bseq_read's warning), as it does in the sequential reader.
        std::unique_ptr<ReadChunk> a, b;
        if (!m.merge_mode) {
            std::unique_lock<std::mutex> lk(m.mu);
            m.cv.wait(lk, [&] { return ((!m.half[0].q.empty() || m.half[0].done) && (!m.half[1].q.empty() || m.half[1].done)) || m.half[0].trunc || m.half[1].trunc || !m.error.empty(); });
            if (!m.error.empty()) die(m.error);
            if (m.half[0].trunc || m.half[1].trunc) m.merge_mode = true;
        }
        if (m.merge_mode) {
            // A truncated record turned up in one of the files.  From here on ONE thread pairs the records the way bseq_read does
            // (kseq_declare.h:112-145): a truncated record of file 1 is dropped; one of file 2 is dropped together with the file-1
            // record read for it; in both cases the chunk ends there and the mates after it stay shifted, as in the reference.
            // (One difference to -P 1, on purpose: such a record never ends the whole input, which the one-thread reader -- like
            // the reference -- does when the record happens to be the first of a chunk.)
            m.join_parsers();
            m.fell_back = true;
            auto c = m.take_spare();
            c->clear();
            long size = 0;
            bseq1_t ra, rb;
            const double t0 = tnow();
            for (;;) {
                const int r1c = m.half_next(0, ra, *c);
                if (r1c == -2) { if (size) break; continue; }
                if (r1c < 0) {
                    if (size == 0 && m.half_next(1, rb, *c) >= 0) std::fprintf(stderr, "[W::bseq_read] the 1st file has fewer sequences.\n");
                    break;
                }
                const int r2c = m.half_next(1, rb, *c);
                if (r2c < 0) {
                    std::fprintf(stderr, "[W::bseq_read] the 2nd file has fewer sequences.\n");
                    if (r2c == -2 && size == 0) continue;
                    break;
                }
                size += ra.l_seq() + rb.l_seq();
                c->recs.push_back_stream(ra); c->recs.push_back_stream(rb);
                if (size >= (long)m.chunk_size) break;
            }
            RecVec::publish();
            m.t_parse += tnow() - t0;
            if (c->recs.size() == 0) { recycle(std::move(c)); return nullptr; }
            return c;
        }
        {
            std::unique_lock<std::mutex> lk(m.mu);
            if (!m.half[0].q.empty()) { a = std::move(m.half[0].q.front()); m.half[0].q.pop_front(); }
            if (!m.half[1].q.empty()) { b = std::move(m.half[1].q.front()); m.half[1].q.pop_front(); }
            m.cv.notify_all();
        }
        const size_t na = a ? a->recs.size() : 0, nb = b ? b->recs.size() : 0, n = std::min(na, nb);
        if (na != nb) {
            std::fprintf(stderr, na > nb ? "[W::bseq_read] the 2nd file has fewer sequences.\n" : "[W::bseq_read] the 1st file has fewer sequences.\n");
            m.join_parsers();                                        // nothing after this chunk
            for (auto &h : m.half) { h.q.clear(); h.done = true; }
        }
        if (n == 0) {
            if (a) recycle(std::move(a));
            if (b) recycle(std::move(b));
            return nullptr;
        }
        auto c = m.take_spare();
        c->clear();
        c->recs.reserve(2 * n);
        for (size_t i = 0; i < n; ++i) { c->recs.push_back_stream(a->recs[i]); c->recs.push_back_stream(b->recs[i]); }
        RecVec::publish();
        c->blocks.insert(c->blocks.end(), a->blocks.begin(), a->blocks.end());      // (the views point into both files' text)
        c->blocks.insert(c->blocks.end(), b->blocks.begin(), b->blocks.end());
        recycle(std::move(a)); recycle(std::move(b));
        return c;
    }
    if (m.bgz_par) {
        for (;;) {
            std::unique_lock<std::mutex> lk(m.mu);
            m.cv.wait(lk, [&] { return m.cur_seg < m.msegs.size() || m.dist_done || !m.error.empty(); });
            if (!m.error.empty()) die(m.error);
            if (m.cur_seg >= m.msegs.size()) return nullptr;         // (the distributor is done and every stretch has been handed out)
            Impl::MemSeg &sg = *m.msegs[m.cur_seg];
            m.cv.wait(lk, [&] { return sg.done || !m.error.empty(); });
            if (!m.error.empty()) die(m.error);
            if (!sg.clean) {
                // from the start of this stretch on, ONE reader: the blocks the stretches from here on were given (as they were
                // given: parsing moved their bounds), what the distributor was filling, then the feeder's remaining blocks
                lk.unlock();
                m.join_parsers();
                std::deque<std::shared_ptr<TextBlock>> rest;
                auto take = [&](Impl::MemSeg &g) {
                    for (size_t i = 0; i < g.blocks.size(); ++i) { g.blocks[i]->begin = g.span[i].first; g.blocks[i]->end = g.span[i].second; rest.push_back(g.blocks[i]); }
                    g.chunks.clear(); g.blocks.clear();
                };
                for (size_t i = m.cur_seg; i < m.msegs.size(); ++i) take(*m.msegs[i]);
                if (m.filling) take(*m.filling);
                m.fell_back = true;
                m.bgz_par = false;
                SeqReader *fd = m.feeder.get();
                const bool ended = m.feeder_ended;
                m.r1.reset(new SeqReader(std::move(rest), ended ? std::function<std::shared_ptr<TextBlock>()>() : [fd] { return fd->take_block(); }));
                return next();
            }
            if (!sg.chunks.empty()) {
                auto c = std::move(sg.chunks.front());
                sg.chunks.pop_front();
                return c;
            }
            sg.blocks.clear();                                       // (the chunks hold the text they point into)
            ++m.cur_seg;
            m.cv.notify_all();
        }
    }
    if (m.r1) {                                                  // one thread, or the rest of the file after a stretch that did not end cleanly
        auto c = m.take_spare();
        const double t0 = tnow();
        const int got = bseq_read((int)m.chunk_size, *m.r1, m.r2.get(), *c);
        m.t_parse += tnow() - t0;
        if (got <= 0) return nullptr;
        if (m.paired_par) {                                          // the first chunk of a pair of files: start a parser per file
            m.first_done = true;
            m.n_per_half = (size_t)got / 2;
            for (unsigned t = 0; t < 2; ++t) m.parsers.emplace_back([this, t] { impl_->parse_half(t); });
        }
        return c;
    }
    // A stretch is handed out once it has been parsed to its end and that end checked: a stretch that did not end between two
    // records (find_cut_points makes that all but impossible) is parsed again, with everything after it, by one sequential reader
    // from where it began -- which, by induction over the stretches before it, is where the sequential parser began a record.
    for (;;) {
        std::unique_lock<std::mutex> lk(m.mu);
        if (m.cur_seg >= m.segs.size()) return nullptr;
        Impl::Segment &sg = m.segs[m.cur_seg];
        m.cv.wait(lk, [&] { return sg.done || !m.error.empty(); });
        if (!m.error.empty()) die(m.error);
        if (!sg.clean) {
            const u64 from = sg.begin;
            lk.unlock();
            m.join_parsers();
            for (auto &s : m.segs) s.chunks.clear();
            m.segs.clear();
            m.fell_back = true;
            m.r1.reset(new SeqReader(m.fq1.c_str(), 0, from, ~0ULL));
            return next();
        }
        if (!sg.chunks.empty()) {
            auto c = std::move(sg.chunks.front());
            sg.chunks.pop_front();
            return c;
        }
        ++m.cur_seg;
        m.cv.notify_all();
    }
}

// ---------------------------------------------------------------------------------------------- pre-packed read container
bool is_pack_container(const char *path)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    char m[8];
    const bool ok = ::pread(fd, m, 8, 0) == 8 && std::memcmp(m, PACK_MAGIC, 8) == 0;
    ::close(fd);
    return ok;
}

namespace {
inline size_t pad8(size_t n) { return (n + 7u) & ~size_t(7); }
void pread_all(int fd, void *dst, size_t n, u64 at, const char *what)
{
    char *d = static_cast<char *>(dst);
    for (size_t got = 0; got < n;) {
        const ssize_t r = ::pread(fd, d + got, n - got, (off_t)(at + got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) die(std::string("read container: short read in ") + what);
        got += (size_t)r;
    }
}
// byte offsets of a chunk's sections behind its header
struct PackLayout { size_t lens, words, bad_word, bad_mask, names, end; };
PackLayout pack_layout(const PackChunkHeader &h)
{
    PackLayout L;
    L.lens = 0;
    L.words = pad8((size_t)h.n_reads * 4);
    L.bad_word = L.words + (size_t)h.n_words * 8;
    L.bad_mask = L.bad_word + (size_t)h.n_bad * 8;
    L.names = pad8(L.bad_mask + (size_t)h.n_bad * 4);
    L.end = pad8(L.names + (size_t)h.names_bytes);
    return L;
}
}  // namespace

std::pair<u64, u64> pack_dataset(const char *fq1, const char *fq2, const char *out_path, unsigned chunk_bases, unsigned parser_threads, int threads,
                                 bool with_names)
{
    if (!chunk_bases) chunk_bases = 1u << 27;
    const int fd = ::open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) die(std::string("Could not open ") + out_path + " for writing.");
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    auto write_all = [&](const void *p, size_t n) {
        const char *c = static_cast<const char *>(p);
        for (size_t off = 0; off < n;) {
            const ssize_t w = ::write(fd, c + off, n - off);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) die("write failed");
            off += (size_t)w;
        }
    };
    PackFileHeader fh{};
    std::memcpy(fh.magic, PACK_MAGIC, 8);
    fh.version = 1; fh.flags = (fq2 ? 1u : 0u) | (with_names ? 2u : 0u);
    write_all(&fh, sizeof(fh));
    // a writer thread takes finished chunk images (at most two waiting) while the next chunk is gathered and packed
    std::mutex wmu;
    std::condition_variable wcv;
    std::deque<std::vector<char>> wq;
    std::vector<std::vector<char>> wfree;
    bool w_done = false;
    std::string w_err;
    std::thread writer([&] {
        for (;;) {
            std::vector<char> buf;
            {
                std::unique_lock<std::mutex> lk(wmu);
                wcv.wait(lk, [&] { return !wq.empty() || w_done; });
                if (wq.empty()) return;
                buf = std::move(wq.front()); wq.pop_front();
            }
            try { write_all(buf.data(), buf.size()); } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(wmu); if (w_err.empty()) w_err = e.what(); }
            std::lock_guard<std::mutex> lk(wmu);
            wfree.push_back(std::move(buf));
            wcv.notify_all();
        }
    });
    struct Joiner { std::thread &t; std::mutex &m; std::condition_variable &cv; bool &done; ~Joiner() { { std::lock_guard<std::mutex> lk(m); done = true; } cv.notify_all(); if (t.joinable()) t.join(); } } joiner{writer, wmu, wcv, w_done};
    ChunkSource source(fq1, fq2, chunk_bases, parser_threads, 0);
    std::vector<const char *> ptrs;
    std::vector<u32> lens, bad_mask;
    std::vector<u64> words, offsets, bad_word;
    std::string names;
    u64 n_total = 0, bases_total = 0;
    for (;;) {
        auto seqs = source.next();
        if (!seqs) break;
        size_t n = seqs->recs.size();
        if (fq2) n -= n & 1u;
        if (!n) { source.recycle(std::move(seqs)); continue; }
        ptrs.resize(n); lens.resize(n); offsets.resize(n + 1);
        u64 total = 0;
        names.clear();
        for (size_t i = 0; i < n; ++i) {
            const bseq1_t &b = seqs->recs[i];
            ptrs[i] = b.seq.data(); lens[i] = (u32)b.seq.size(); total += b.seq.size();
            if (with_names) { names.append(b.name.data(), b.name.size()); names.push_back('\0'); }
        }
        const u64 n_words = bns_packed_words(total, n);
        words.resize((size_t)n_words + 1);
        u64 n_bad = 0;
        if (bad_word.size() < 4096) { bad_word.resize(4096); bad_mask.resize(4096); }
        int rc = bns_pack_reads_ptrs(ptrs.data(), lens.data(), n, offsets.data(), words.data(), bad_word.data(), bad_mask.data(), bad_word.size(), &n_bad,
                                     std::max(1, threads));
        if (rc != BNS_OK && n_bad > bad_word.size()) {
            bad_word.resize((size_t)n_bad); bad_mask.resize((size_t)n_bad);
            rc = bns_pack_reads_ptrs(ptrs.data(), lens.data(), n, offsets.data(), words.data(), bad_word.data(), bad_mask.data(), bad_word.size(), &n_bad,
                                     std::max(1, threads));
        }
        if (rc != BNS_OK) die("bns_pack_reads_ptrs failed");
        PackChunkHeader h{};
        h.magic = PACK_CHUNK_MAGIC; h.n_reads = (u32)n; h.total_bases = total; h.n_words = n_words; h.n_bad = n_bad; h.names_bytes = names.size();
        const PackLayout L = pack_layout(h);
        h.payload_bytes = L.end;
        std::vector<char> img;
        {
            std::unique_lock<std::mutex> lk(wmu);
            wcv.wait(lk, [&] { return wq.size() < 2 || !w_err.empty(); });
            if (!w_err.empty()) die(w_err);
            if (!wfree.empty()) { img = std::move(wfree.back()); wfree.pop_back(); }
        }
        img.assign(sizeof(h) + L.end, 0);
        char *o = img.data();
        std::memcpy(o, &h, sizeof(h)); o += sizeof(h);
        std::memcpy(o + L.lens, lens.data(), n * 4);
        std::memcpy(o + L.words, words.data(), (size_t)n_words * 8);
        std::memcpy(o + L.bad_word, bad_word.data(), (size_t)n_bad * 8);
        std::memcpy(o + L.bad_mask, bad_mask.data(), (size_t)n_bad * 4);
        std::memcpy(o + L.names, names.data(), names.size());
        {
            std::lock_guard<std::mutex> lk(wmu);
            wq.push_back(std::move(img));
            wcv.notify_all();
        }
        n_total += n; bases_total += total;
        source.recycle(std::move(seqs));
    }
    {
        std::unique_lock<std::mutex> lk(wmu);
        w_done = true;
        wcv.notify_all();
    }
    writer.join();
    if (!w_err.empty()) die(w_err);
    return {n_total, bases_total};
}

namespace {
// One chunk of a container -> the page-locked buffers of the GPU call (what pack_chunk leaves) and the chunk's records: names
// as views into a block held by `seqs`, sequences as views of the right LENGTH over a filler (the Kraken formatter prints
// lengths, never bases).
void load_packed_chunk(ClassifierGeneric &c, bns_ctx *ctx, int fd, u64 off, const PackChunkHeader &h, bool paired, bool has_names, ReadChunk &seqs, ChunkResult &r)
{
    const unsigned n = h.n_reads, inc = paired ? 2u : 1u, n_units = n / inc;
    const PackLayout L = pack_layout(h);
    r.n = n; r.is_paired = paired ? 1 : 0;
    r.want_runs = c.get_emit_kraken() != 0;
    r.taxon_only = !c.get_emit_kraken() && !c.get_emit_fastq();
    r.taxon.resize(ctx, n_units);
    if (!r.taxon_only) { r.missing.resize(ctx, n_units); r.ambig.resize(ctx, n_units); r.n_hits.resize(ctx, n_units); }
    r.run_tax.clear(); r.run_len.clear();
    if (r.want_runs) { r.run_start.resize(ctx, n_units); r.n_runs.resize(ctx, n_units); }
    r.n_bad = h.n_bad; r.t_pack = r.t_call = r.t_copy = 0;
    seqs.clear();
    if (!n) return;
    // (a damaged or crafted container must not size buffers or index device memory: the sections lie inside the payload, the
    // invalid-base list below is checked entry by entry before it reaches the scatter kernel)
    if (L.end > h.payload_bytes || h.n_bad > h.n_words) die("read container: damaged chunk header");
    const double t0 = tnow();
    const u64 base = off + sizeof(PackChunkHeader);
    r.seq_lens.resize(n);
    pread_all(fd, r.seq_lens.data(), (size_t)n * 4, base + L.lens, "lengths");
    r.offsets.resize(ctx, (size_t)n + 1);
    u64 *o = r.offsets.data();
    u64 acc = 0; u32 max_len = 0;
    for (unsigned i = 0; i < n; ++i) { o[i] = acc; acc += r.seq_lens[i]; max_len = std::max(max_len, r.seq_lens[i]); }
    o[n] = acc;
    if (acc != h.total_bases || h.n_words != bns_packed_words(acc, n)) die("read container: chunk header and lengths disagree");
    u64 *words = reinterpret_cast<u64 *>(r.words.reserve(ctx, (size_t)h.n_words * 8 + 8));
    pread_all(fd, words, (size_t)h.n_words * 8, base + L.words, "words");
    if (r.bad_word.size() < std::max<size_t>(1, (size_t)h.n_bad)) { r.bad_word.resize(std::max<size_t>(4096, (size_t)h.n_bad)); r.bad_mask.resize(r.bad_word.size()); }
    if (h.n_bad) {
        pread_all(fd, r.bad_word.data(), (size_t)h.n_bad * 8, base + L.bad_word, "invalid-base list");
        pread_all(fd, r.bad_mask.data(), (size_t)h.n_bad * 4, base + L.bad_mask, "invalid-base list");
        for (u64 i = 0; i < h.n_bad; ++i)
            if (r.bad_word[i] >= h.n_words) die("read container: damaged invalid-base list");
    }
    if (!c.get_emit_kraken() && !c.get_emit_fastq()) { r.t_pack = tnow() - t0; return; }   // (-K: no per-read text, so no names and no records)
    seqs.arena.emplace_back((size_t)max_len + 1, 'N');
    const char *filler = seqs.arena.back().data();
    const char *np = nullptr, *ne = nullptr;
    if (has_names && h.names_bytes) {
        seqs.arena.emplace_back((size_t)h.names_bytes, '\0');
        pread_all(fd, seqs.arena.back().data(), (size_t)h.names_bytes, base + L.names, "names");
        np = seqs.arena.back().data(); ne = np + h.names_bytes;
    }
    seqs.recs.reserve(n);
    for (unsigned i = 0; i < n; ++i) {
        bseq1_t b;
        if (np && np < ne) {
            const char *z = static_cast<const char *>(std::memchr(np, 0, (size_t)(ne - np)));
            if (!z) die("read container: names section is short");
            b.name = std::string_view(np, (size_t)(z - np));
            np = z + 1;
        }
        b.seq = std::string_view(filler, r.seq_lens[i]);
        seqs.recs.push_back_stream(b);
    }
    RecVec::publish();
    r.t_pack = tnow() - t0;
}
}  // namespace

// ---------------------------------------------------------------------------------------------- text on the device
namespace {
// One plain FASTA / FASTQ file classified WITHOUT a host parser or packer (bns_classify_text: record boundaries, names and the
// 2-bit words are made by kernels from the file's bytes).  What the host still does: read(2) into page-locked blocks, one
// library call per block, Kraken lines from the names and results that come back.
//
// The file is cut into blocks of B bytes at NOMINAL offsets b * B.  Block b is the records that START in [start_b, (b + 1) * B):
// start_b = where the first record at or behind b * B starts = where block b - 1 stopped (bns_classify_text's `limit`: the record
// that straddles a nominal end belongs to the block it starts in, so every block's buffer holds SLACK bytes beyond its end).
// With one device start_b is simply the previous call's answer.  With G devices the blocks are in flight side by side, so a
// caller that does not yet know where its block's first record starts GUESSES it from the text (find_record_start's strict test)
// -- and the guess is checked when the block in front is done: a block whose guess was wrong is classified again from the right
// place before anything of it is printed.  Output is in file order.  Anything the kernels do not take (status IRREGULAR /
// NO_RECORD: CRLF text, wrapped quality, a record longer than SLACK, ...) ends this path at a record boundary; the caller parses
// the rest of the file on the host (process_dataset below), so the records and their order are always those of kseq_read.
struct TextJob {
    u64 seq = 0, file_off = 0, start = 0, end = 0;
    size_t bytes = 0;                                          // text bytes in the buffer, from file_off
    bool last = false, guessed = false, ok = false, prefetched = false;
    int status = 0;
    u32 why = 0;
    unsigned pieces_left = 0;
    PinnedBuf text;
    u64 n_records = 0;
    unsigned mates = 1;                                        // records per unit (2: a pair of files, mates interleaved)
    PinArr<u32> taxon, missing, ambig, n_hits, n_runs, seq_len, name_off;
    PinArr<u64> run_start;
    PinArr<char> names;
    PinArr<u32> run_tax, run_len;                              // (page-locked: the library copies the hit runs straight into them)
};

unsigned format_text_job(ClassifierGeneric &c, const TextJob &j, std::vector<ClassifierGeneric::Work::Part> &parts)
{
    const unsigned inc = j.mates, n = (unsigned)(j.n_records / inc);
    if (!n) return 0;
    const unsigned nt = (unsigned)std::max(1, std::min<int>(c.nt_, (int)(n / 4096 + 1)));
    if (parts.size() < nt) parts.resize(nt);
    std::vector<u64> ncls(nt * 2, 0);
    const bool lines = c.get_emit_kraken() != 0;
    static const char filler = 'N';
    parallel_units(nt, n, [&](unsigned lo, unsigned hi, unsigned t) {
        ClassifierGeneric::Work::Part &part = parts[t];
        part.n = 0; part.s.clear();
        u64 n_cls[2] = {0, 0};
        if (lines) part.ensure((size_t)(hi - lo) * 48 + 4096);
        for (unsigned u = lo; u < hi; ++u) {
            ++n_cls[j.taxon[u] == 0];
            if (!lines || !(c.get_emit_all() || j.taxon[u])) continue;
            bseq1_t b;
            const size_t r = (size_t)u * inc;                    // (the line prints the first mate's name and length, classifier.h:112-129)
            b.name = std::string_view(j.names.data() + j.name_off[r], j.name_off[r + 1] - j.name_off[r]);
            b.seq = std::string_view(&filler, j.seq_len[r]);     // (only its length is printed)
            const HitRuns runs{j.run_tax.data() + j.run_start[u], j.run_len.data() + j.run_start[u], j.n_runs[u]};
            const size_t bound = kraken_line_bound(runs, b);
            if (part.n + bound > part.cap) part.ensure(std::max(part.n + bound, part.cap * 2));
            part.n = (size_t)(kraken_line_raw(part.p + part.n, runs, j.taxon[u], j.ambig[u], j.missing[u], b) - part.p);
        }
        ncls[t * 2] = n_cls[0]; ncls[t * 2 + 1] = n_cls[1];
    });
    static std::mutex tally_mu;
    std::lock_guard<std::mutex> lk(tally_mu);
    for (unsigned t = 0; t < nt; ++t) { c.classified_[0] += ncls[t * 2]; c.classified_[1] += ncls[t * 2 + 1]; }
    return nt;
}

bool text_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    if (fq2 || c.get_emit_fastq()) return false;               // (FASTQ-style output prints bases and qualities: the host parser has them)
    if (const char *e = std::getenv("BNS_TEXT_GPU")) if (e[0] == '0') return false;
    struct stat st;
    if (::stat(fq1, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) return false;
    unsigned char m[2] = {0, 0};
    const int f = ::open(fq1, O_RDONLY);
    if (f < 0) return false;
    const bool plain = ::pread(f, m, 2, 0) == 2 && !(m[0] == 0x1f && m[1] == 0x8b) && (m[0] == '>' || m[0] == '@' || m[0] == '\n');
    ::close(f);
    return plain;
}

// -> the file offset the host parser has to go on from (== the file's size: nothing left)
u64 process_text_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out)
{
    const int fd = ::open(fq1, O_RDONLY);
    if (fd < 0) die(std::string("Could not open ") + fq1 + " for reading.");
    struct FdCloser { int fd; ~FdCloser() { ::close(fd); } } closer{fd};
    const u64 fsize = (u64)::lseek(fd, 0, SEEK_END);
    const int ofd = fileno(out);
    std::fflush(out);
    char first_byte = 0;
    (void)!::pread(fd, &first_byte, 1, 0);
    const bool fastq = first_byte == '@';
    const unsigned G = (unsigned)c.ctxs_.size();
    auto env_mb = [](const char *name, u64 dflt) { const char *e = std::getenv(name); return e && std::atol(e) > 0 ? (u64)std::atol(e) << 20 : dflt; };
    // (64 MiB: one upload piece.  Against 128 MiB on one box, interleaved, 256 M reads: -K 1.78-1.95 s against 1.91-1.99, Kraken lines
    // 2.20-2.34 against 2.22-2.86 -- half the page-locked memory to set up at the start, the formatters fed in smaller portions;
    // tools/r05_block_ab.sh, profiles/r05_cli_blocks.txt)
    u64 B = std::min<u64>(env_mb("BNS_TEXT_BLOCK_MB", 64ull << 20), 1ull << 30);
    u64 SLACK = std::min<u64>(env_mb("BNS_TEXT_SLACK_MB", 4ull << 20), B);
    if (const char *e = std::getenv("BNS_TEXT_BLOCK_BYTES")) { B = (u64)std::max(64L, std::atol(e)); SLACK = std::min<u64>(SLACK, std::max<u64>(B / 2, 2048)); }   // (tests: many blocks on small files)
    const u64 n_blocks = std::max<u64>(1, (fsize + B - 1) / B);
    unsigned R = (unsigned)std::max(2, std::min(8, usable_cpus() / 2));
    if (const char *e = std::getenv("BNS_TEXT_READERS")) R = (unsigned)std::max(1, std::min(32, std::atoi(e)));
    const size_t PIECE = 8u << 20;
    const bool want_runs = c.get_emit_kraken() != 0;
    const bool taxon_only = !want_runs;                        // (-K: the tally and the -b file read the taxon alone)
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;

    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<TextJob>> spare;               // recycled jobs (their page-locked buffers with them)
    unsigned jobs_made = 0;
    unsigned max_jobs = 2 * G + 4;                             // blocks in flight: in a call, uploaded ahead of it (prefetch), read ahead of that, being formatted
    if (const char *e = std::getenv("BNS_TEXT_JOBS")) max_jobs = (unsigned)std::max(2, std::min(64, std::atoi(e)));
    struct Piece { TextJob *job; size_t off, len; };
    std::deque<Piece> pieces;                                  // reads to do
    std::map<u64, std::unique_ptr<TextJob>> loading, loaded, done, verified;
    u64 next_load = 0, next_verify = 0;
    u64 verified_end = 0;                                      // where the first record of block next_verify starts
    std::map<u64, u64> end_of;                                 // block -> where it stopped (as far as known)
    std::deque<std::unique_ptr<TextJob>> redo;                 // blocks whose guessed start was wrong
    bool cancel = false, stop_loading = false;
    u64 resume_at = fsize;                                     // the host parser's share starts here (fsize: nothing)
    std::string error;
    double t_read = 0, t_call = 0, t_format = 0, t_write = 0, t_alloc = 0;
    u64 n_guess = 0, n_redo = 0, n_ahead = 0;
    double t_idle = 0;                                         // callers waiting for a block to be read
    auto fail_with = [&](const std::string &w) { if (error.empty()) error = w; cancel = true; cv.notify_all(); };

    // ---- readers: a loader hands out blocks (a job each, from the pool) cut into pieces; R threads pread the pieces
    auto reader = [&] {
        try {
            for (;;) {
                Piece pc{nullptr, 0, 0};
                {
                    std::unique_lock<std::mutex> lk(mu);
                    for (;;) {
                        if (cancel) return;
                        if (!pieces.empty()) { pc = pieces.front(); pieces.pop_front(); break; }
                        // nothing to read: open the next block if a job is to be had
                        if (!stop_loading && next_load < n_blocks && (!spare.empty() || jobs_made < max_jobs)) {
                            std::unique_ptr<TextJob> j;
                            if (!spare.empty()) { j = std::move(spare.back()); spare.pop_back(); }
                            else { j = std::make_unique<TextJob>(); ++jobs_made; }
                            j->seq = next_load++;
                            j->file_off = j->seq * B;
                            j->bytes = (size_t)std::min<u64>(fsize - j->file_off, B + SLACK);
                            j->last = j->file_off + j->bytes >= fsize;
                            j->guessed = j->ok = j->prefetched = false; j->n_records = 0; j->status = 0; j->why = 0;
                            TextJob *jp = j.get();
                            const u64 seq = j->seq;
                            loading[seq] = std::move(j);
                            lk.unlock();
                            const double ta = tnow();
                            jp->text.reserve(c.ctxs_[seq % G], (size_t)(B + SLACK) + 256);      // (page-locks on first use: 0.2 ms per MiB, once per job)
                            const double tb = tnow();
                            lk.lock();
                            t_alloc += tb - ta;
                            unsigned np = 0;
                            for (size_t o = 0; o < jp->bytes; o += PIECE) { pieces.push_back(Piece{jp, o, std::min(PIECE, jp->bytes - o)}); ++np; }
                            jp->pieces_left = np;
                            if (!np) { loaded[seq] = std::move(loading[seq]); loading.erase(seq); }
                            cv.notify_all();
                            continue;
                        }
                        if (next_load >= n_blocks || stop_loading) { if (pieces.empty() && loading.empty()) return; }
                        cv.wait(lk);
                    }
                }
                const double t0 = tnow();
                pread_all(fd, pc.job->text.p + pc.off, pc.len, pc.job->file_off + pc.off, "text block");
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu);
                t_read += t1 - t0;
                if (--pc.job->pieces_left == 0) {
                    const u64 seq = pc.job->seq;
                    loaded[seq] = std::move(loading[seq]);
                    loading.erase(seq);
                }
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    };

    // ---- one library call on a block from a known (or guessed) start
    auto call_block = [&](bns_ctx *ctx, TextJob &j) {
        const u64 rel = j.start - j.file_off;
        u64 cap = (j.bytes - rel) / 160 + 4096, names_cap = cap * 24, runs_cap = cap * 4;   // (316 bytes, ~10 of name and 1-3 hit runs per 150-bp FASTQ record; BNS_TEXT_CAP doubles them)
        for (;;) {
            j.taxon.resize(ctx, cap);
            bns_text_out o{};
            o.taxon = j.taxon.data();
            if (!taxon_only) {
                j.missing.resize(ctx, cap); j.ambig.resize(ctx, cap); j.n_hits.resize(ctx, cap); j.seq_len.resize(ctx, cap); j.name_off.resize(ctx, cap + 1);
                j.run_start.resize(ctx, cap); j.n_runs.resize(ctx, cap); j.names.resize(ctx, names_cap);
                o.missing = j.missing.data(); o.ambig = j.ambig.data(); o.n_hits = j.n_hits.data(); o.seq_len = j.seq_len.data();
                o.name_off = j.name_off.data(); o.names = j.names.data(); o.names_cap = names_cap;
                o.run_start = j.run_start.data(); o.n_runs = j.n_runs.data();
                j.run_tax.resize(ctx, runs_cap); j.run_len.resize(ctx, runs_cap);
                o.run_tax = j.run_tax.data(); o.run_len = j.run_len.data(); o.runs_cap = runs_cap;
            }
            bns_text_info info{};
            const char *tp = j.text.p + rel;
            const u64 tb = j.bytes - rel;
            const u64 limit = j.last ? ~0ULL : (j.file_off + B) - j.start;
            chk(ctx, bns_classify_text(ctx, &tp, &tb, 1, limit, (j.last ? BNS_TEXT_FINAL : 0) | BNS_TEXT_TRIM_READNO, cap, &o, &info), "bns_classify_text");
            if (info.status == BNS_TEXT_CAP) { cap *= 2; names_cap *= 2; runs_cap *= 2; continue; }      // (short records, long names or many runs: once more with room)
            j.n_records = info.n_records; j.status = info.status; j.why = info.why;
            j.end = j.start + info.consumed[0];
            j.ok = info.status == BNS_TEXT_OK && (j.last ? j.end == j.file_off + j.bytes : j.end >= j.file_off + B);
            return;
        }
    };
    // blocks leave `done` in file order: a block whose first record is where the block in front stopped is verified (and stays
    // classified); one whose guess was wrong goes back to a caller.  (called with mu held)
    auto sequence = [&] {
        for (;;) {
            if (resume_at != fsize) return;                    // (handed over: what other devices still finish is dropped)
            auto it = done.find(next_verify);
            if (it == done.end()) return;
            TextJob &j = *it->second;
            if (j.start != verified_end) {                     // guessed wrong (or behind a block that was): classify again from the right place
                j.start = verified_end; j.guessed = false;
                ++n_redo;
                redo.push_back(std::move(it->second));
                done.erase(it);
                cv.notify_all();
                return;
            }
            // the kernels do not take (all of) this text: what they took is printed, the host parser goes on from where they stopped
            if (!j.ok) { resume_at = j.end; stop_loading = true; }
            verified_end = j.end;
            end_of[next_verify] = j.end;                       // (a fact now, whatever the block's caller guessed)
            verified[next_verify] = std::move(it->second);
            done.erase(it);
            ++next_verify;
            cv.notify_all();
        }
    };
    // blocks go to the devices in turn (block b to device b % G), so that a caller knows which block is its next one and can start
    // that block's upload (bns_text_prefetch) before it classifies the current one: the link stays busy across calls
    auto caller = [&](unsigned g) {
        try {
            u64 mine = g;
            for (;;) {
                std::unique_ptr<TextJob> j;
                TextJob *ahead = nullptr;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    for (;;) {
                        const double tw0 = tnow();
                        cv.wait(lk, [&] { return cancel || resume_at != fsize || !redo.empty() || loaded.count(mine) || (mine >= n_blocks && next_verify >= n_blocks); });
                        if (mine < n_blocks) t_idle += tnow() - tw0;
                        if (cancel || resume_at != fsize) return;
                        if (!redo.empty()) { j = std::move(redo.front()); redo.pop_front(); break; }
                        if (!loaded.count(mine)) return;       // (every block is verified)
                        const u64 b = mine;
                        TextJob &nj = *loaded[b];
                        if (b == 0) nj.start = 0;
                        else if (end_of.count(b - 1)) nj.start = end_of[b - 1];
                        else {                                 // the block in front is still on another device: guess from the text
                            const long at = find_record_start(nj.text.p, std::min<size_t>(nj.bytes, (size_t)SLACK), fastq);
                            if (at < 0) { cv.wait(lk, [&] { return cancel || resume_at != fsize || end_of.count(b - 1) || !redo.empty(); }); continue; }
                            nj.start = nj.file_off + (u64)at; nj.guessed = true; ++n_guess;
                        }
                        j = std::move(loaded[b]); loaded.erase(b);
                        mine += G;
                        break;
                    }
                    auto it = loaded.find(mine);
                    if (it != loaded.end() && !it->second->prefetched) { ahead = it->second.get(); ahead->prefetched = true; ++n_ahead; }
                }
                const double t0 = tnow();
                if (j->start > j->file_off + j->bytes) die("text block: its first record starts behind its buffer");
                if (ahead) {                                   // (only this caller takes that block: it stays where it is until then)
                    const char *tp = ahead->text.p; const u64 tb = ahead->bytes;
                    chk(c.ctxs_[g], bns_text_prefetch(c.ctxs_[g], &tp, &tb, 1), "bns_text_prefetch");
                }
                call_block(c.ctxs_[g], *j);
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu);
                t_call += t1 - t0;
                if (!j->guessed) end_of[j->seq] = j->end;      // (a guessed block's end is only as good as its guess)
                const u64 seq = j->seq;
                done[seq] = std::move(j);
                sequence();
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    };

    // ---- formatters (alternate blocks) and the writer (file order)
    constexpr unsigned NF = 2, NSETS = 2 * NF;
    std::vector<ClassifierGeneric::Work::Part> out_sets[NSETS];
    std::vector<u32> w_taxa[NSETS];
    bool w_pending[NSETS] = {};
    unsigned w_parts[NSETS] = {};
    u64 w_next = 0, n_final = ~0ULL;                           // n_final: blocks this path prints (known when loading ends or the path hands over)
    auto write_all = [&](const char *p, size_t n) {
        for (size_t off = 0; off < n;) { const ssize_t w = ::write(ofd, p + off, n - off); if (w <= 0) die("write failed"); off += (size_t)w; }
    };
    auto writer = [&] {
        try {
            for (;;) {
                unsigned set;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return cancel || w_pending[w_next % NSETS] || w_next >= n_final; });
                    if (cancel || (!w_pending[w_next % NSETS] && w_next >= n_final)) return;
                    set = (unsigned)(w_next % NSETS);
                }
                const double t0 = tnow();
                for (unsigned t = 0; t < w_parts[set]; ++t) write_all(out_sets[set][t].p, out_sets[set][t].n);
                if (c.taxon_out_ && !w_taxa[set].empty())
                    if (std::fwrite(w_taxa[set].data(), 4, w_taxa[set].size(), c.taxon_out_) != w_taxa[set].size()) die("write failed (taxon file)");
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu);
                t_write += t1 - t0;
                w_pending[set] = false; ++w_next;
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    };
    auto formatter = [&](unsigned f) {
        try {
            for (u64 next = f;; next += NF) {
                std::unique_ptr<TextJob> j;
                const unsigned set = (unsigned)(next % NSETS);
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return cancel || (verified.count(next) && !w_pending[set]) || (next >= n_final && !verified.count(next)); });
                    if (cancel || !verified.count(next)) return;
                    j = std::move(verified[next]); verified.erase(next);
                }
                if (j->seq == 0) std::fprintf(stderr, "nseq: %i\n", (int)j->n_records);
                const double t0 = tnow();
                const unsigned np = format_text_job(c, *j, out_sets[set]);
                w_taxa[set].clear();
                if (c.taxon_out_ && j->n_records) w_taxa[set].assign(j->taxon.data(), j->taxon.data() + j->n_records);
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu);
                t_format += t1 - t0;
                w_pending[set] = true; w_parts[set] = np;
                spare.push_back(std::move(j));
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    };

    std::vector<std::thread> readers, callers, formatters;
    for (unsigned r = 0; r < R; ++r) readers.emplace_back(reader);
    for (unsigned g = 0; g < G; ++g) callers.emplace_back(caller, g);
    for (unsigned f = 0; f < NF; ++f) formatters.emplace_back(formatter, f);
    std::thread wr(writer);
    for (auto &t : callers) t.join();
    {
        std::lock_guard<std::mutex> lk(mu);
        n_final = next_verify;                                 // every block up to here is verified (or the path has handed over there)
        stop_loading = true;
        cv.notify_all();
    }
    for (auto &t : formatters) t.join();
    wr.join();
    { std::lock_guard<std::mutex> lk(mu); cancel = true; cv.notify_all(); }
    for (auto &t : readers) t.join();
    for (bns_ctx *cx : c.ctxs_) (void)bns_text_prefetch(cx, nullptr, nullptr, 0);      // (blocks uploaded ahead of a call that never came: handed over, or failed)
    if (!error.empty()) die(error);
    if (timing)
        std::fprintf(stderr, "[timing] text on the device: %llu blocks of %llu MiB on %u device(s), %u readers: page-lock %.3f s, pread %.3f (summed), calls %.3f (summed), format %.3f, write %.3f; "
                             "callers waited %.3f s for blocks, %llu uploads started ahead of their call; %llu guessed starts, %llu classified again%s\n",
                     (unsigned long long)next_verify, (unsigned long long)(B >> 20), G, R, t_alloc, t_read, t_call, t_format, t_write, t_idle, (unsigned long long)n_ahead,
                     (unsigned long long)n_guess, (unsigned long long)n_redo, resume_at != fsize ? "; the host parser takes the rest" : "");
    return resume_at;
}
// ---- finished blocks -> text, in block order (formatter threads taking alternate blocks, one writer): what process_text_gpu does inline,
// as an object of its own for the BGZF path below
class TextSink {
public:
    TextSink(ClassifierGeneric &c, int ofd, std::function<void(std::unique_ptr<TextJob>)> recycle) : c_(c), ofd_(ofd), recycle_(std::move(recycle))
    {
        for (unsigned f = 0; f < NF; ++f) formatters_.emplace_back([this, f] { format_loop(f); });
        writer_ = std::thread([this] { write_loop(); });
    }
    ~TextSink() { try { finish(0, true); } catch (...) {} }
    void submit(std::unique_ptr<TextJob> j)
    {
        std::lock_guard<std::mutex> lk(mu_);
        const u64 seq = j->seq;
        ready_[seq] = std::move(j);
        cv_.notify_all();
    }
    // every block below n_final has been (or will be) submitted: returns when they are written.  abandon: stop at once.
    void finish(u64 n_final, bool abandon = false)
    {
        if (joined_) return;
        { std::lock_guard<std::mutex> lk(mu_); n_final_ = n_final; if (abandon) cancel_ = true; cv_.notify_all(); }
        for (auto &t : formatters_) t.join();
        writer_.join();
        joined_ = true;
        if (!abandon && !error_.empty()) die(error_);
    }
    bool failed() { std::lock_guard<std::mutex> lk(mu_); return !error_.empty(); }
    double t_format = 0, t_write = 0;
private:
    static constexpr unsigned NF = 2, NSETS = 2 * NF;
    void fail(const std::string &w) { std::lock_guard<std::mutex> lk(mu_); if (error_.empty()) error_ = w; cancel_ = true; cv_.notify_all(); }
    void format_loop(unsigned f)
    {
        try {
            for (u64 next = f;; next += NF) {
                std::unique_ptr<TextJob> j;
                const unsigned set = (unsigned)(next % NSETS);
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return cancel_ || (ready_.count(next) && !w_pending_[set]) || (next >= n_final_ && !ready_.count(next)); });
                    if (cancel_ || !ready_.count(next)) return;
                    j = std::move(ready_[next]); ready_.erase(next);
                }
                if (j->seq == 0) std::fprintf(stderr, "nseq: %i\n", (int)j->n_records);
                const double t0 = tnow();
                const unsigned np = format_text_job(c_, *j, out_sets_[set]);
                w_taxa_[set].clear();
                if (c_.taxon_out_ && j->n_records) w_taxa_[set].assign(j->taxon.data(), j->taxon.data() + j->n_records / j->mates);
                const double t1 = tnow();
                recycle_(std::move(j));
                std::lock_guard<std::mutex> lk(mu_);
                t_format += t1 - t0;
                w_pending_[set] = true; w_parts_[set] = np;
                cv_.notify_all();
            }
        } catch (const std::exception &e) { fail(e.what()); }
    }
    void write_loop()
    {
        try {
            for (;;) {
                unsigned set;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return cancel_ || w_pending_[w_next_ % NSETS] || w_next_ >= n_final_; });
                    if (cancel_ || (!w_pending_[w_next_ % NSETS] && w_next_ >= n_final_)) return;
                    set = (unsigned)(w_next_ % NSETS);
                }
                const double t0 = tnow();
                for (unsigned t = 0; t < w_parts_[set]; ++t) {
                    const char *p = out_sets_[set][t].p;
                    for (size_t off = 0, n = out_sets_[set][t].n; off < n;) { const ssize_t w = ::write(ofd_, p + off, n - off); if (w <= 0) die("write failed"); off += (size_t)w; }
                }
                if (c_.taxon_out_ && !w_taxa_[set].empty())
                    if (std::fwrite(w_taxa_[set].data(), 4, w_taxa_[set].size(), c_.taxon_out_) != w_taxa_[set].size()) die("write failed (taxon file)");
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu_);
                t_write += t1 - t0;
                w_pending_[set] = false; ++w_next_;
                cv_.notify_all();
            }
        } catch (const std::exception &e) { fail(e.what()); }
    }
    ClassifierGeneric &c_;
    int ofd_;
    std::function<void(std::unique_ptr<TextJob>)> recycle_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::map<u64, std::unique_ptr<TextJob>> ready_;
    std::vector<ClassifierGeneric::Work::Part> out_sets_[NSETS];
    std::vector<u32> w_taxa_[NSETS];
    bool w_pending_[NSETS] = {};
    unsigned w_parts_[NSETS] = {};
    u64 w_next_ = 0, n_final_ = ~0ULL;
    bool cancel_ = false, joined_ = false;
    std::string error_;
    std::vector<std::thread> formatters_;
    std::thread writer_;
};

bool bgzf_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    if (fq2 || c.get_emit_fastq()) return false;
    if (const char *e = std::getenv("BNS_TEXT_GPU")) if (e[0] == '0') return false;
    struct stat st;
    if (::stat(fq1, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    return is_bgzf_file(fq1) && !std::getenv("BNS_NO_BGZF");
}

// A BGZF file as text in DEVICE memory, batch by batch in file order: compressed members up (pread into page-locked memory,
// bns_inflate_members_device: one member per wavefront, thousands per batch, two batches side by side on inflater handles of their
// own), their text left in HBM behind HEAD bytes of room (for what the caller could not finish of the batch in front: the record that
// straddles two batches).  The producer half of process_bgzf_gpu / process_bgzf_gpu_pair; one device.
class BgzfDeviceSource {
public:
    struct Item { u64 seq = 0; int tbuf = -1; u64 text_bytes = 0; bool last = false; };
    u64 HEAD = 0, TEXT_MAX = 0;
    unsigned R = 0, NI = 0;
    // (what the timing line prints)
    double t_read = 0, t_inflate = 0, t_kernel = 0, t_split = 0, t_pin = 0, t_wait_inf = 0, t_wait_next = 0, t_wait_walk = 0, t_first_inflated = 0;
    u64 n_members = 0, text_total = 0;

    BgzfDeviceSource(ClassifierGeneric &c, const char *path) : ctx_(c.ctxs_[0])
    {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) die(std::string("Could not open ") + path + " for reading.");
        fsize_ = (u64)::lseek(fd_, 0, SEEK_END);
        auto env_num = [](const char *name, u64 dflt) { const char *e = std::getenv(name); return e && std::atol(e) > 0 ? (u64)std::atol(e) : dflt; };
        MEMB_ = std::min<u64>(env_num("BNS_BGZF_BATCH_MEMBERS", 16384), 1u << 20);   // members per batch
        HEAD = std::min<u64>(env_num("BNS_BGZF_HEAD_MB", 64) << 20, 256ull << 20);   // room in front of a batch's text for what the batch before left
        if (const char *e = std::getenv("BNS_BGZF_HEAD_BYTES")) HEAD = (u64)std::max(4096L, std::min(256L << 20, std::atol(e)));       // (tests: windows of a few records)
        TEXT_MAX = std::min<u64>(MEMB_ * 65536ull, (2047ull << 20) - HEAD);           // (a call takes less than 2^31 bytes of text, what the batch in front left included)
        NI = (unsigned)std::max<u64>(1, std::min<u64>(8, env_num("BNS_BGZF_GPU_THREADS", 2)));
        R = (unsigned)std::max(2, std::min(6, usable_cpus() / 3));
        // The file is read in RANGES of CB compressed bytes at nominal offsets (plus one member's worth of slack), side by side and ahead;
        // a walker goes over the ranges in file order and finds the members in the bytes that were just read -- no page of a mapping
        // is touched (walking the headers over a mapping was a page fault per member: 1.8-2.5 s per 460 k members, the longest stage).
        // A batch = the members that START in a range (the one that straddles its end included: hence the slack).
        // CB: 96 MiB = ~3 k members a batch.  The member-per-wavefront inflate kernel is at its rate from ~4 k members in flight (two
        // handles work side by side), and a slot is page-locked before its first use, 0.45 ms per MiB with the other threads' HIP calls
        // waiting behind it: with 384 MiB ranges (what the member-per-lane kernel wanted) the GPU stood idle for the first 0.3 s of a
        // file (profiles/r05_bgzf_trace.txt: 64 M reads 1.35 s with 384 MiB, 0.92 with 128, 0.88 with 96 and with 64, 1.04 with 48).
        const u64 CB = std::max<u64>(1u << 20, env_num("BNS_BGZF_RANGE_MB", 96) << 20);
        // (the FIRST range is short: a slot is page-locked before it is read -- 0.45 ms per MiB -- and nothing is inflated until the first one
        // is; one short range only: every size step re-allocates the inflaters' device buffers and the result arrays, a drained device each)
        range_off_.push_back(0);
        for (u64 ramp : {CB / 12}) if (ramp >= (1u << 20) && range_off_.back() + ramp < fsize_) range_off_.push_back(range_off_.back() + ramp);
        while (range_off_.back() + CB < fsize_) range_off_.push_back(range_off_.back() + CB);
        range_off_.push_back(std::max<u64>(fsize_, range_off_.back()));
        n_ranges_ = range_off_.size() - 1;
        NS_ = NI + 3;
        // device text buffers, HEAD + TEXT_MAX each: one per inflater, one with the caller, one inflated and waiting
        tbufs_.assign(NI + 2, nullptr);
        try {
            for (auto &p : tbufs_) chk(ctx_, bns_dev_alloc(ctx_, (size_t)(HEAD + TEXT_MAX) + 4096, &p), "bns_dev_alloc");
            for (unsigned i = 0; i < tbufs_.size(); ++i) free_t_.push_back((int)i);
            // (the handles are made HERE, before a reader page-locks its first slot: a stream created behind five hipHostMallocs waited 0.3 s)
            handles_.assign(NI, nullptr);
            for (auto &h : handles_) if (bns_inflater_create(c.devices_[0], &h) != BNS_OK) die("BGZF input: could not open an inflater on the GPU");
        } catch (...) { free_all(); throw; }
        t_begin_ = tnow();
        splitter_ = std::thread([this] { split_loop(); });
        for (unsigned r = 0; r < R; ++r) readers_.emplace_back([this] { read_loop(); });
        for (unsigned i = 0; i < NI; ++i) inflaters_.emplace_back([this, i] { inflate_loop(handles_[i]); });
    }
    // everybody home (the figures above are final after this)
    void stop()
    {
        cancel();
        if (splitter_.joinable()) splitter_.join();
        for (auto &t : readers_) if (t.joinable()) t.join();
        for (auto &t : inflaters_) if (t.joinable()) t.join();
    }
    ~BgzfDeviceSource()
    {
        stop();
        loaded_.clear(); inflated_.clear(); reading_.clear(); read_done_.clear();     // (their slots go back to spare_ while it still exists)
        for (Slot *p : all_slots_) delete p;
        free_all();
    }
    BgzfDeviceSource(const BgzfDeviceSource &) = delete;
    BgzfDeviceSource &operator=(const BgzfDeviceSource &) = delete;

    // the next batch in file order; false: there is none (the file is done, cancel() was called, or a thread failed: error())
    bool next(Item &it)
    {
        std::unique_lock<std::mutex> lk(mu_);
        const double tw = tnow();
        cv_.wait(lk, [&] { return cancel_ || inflated_.count(next_out_) || next_out_ >= n_batches_; });
        t_wait_next += tnow() - tw;
        if (next_out_ == 0) t_first_inflated = tnow() - t_begin_;
        if (cancel_ || !inflated_.count(next_out_)) return false;
        std::unique_ptr<Batch> b = std::move(inflated_[next_out_]); inflated_.erase(next_out_);
        it.seq = next_out_++; it.tbuf = b->tbuf; it.text_bytes = b->text_bytes; it.last = b->last;
        return true;
    }
    char *buf(int t) const { return static_cast<char *>(tbufs_[(size_t)t]); }
    void release(int t) { std::lock_guard<std::mutex> lk(mu_); free_t_.push_back(t); cv_.notify_all(); }
    void cancel() { std::lock_guard<std::mutex> lk(mu_); cancel_ = true; cv_.notify_all(); }
    std::string error() { std::lock_guard<std::mutex> lk(mu_); return error_; }

private:
    struct Slot { PinnedBuf comp; u64 seq = 0, file_off = 0; size_t bytes = 0; unsigned pieces_left = 0; };
    struct Batch {
        u64 seq = 0, text_bytes = 0;
        bool last = false;
        std::shared_ptr<Slot> slot;
        std::vector<u64> in_off, out_off;
        std::vector<u32> in_len, out_len, want_crc, crc, status;
        int tbuf = -1;                                          // device text buffer it was inflated into
    };
    struct Piece { Slot *s; size_t off, len; };
    static constexpr u64 SLACK = 65536 + 64;

    void free_all()
    {
        for (bns_inflater *h : handles_) if (h) bns_inflater_destroy(h);
        handles_.clear();
        for (void *p : tbufs_) if (p) bns_dev_free(ctx_, p);
        tbufs_.clear();
        if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
    }
    void fail_with(const std::string &w) { if (error_.empty()) error_ = w; cancel_ = true; cv_.notify_all(); }     // (mu_ held)

    // ---- readers: ranges of the file into page-locked slots, piece by piece
    void read_loop()
    {
        try {
            for (;;) {
                Piece pc{nullptr, 0, 0};
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    for (;;) {
                        if (cancel_) return;
                        if (!pieces_.empty()) { pc = pieces_.front(); pieces_.pop_front(); break; }
                        if (next_range_ < n_ranges_ && (!spare_.empty() || all_slots_.size() < NS_)) {
                            Slot *sl;
                            if (!spare_.empty()) { sl = spare_.back(); spare_.pop_back(); }
                            else { sl = new Slot(); all_slots_.push_back(sl); }
                            sl->seq = next_range_++;
                            sl->file_off = range_off_[sl->seq];
                            sl->bytes = (size_t)std::min<u64>(fsize_ - sl->file_off, range_off_[sl->seq + 1] - sl->file_off + SLACK);
                            reading_[sl->seq] = std::shared_ptr<Slot>(sl, [this](Slot *q) { std::lock_guard<std::mutex> g(mu_); spare_.push_back(q); cv_.notify_all(); });
                            lk.unlock();
                            const double tp0 = tnow();
                            sl->comp.reserve(ctx_, sl->bytes + 256);
                            const double tp1 = tnow();
                            lk.lock();
                            t_pin += tp1 - tp0;
                            const size_t PIECE = 8u << 20;
                            unsigned np = 0;
                            for (size_t o = 0; o < sl->bytes; o += PIECE) { pieces_.push_back(Piece{sl, o, std::min(PIECE, sl->bytes - o)}); ++np; }
                            sl->pieces_left = np;
                            if (!np) { read_done_[sl->seq] = std::move(reading_[sl->seq]); reading_.erase(sl->seq); }
                            cv_.notify_all();
                            continue;
                        }
                        if (next_range_ >= n_ranges_ && reading_.empty()) return;
                        cv_.wait(lk);
                    }
                }
                const double t0 = tnow();
                pread_all(fd_, pc.s->comp.p + pc.off, pc.len, pc.s->file_off + pc.off, "BGZF members");
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu_);
                t_read += t1 - t0;
                if (--pc.s->pieces_left == 0) { const u64 q = pc.s->seq; read_done_[q] = std::move(reading_[q]); reading_.erase(q); }
                cv_.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu_); fail_with(e.what()); }
    }
    // ---- walker: the members of every range, in file order -> batches
    void split_loop()
    {
        try {
            u64 at = 0, seq = 0;
            for (u64 r = 0; r < n_ranges_; ++r) {
                std::shared_ptr<Slot> sl;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    const double tw = tnow();
                    cv_.wait(lk, [&] { return cancel_ || read_done_.count(r); });
                    t_wait_walk += tnow() - tw;
                    if (cancel_) return;
                    sl = std::move(read_done_[r]); read_done_.erase(r);
                }
                const double t0 = tnow();
                const u64 range_end = range_off_[r + 1];
                const unsigned char *buf = reinterpret_cast<const unsigned char *>(sl->comp.p);
                std::unique_ptr<Batch> cur;
                auto emit = [&](bool last) {
                    if (!cur) { cur = std::make_unique<Batch>(); cur->slot = sl; }
                    cur->seq = seq++; cur->last = last;
                    std::lock_guard<std::mutex> lk(mu_);
                    n_members += cur->in_off.size(); text_total += cur->text_bytes;
                    const u64 q = cur->seq;
                    loaded_[q] = std::move(cur);
                    if (last) n_batches_ = seq;
                    cv_.notify_all();
                };
                while (at < range_end) {
                    if (at < sl->file_off) die("BGZF input: member walk fell behind its range");
                    const size_t rel = (size_t)(at - sl->file_off);
                    size_t pay = 0;
                    const size_t msz = bgzf_member(buf + rel, sl->bytes - rel, pay);
                    if (!msz) die(at + 18 > fsize_ ? "truncated BGZF member" : "damaged BGZF member header (or gzip members without the BC field after BGZF ones)");
                    if (at + msz > fsize_ || rel + msz > sl->bytes) die("truncated BGZF member");
                    if (msz < pay + 8) die("damaged BGZF member");
                    const unsigned char *t = buf + rel + msz - 8;
                    const u32 crc = t[0] | ((u32)t[1] << 8) | ((u32)t[2] << 16) | ((u32)t[3] << 24);
                    const u32 isize = t[4] | ((u32)t[5] << 8) | ((u32)t[6] << 16) | ((u32)t[7] << 24);
                    if (isize > 65536u) die("damaged BGZF member (recorded text size above 64 KiB)");
                    if (isize) {
                        if (cur && (cur->in_off.size() >= MEMB_ || cur->text_bytes + isize > TEXT_MAX)) emit(false);      // (a range that inflates to more than a buffer holds: several batches)
                        if (!cur) { cur = std::make_unique<Batch>(); cur->slot = sl; }
                        cur->in_off.push_back(rel + pay); cur->in_len.push_back((u32)(msz - pay - 8));
                        cur->out_off.push_back(cur->text_bytes); cur->out_len.push_back(isize); cur->want_crc.push_back(crc);
                        cur->text_bytes += isize;
                    }
                    at += msz;
                }
                const bool file_done = at >= fsize_;
                t_split += tnow() - t0;
                if (cur || file_done) emit(file_done);
                if (file_done) break;
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu_); fail_with(e.what()); }
    }
    // ---- inflaters: a handle each; batches in file order, each into a free device text buffer (behind HEAD bytes of room)
    void inflate_loop(bns_inflater *h)
    {
        try {
            for (;;) {
                std::unique_ptr<Batch> b;
                int tb = -1;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    const double tw = tnow();
                    cv_.wait(lk, [&] { return cancel_ || (loaded_.count(next_inflate_) && !free_t_.empty()) || next_inflate_ >= n_batches_; });
                    t_wait_inf += tnow() - tw;
                    if (cancel_ || !loaded_.count(next_inflate_)) break;
                    b = std::move(loaded_[next_inflate_]); loaded_.erase(next_inflate_); ++next_inflate_;
                    tb = free_t_.back(); free_t_.pop_back();
                }
                const size_t n = b->in_off.size();
                b->crc.assign(n, 0); b->status.assign(n, 0);
                const double t0 = tnow();
                if (n) {
                    const int rc = bns_inflate_members_device(h, reinterpret_cast<const uint8_t *>(b->slot->comp.p), b->slot->bytes, b->in_off.data(), b->in_len.data(),
                                                              b->out_off.data(), b->out_len.data(), n, static_cast<char *>(tbufs_[(size_t)tb]) + HEAD, b->text_bytes,
                                                              b->crc.data(), b->status.data());
                    if (rc != BNS_OK) die(std::string("bns_inflate_members_device: ") + bns_inflater_error(h));
                    for (size_t i = 0; i < n; ++i)
                        if (b->status[i] != BNS_INF_OK || b->crc[i] != b->want_crc[i]) die("BGZF member does not inflate to its recorded size and checksum");
                }
                const double t1 = tnow();
                b->tbuf = tb;
                b->slot.reset();                                // (the compressed bytes are done with: the slot goes back to the readers)
                std::lock_guard<std::mutex> lk(mu_);
                t_inflate += t1 - t0;
                t_kernel += std::max(0.f, bns_inflater_last_kernel_ms(h)) * 1e-3;
                const u64 seq = b->seq;
                inflated_[seq] = std::move(b);
                cv_.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu_); fail_with(e.what()); }
    }

    bns_ctx *ctx_;
    int fd_ = -1;
    u64 fsize_ = 0, MEMB_ = 0, n_ranges_ = 0;
    unsigned NS_ = 0;
    std::vector<u64> range_off_;
    std::vector<void *> tbufs_;
    std::vector<bns_inflater *> handles_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Slot *> spare_, all_slots_;                    // (slots go back to spare_ when the last batch that points into them lets go)
    std::deque<Piece> pieces_;
    std::map<u64, std::shared_ptr<Slot>> reading_, read_done_;
    std::map<u64, std::unique_ptr<Batch>> loaded_, inflated_;
    std::vector<int> free_t_;
    u64 next_range_ = 0, next_inflate_ = 0, next_out_ = 0, n_batches_ = ~0ULL;
    bool cancel_ = false;
    std::string error_;
    double t_begin_ = 0;
    std::thread splitter_;
    std::vector<std::thread> readers_, inflaters_;
};

// the result arrays of one bns_classify_text call, sized for `cap` records (names_cap / runs_cap bytes / runs)
static void size_text_job(bns_ctx *ctx, TextJob &j, bns_text_out &o, bool taxon_only, u64 cap, u64 names_cap, u64 runs_cap)
{
    j.taxon.resize(ctx, cap);
    o = bns_text_out{};
    o.taxon = j.taxon.data();
    if (taxon_only) return;
    j.missing.resize(ctx, cap); j.ambig.resize(ctx, cap); j.n_hits.resize(ctx, cap); j.seq_len.resize(ctx, cap); j.name_off.resize(ctx, cap + 1);
    j.run_start.resize(ctx, cap); j.n_runs.resize(ctx, cap); j.names.resize(ctx, names_cap);
    o.missing = j.missing.data(); o.ambig = j.ambig.data(); o.n_hits = j.n_hits.data(); o.seq_len = j.seq_len.data();
    o.name_off = j.name_off.data(); o.names = j.names.data(); o.names_cap = names_cap;
    o.run_start = j.run_start.data(); o.n_runs = j.n_runs.data();
    j.run_tax.resize(ctx, runs_cap); j.run_len.resize(ctx, runs_cap);
    o.run_tax = j.run_tax.data(); o.run_len = j.run_len.data(); o.runs_cap = runs_cap;
}

// A BGZF file whose text never leaves the device: BgzfDeviceSource's batches, what the batch in front could not finish copied in front
// of the next one's text (device to device), bns_classify_text on it where it lies, names and results down.
// -> true: the whole file was classified.  false: the kernels handed text back (not in their regular form) after `units_done`
// units had been printed: the caller reads the file with the host parser and leaves those out.
bool process_bgzf_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    bns_ctx *ctx = c.ctxs_[0];
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) (void)bns_set_timing(ctx, 1);             // (HIP events around the parse and classify kernels: the sums in the timing line)
    const bool want_runs = c.get_emit_kraken() != 0, taxon_only = !want_runs;
    std::mutex mu;
    std::vector<std::unique_ptr<TextJob>> spare_j;
    auto recycle_job = [&](std::unique_ptr<TextJob> j) { std::lock_guard<std::mutex> lk(mu); spare_j.push_back(std::move(j)); };
    TextSink sink(c, ofd, recycle_job);
    BgzfDeviceSource src(c, fq1);
    const u64 HEAD = src.HEAD;
    double t_gpu_parse = 0, t_gpu_cls = 0, t_call = 0;

    bool handed_back = false;
    u64 n_jobs = 0;                                        // jobs handed to the sink (one per call that took records: a batch as a rule)
    std::string failure;
    try {
        int prev_t = -1;
        u64 tail_off = 0, tail_len = 0;                        // what the batch in front left: src.buf(prev_t) + tail_off, tail_len bytes
        BgzfDeviceSource::Item b;
        while (src.next(b)) {
            std::unique_ptr<TextJob> j;
            if (tail_len > HEAD) { src.release(b.tbuf); handed_back = true; break; }  // (a record longer than HEAD: the host parser's)
            char *base = src.buf(b.tbuf);
            const double t0 = tnow();
            if (tail_len) chk(ctx, bns_dev_copy(ctx, base + HEAD - tail_len, src.buf(prev_t) + tail_off, (size_t)tail_len), "bns_dev_copy");
            // (the buffer of the batch in front is free from here on -- not after this batch's classify: held that long, the classify
            // stage sat on two of the three buffers and the two inflaters took turns on the third)
            if (prev_t >= 0) { src.release(prev_t); prev_t = -1; }
            const char *tp = base + HEAD - tail_len;
            const u64 tbytes = tail_len + b.text_bytes;
            u64 cap = tbytes / 160 + 4096, names_cap = cap * 24, runs_cap = cap * 4;
            // One call as a rule.  BNS_TEXT_CAP (records of a few bytes, long names, many runs): what the call took is printed as a job
            // of its own and the next call goes on from there ON THE SAME TEXT with arrays twice the size, until the batch is used up --
            // only the truly unfinished last record goes in front of the next batch.
            u64 used = 0;
            bns_text_info info{};
            bool ok = true;
            for (;;) {
                if (!j) { std::lock_guard<std::mutex> lk(mu); if (!spare_j.empty()) { j = std::move(spare_j.back()); spare_j.pop_back(); } }
                if (!j) j = std::make_unique<TextJob>();
                bns_text_out o{};
                size_text_job(ctx, *j, o, taxon_only, cap, names_cap, runs_cap);
                const char *cp = tp + used;
                const u64 cb = tbytes - used;
                chk(ctx, bns_classify_text(ctx, &cp, &cb, 1, ~0ULL, BNS_TEXT_DEVICE | BNS_TEXT_TRIM_READNO | (b.last ? BNS_TEXT_FINAL : 0), cap, &o, &info), "bns_classify_text");
                t_gpu_parse += info.ms_parse * 1e-3; t_gpu_cls += info.ms_classify * 1e-3;
                if (info.status == BNS_TEXT_CAP) { cap *= 2; names_cap *= 2; runs_cap *= 2; if (info.n_records == 0) continue; }
                used += info.consumed[0];
                j->seq = n_jobs++; j->n_records = info.n_records;
                units_done += info.n_records;
                sink.submit(std::move(j));
                if (info.status == BNS_TEXT_CAP) continue;
                // (a batch without one complete record is not an error as long as more text follows: all of it waits in front of the next one)
                ok = (info.status == BNS_TEXT_OK || (info.status == BNS_TEXT_NO_RECORD && !b.last)) && (!b.last || used == tbytes);
                break;
            }
            t_call += tnow() - t0;
            if (!ok) handed_back = true;
            // the unfinished rest stays where it is until the next batch has taken it
            prev_t = b.tbuf;
            tail_off = (HEAD - tail_len) + used;
            tail_len = tbytes - used;
            if (handed_back) break;
        }
    } catch (const std::exception &e) { failure = e.what(); }
    src.stop();
    if (failure.empty()) failure = src.error();
    if (!failure.empty()) { sink.finish(0, true); die(failure); }
    sink.finish(n_jobs);
    if (timing)
        std::fprintf(stderr, "[timing] BGZF text on the device: %llu jobs, %llu members, %.2f GB of text; header walk %.3f s, pread %.3f (summed over %u readers), inflate calls %.3f (summed over %u handles) of which kernel %.3f, "
                             "classify calls %.3f (their kernels: text %.3f, classify %.3f), format %.3f, write %.3f; page-lock %.3f (summed), first batch inflated after %.3f s, waits: walker for bytes %.3f, inflaters for batches / buffers %.3f (summed), classify for text %.3f%s\n",
                     (unsigned long long)n_jobs, (unsigned long long)src.n_members, src.text_total / 1e9, src.t_split, src.t_read, src.R, src.t_inflate, src.NI, src.t_kernel, t_call, t_gpu_parse, t_gpu_cls,
                     sink.t_format, sink.t_write, src.t_pin, src.t_first_inflated, src.t_wait_walk, src.t_wait_inf, src.t_wait_next, handed_back ? "; the host parser takes the rest" : "");
    return !handed_back;
}

bool bgzf_pair_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    if (!fq2 || c.get_emit_fastq() || std::getenv("BNS_NO_BGZF")) return false;
    if (const char *e = std::getenv("BNS_TEXT_GPU")) if (e[0] == '0') return false;
    for (const char *p : {fq1, fq2}) {
        struct stat st;
        if (::stat(p, &st) != 0 || !S_ISREG(st.st_mode) || !is_bgzf_file(p)) return false;
    }
    return true;
}

// A PAIR of BGZF files, both inflated into device memory (a BgzfDeviceSource each) and paired there: bns_classify_text with two streams
// of device text -- record i of the one file and record i of the other are mates (kseq_declare.h:116-131).  The two files' batches do
// not end at the same record, so each side keeps a WINDOW: what its last call left, with the next batch behind it (the rest copied
// into the room in front of the new batch's text, device to device) whenever less than LOW bytes are left; a call takes the pairs
// both windows hold and says where it stopped in either.  One device, calls in file order.
// -> true: everything was classified; false: text handed back after `units_done` pairs (the host parser reads both files and leaves
// those out)
bool process_bgzf_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    bns_ctx *ctx = c.ctxs_[0];
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) (void)bns_set_timing(ctx, 1);
    const bool want_runs = c.get_emit_kraken() != 0, taxon_only = !want_runs;
    std::mutex mu;
    std::vector<std::unique_ptr<TextJob>> spare_j;
    auto recycle_job = [&](std::unique_ptr<TextJob> j) { std::lock_guard<std::mutex> lk(mu); spare_j.push_back(std::move(j)); };
    TextSink sink(c, ofd, recycle_job);
    BgzfDeviceSource src0(c, fq1), src1(c, fq2);
    struct Side { BgzfDeviceSource *src; int t = -1; u64 off = 0, len = 0; bool exhausted = false; } side[2] = {{&src0}, {&src1}};
    const u64 HEAD = src0.HEAD, LOW = HEAD / 2;
    double t_gpu_parse = 0, t_gpu_cls = 0, t_call = 0;
    bool handed_back = false;
    u64 n_calls = 0;
    std::string failure;
    try {
        for (;;) {
            // a window that has run low takes the next batch of its file behind what is left of it
            for (Side &d : side) {
                while (!d.exhausted && d.len < LOW) {
                    BgzfDeviceSource::Item it;
                    if (!d.src->next(it)) {
                        const std::string e = d.src->error();
                        if (!e.empty()) die(e);
                        d.exhausted = true;
                        break;
                    }
                    char *base = d.src->buf(it.tbuf);
                    if (d.len) chk(ctx, bns_dev_copy(ctx, base + HEAD - d.len, d.src->buf(d.t) + d.off, (size_t)d.len), "bns_dev_copy");
                    if (d.t >= 0) d.src->release(d.t);
                    d.t = it.tbuf; d.off = HEAD - d.len; d.len += it.text_bytes;
                    if (it.last) d.exhausted = true;
                }
            }
            const bool final_call = side[0].exhausted && side[1].exhausted;
            if (final_call && side[0].len == 0 && side[1].len == 0) break;
            std::unique_ptr<TextJob> j;
            { std::lock_guard<std::mutex> lk(mu); if (!spare_j.empty()) { j = std::move(spare_j.back()); spare_j.pop_back(); } }
            if (!j) j = std::make_unique<TextJob>();
            const double t0 = tnow();
            const char *tp[2] = {side[0].t >= 0 ? side[0].src->buf(side[0].t) + side[0].off : nullptr, side[1].t >= 0 ? side[1].src->buf(side[1].t) + side[1].off : nullptr};
            const u64 tb[2] = {side[0].len, side[1].len};
            u64 cap = (tb[0] + tb[1]) / 160 + 4096, names_cap = cap * 24, runs_cap = cap * 4;
            bns_text_info info{};
            for (;;) {
                bns_text_out o{};
                size_text_job(ctx, *j, o, taxon_only, cap, names_cap, runs_cap);
                chk(ctx, bns_classify_text(ctx, tp, tb, 2, ~0ULL, BNS_TEXT_DEVICE | BNS_TEXT_TRIM_READNO | (final_call ? BNS_TEXT_FINAL : 0), cap, &o, &info), "bns_classify_text");
                if (info.status == BNS_TEXT_CAP && info.n_records == 0) { cap *= 2; names_cap *= 2; runs_cap *= 2; continue; }
                break;
            }
            j->seq = n_calls; j->mates = 2; j->n_records = info.n_records;
            for (int s = 0; s < 2; ++s) { side[s].off += info.consumed[s]; side[s].len -= info.consumed[s]; }
            t_call += tnow() - t0;
            t_gpu_parse += info.ms_parse * 1e-3; t_gpu_cls += info.ms_classify * 1e-3;
            units_done += info.n_records / 2;
            sink.submit(std::move(j));
            ++n_calls;
            const bool more_text = (!side[0].exhausted && side[0].len < LOW) || (!side[1].exhausted && side[1].len < LOW);
            if (!(info.status == BNS_TEXT_OK || info.status == BNS_TEXT_CAP || (info.status == BNS_TEXT_NO_RECORD && !final_call))) handed_back = true;
            // (nothing paired and no window about to grow: records longer than a window holds, or one file far behind the other)
            else if (info.n_records == 0 && !more_text && !final_call) handed_back = true;
            if (handed_back) break;
            if (final_call && info.status != BNS_TEXT_CAP) {
                if (side[0].len || side[1].len)           // kseq_declare.h:116-120 / 134-137: one file holds more records than the other
                    std::fprintf(stderr, "[W::%s] the %s file has fewer sequences.\n", "bseq_read", side[0].len ? "2nd" : "1st");
                break;
            }
        }
    } catch (const std::exception &e) { failure = e.what(); }
    src0.stop(); src1.stop();
    if (failure.empty()) failure = src0.error();
    if (failure.empty()) failure = src1.error();
    if (!failure.empty()) { sink.finish(0, true); die(failure); }
    sink.finish(n_calls);
    if (timing)
        std::fprintf(stderr, "[timing] pair of BGZF files, text on the device: %llu calls, %llu + %llu members, %.2f + %.2f GB of text; pread %.3f s (summed), inflate calls %.3f of which kernel %.3f (summed over %u handles), "
                             "classify calls %.3f (their kernels: text %.3f, classify %.3f), format %.3f, write %.3f; first batches inflated after %.3f / %.3f s, classify waited %.3f s for text%s\n",
                     (unsigned long long)n_calls, (unsigned long long)src0.n_members, (unsigned long long)src1.n_members, src0.text_total / 1e9, src1.text_total / 1e9, src0.t_read + src1.t_read,
                     src0.t_inflate + src1.t_inflate, src0.t_kernel + src1.t_kernel, src0.NI + src1.NI, t_call, t_gpu_parse, t_gpu_cls, sink.t_format, sink.t_write,
                     src0.t_first_inflated, src1.t_first_inflated, src0.t_wait_next + src1.t_wait_next, handed_back ? "; the host parser takes the rest" : "");
    return !handed_back;
}

bool pair_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    if (!fq2 || c.get_emit_fastq()) return false;
    if (const char *e = std::getenv("BNS_TEXT_GPU")) if (e[0] == '0') return false;
    for (const char *p : {fq1, fq2}) {
        struct stat st;
        if (::stat(p, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) return false;
        unsigned char m[2] = {0, 0};
        const int f = ::open(p, O_RDONLY);
        if (f < 0) return false;
        const bool plain = ::pread(f, m, 2, 0) == 2 && !(m[0] == 0x1f && m[1] == 0x8b) && (m[0] == '>' || m[0] == '@' || m[0] == '\n');
        ::close(f);
        if (!plain) return false;
    }
    return true;
}

// A PAIR of plain files as text on the device (bns_classify_text with two streams: record i of the one file and record i of the
// other are mates, kseq_declare.h:116-131).  Two files cannot be cut at the same RECORD by byte offsets, so: file 1 is cut into
// blocks at nominal offsets like a single file (block b = the records that start in it: `limit`); file 2 gets blocks of its own
// nominal size -- B scaled by the files' sizes, both hold the same number of records -- read with ROOM on both sides, and every call
// is handed file 2 from where the call in front stopped to the end of its block's buffer.  The device pairs record for record and
// says where it stopped in both.  Blocks are read and uploaded ahead (bns_text_prefetch: both files' buffers), one device, calls in
// file order.  Where file 2 drifts out of its buffer (mates whose sizes differ more in one stretch of the files than the room
// allows), or the kernels hand text back, this path stops: the caller reads both files with the host parser and leaves out the
// units that were printed.  -> true: everything was classified
bool process_text_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    const char *paths[2] = {fq1, fq2};
    int fds[2] = {-1, -1};
    struct FdCloser { int *f; ~FdCloser() { for (int i = 0; i < 2; ++i) if (f[i] >= 0) ::close(f[i]); } } closer{fds};
    u64 fsize[2];
    for (int s = 0; s < 2; ++s) {
        fds[s] = ::open(paths[s], O_RDONLY);
        if (fds[s] < 0) die(std::string("Could not open ") + paths[s] + " for reading.");
        fsize[s] = (u64)::lseek(fds[s], 0, SEEK_END);
    }
    std::fflush(out);
    const int ofd = fileno(out);
    bns_ctx *ctx = c.ctxs_[0];
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    auto env_mb = [](const char *name, u64 dflt) { const char *e = std::getenv(name); return e && std::atol(e) > 0 ? (u64)std::atol(e) << 20 : dflt; };
    u64 B = std::min<u64>(env_mb("BNS_TEXT_BLOCK_MB", 96ull << 20), 1ull << 29);
    u64 ROOM = env_mb("BNS_TEXT_ROOM_MB", 16ull << 20);        // file 2's buffer reaches this far in front of and behind its nominal block
    u64 SLACK = 4ull << 20;
    if (const char *e = std::getenv("BNS_TEXT_BLOCK_BYTES")) { B = (u64)std::max(64L, std::atol(e)); ROOM = std::max<u64>(B, 4096); SLACK = std::max<u64>(B / 2, 2048); }   // (tests)
    const u64 n_blocks = std::max<u64>(1, (fsize[0] + B - 1) / B);
    // file 2's nominal block: file 1's, scaled by the files' sizes (both hold the same records: where file 1 is at b * B, file 2 is at
    // about b * B * size2 / size1 -- NOT size2 / n_blocks: file 1's last block is a partial one, and the difference adds up block by block)
    const u64 B2 = std::max<u64>(1, (u64)((long double)B * (long double)fsize[1] / (long double)std::max<u64>(1, fsize[0])) + 1);
    auto off2 = [&](u64 b) { return (u64)((long double)b * (long double)B * (long double)fsize[1] / (long double)std::max<u64>(1, fsize[0])); };
    unsigned R = (unsigned)std::max(2, std::min(8, usable_cpus() / 2));
    if (const char *e = std::getenv("BNS_TEXT_READERS")) R = (unsigned)std::max(1, std::min(32, std::atoi(e)));
    const size_t PIECE = 8u << 20;
    const bool want_runs = c.get_emit_kraken() != 0, taxon_only = !want_runs;

    struct PairJob {
        u64 seq = 0;
        u64 off[2] = {0, 0};                                   // file offset of text[s][0]
        size_t bytes[2] = {0, 0};
        bool last = false, prefetched = false;
        unsigned pieces_left = 0;
        PinnedBuf text[2];
    };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<PairJob>> spare;
    unsigned jobs_made = 0;
    const unsigned max_jobs = 5;
    struct Piece { PairJob *j; int s; size_t off, len; };
    std::deque<Piece> pieces;
    std::map<u64, std::unique_ptr<PairJob>> loading, loaded;
    u64 next_load = 0;
    bool cancel = false;
    std::string error;
    double t_read = 0, t_call = 0, t_idle = 0;
    u64 n_ahead = 0;
    auto fail_with = [&](const std::string &w) { if (error.empty()) error = w; cancel = true; cv.notify_all(); };
    std::vector<std::unique_ptr<TextJob>> spare_j;
    auto recycle_job = [&](std::unique_ptr<TextJob> j) { std::lock_guard<std::mutex> lk(mu); spare_j.push_back(std::move(j)); cv.notify_all(); };
    TextSink sink(c, ofd, recycle_job);

    auto reader = [&] {
        try {
            for (;;) {
                Piece pc{nullptr, 0, 0, 0};
                {
                    std::unique_lock<std::mutex> lk(mu);
                    for (;;) {
                        if (cancel) return;
                        if (!pieces.empty()) { pc = pieces.front(); pieces.pop_front(); break; }
                        if (next_load < n_blocks && (!spare.empty() || jobs_made < max_jobs)) {
                            std::unique_ptr<PairJob> j;
                            if (!spare.empty()) { j = std::move(spare.back()); spare.pop_back(); }
                            else { j = std::make_unique<PairJob>(); ++jobs_made; }
                            const u64 b = j->seq = next_load++;
                            j->prefetched = false;
                            j->off[0] = b * B;
                            j->bytes[0] = (size_t)std::min<u64>(fsize[0] - j->off[0], B + SLACK);
                            j->last = j->off[0] + j->bytes[0] >= fsize[0];
                            const u64 lo2 = off2(b) > ROOM ? off2(b) - ROOM : 0;
                            const u64 hi2 = (j->last || b + 1 == n_blocks) ? fsize[1] : std::min<u64>(fsize[1], off2(b + 1) + ROOM);
                            j->off[1] = std::min(lo2, fsize[1]);
                            j->bytes[1] = (size_t)(hi2 > j->off[1] ? hi2 - j->off[1] : 0);
                            PairJob *jp = j.get();
                            loading[b] = std::move(j);
                            lk.unlock();
                            jp->text[0].reserve(ctx, (size_t)std::max<u64>(B + SLACK, jp->bytes[0]) + 256);
                            jp->text[1].reserve(ctx, (size_t)std::max<u64>(B2 + 2 * ROOM, jp->bytes[1]) + 256);
                            lk.lock();
                            unsigned np = 0;
                            for (int s = 0; s < 2; ++s)
                                for (size_t o = 0; o < jp->bytes[s]; o += PIECE) { pieces.push_back(Piece{jp, s, o, std::min(PIECE, jp->bytes[s] - o)}); ++np; }
                            jp->pieces_left = np;
                            if (!np) { loaded[b] = std::move(loading[b]); loading.erase(b); }
                            cv.notify_all();
                            continue;
                        }
                        if (next_load >= n_blocks && loading.empty()) return;
                        cv.wait(lk);
                    }
                }
                const double t0 = tnow();
                pread_all(fds[pc.s], pc.j->text[pc.s].p + pc.off, pc.len, pc.j->off[pc.s] + pc.off, "text block");
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu);
                t_read += t1 - t0;
                if (--pc.j->pieces_left == 0) { const u64 b = pc.j->seq; loaded[b] = std::move(loading[b]); loading.erase(b); }
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    };
    std::vector<std::thread> readers;
    for (unsigned r = 0; r < R; ++r) readers.emplace_back(reader);

    bool handed_back = false;
    u64 n_done = 0;
    try {
        u64 pos[2] = {0, 0};                                   // where the next call starts in either file
        for (u64 b = 0; b < n_blocks; ++b) {
            std::unique_ptr<PairJob> j;
            std::unique_ptr<TextJob> tj;
            PairJob *ahead = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                const double tw = tnow();
                cv.wait(lk, [&] { return cancel || loaded.count(b); });
                t_idle += tnow() - tw;
                if (cancel) break;
                j = std::move(loaded[b]); loaded.erase(b);
                auto it = loaded.find(b + 1);
                if (it != loaded.end() && !it->second->prefetched) { ahead = it->second.get(); ahead->prefetched = true; ++n_ahead; }
                if (!spare_j.empty()) { tj = std::move(spare_j.back()); spare_j.pop_back(); }
            }
            if (!tj) tj = std::make_unique<TextJob>();
            // both starts inside their buffers?  (file 1: always, by the limit rule; file 2: as long as it has not drifted by more than ROOM)
            if (pos[0] < j->off[0] || pos[0] > j->off[0] + j->bytes[0] || pos[1] < j->off[1] || pos[1] > j->off[1] + j->bytes[1]) { handed_back = true; break; }
            const double t0 = tnow();
            if (ahead) {
                const char *tp[2] = {ahead->text[0].p, ahead->text[1].p};
                const u64 tb[2] = {ahead->bytes[0], ahead->bytes[1]};
                chk(ctx, bns_text_prefetch(ctx, tp, tb, 2), "bns_text_prefetch");
            }
            const char *tp[2] = {j->text[0].p + (pos[0] - j->off[0]), j->text[1].p + (pos[1] - j->off[1])};
            const u64 tb[2] = {j->off[0] + j->bytes[0] - pos[0], j->off[1] + j->bytes[1] - pos[1]};
            const u64 limit = j->last ? ~0ULL : (j->off[0] + B) - pos[0];
            u64 cap = (tb[0] + tb[1]) / 160 + 4096, names_cap = cap * 24, runs_cap = cap * 4;
            bns_text_info info{};
            for (;;) {
                tj->taxon.resize(ctx, cap);
                bns_text_out o{};
                o.taxon = tj->taxon.data();
                if (!taxon_only) {
                    tj->missing.resize(ctx, cap); tj->ambig.resize(ctx, cap); tj->n_hits.resize(ctx, cap); tj->seq_len.resize(ctx, cap); tj->name_off.resize(ctx, cap + 1);
                    tj->run_start.resize(ctx, cap); tj->n_runs.resize(ctx, cap); tj->names.resize(ctx, names_cap);
                    o.missing = tj->missing.data(); o.ambig = tj->ambig.data(); o.n_hits = tj->n_hits.data(); o.seq_len = tj->seq_len.data();
                    o.name_off = tj->name_off.data(); o.names = tj->names.data(); o.names_cap = names_cap;
                    o.run_start = tj->run_start.data(); o.n_runs = tj->n_runs.data();
                    tj->run_tax.resize(ctx, runs_cap); tj->run_len.resize(ctx, runs_cap);
                    o.run_tax = tj->run_tax.data(); o.run_len = tj->run_len.data(); o.runs_cap = runs_cap;
                }
                chk(ctx, bns_classify_text(ctx, tp, tb, 2, limit, (j->last ? BNS_TEXT_FINAL : 0) | BNS_TEXT_TRIM_READNO, cap, &o, &info), "bns_classify_text");
                if (info.status == BNS_TEXT_CAP) { cap *= 2; names_cap *= 2; runs_cap *= 2; continue; }
                break;
            }
            tj->seq = b; tj->mates = 2; tj->n_records = info.n_records;
            pos[0] += info.consumed[0]; pos[1] += info.consumed[1];
            // done with the block: file 1 handed over everything that starts in it (the last block: whatever pairs there were)
            const bool ok = info.status == BNS_TEXT_OK && (j->last || pos[0] >= j->off[0] + B);
            t_call += tnow() - t0;
            units_done += info.n_records / 2;
            sink.submit(std::move(tj));
            n_done = b + 1;
            { std::lock_guard<std::mutex> lk(mu); spare.push_back(std::move(j)); cv.notify_all(); }
            if (!ok) { handed_back = true; break; }
            if (b + 1 == n_blocks && (pos[0] < fsize[0] || pos[1] < fsize[1])) {
                // kseq_declare.h:116-120 / 134-137: one file holds more records than the other
                std::fprintf(stderr, "[W::%s] the %s file has fewer sequences.\n", "bseq_read", pos[0] < fsize[0] ? "2nd" : "1st");
            }
        }
    } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    { std::lock_guard<std::mutex> lk(mu); cancel = true; cv.notify_all(); }
    for (auto &t : readers) t.join();
    (void)bns_text_prefetch(ctx, nullptr, nullptr, 0);          // (a block uploaded ahead of a call that never came)
    if (!error.empty()) { sink.finish(0, true); die(error); }
    sink.finish(n_done);
    if (timing)
        std::fprintf(stderr, "[timing] pair of files, text on the device: %llu blocks of %llu + %llu MiB, %u readers: pread %.3f s (summed), calls %.3f, format %.3f, write %.3f; "
                             "waited %.3f s for blocks, %llu uploads started ahead of their call%s\n",
                     (unsigned long long)n_done, (unsigned long long)(B >> 20), (unsigned long long)(B2 >> 20), R, t_read, t_call, sink.t_format, sink.t_write, t_idle,
                     (unsigned long long)n_ahead, handed_back ? "; the host parser takes the rest" : "");
    return !handed_back;
}
}  // namespace

void process_dataset(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, unsigned chunk_size, unsigned parser_threads,
                     u64 segment_bytes)
{
    int is_paired = fq2 != nullptr;
    const int fd = fileno(out);
    // one plain FASTA / FASTQ file: parsed, packed and classified on the device from its bytes (process_text_gpu); whatever the kernels
    // hand back (text that is not in their regular form) is parsed here, from the record boundary they stopped at
    u64 text_begin = 0;
    if (!is_pack_container(fq1) && text_gpu_wanted(c, fq1, fq2)) {
        struct stat st;
        text_begin = process_text_gpu(c, fq1, out);
        if (::stat(fq1, &st) == 0 && text_begin >= (u64)st.st_size) return;
    }
    // a BGZF file: members inflated on the device, their text parsed and classified where it lies (process_bgzf_gpu).  Text the kernels
    // hand back: the whole file goes through the host parser, which leaves out the units that were printed already
    u64 skip_units = 0;
    bool quiet_nseq = text_begin != 0;
    if (!is_pack_container(fq1) && bgzf_gpu_wanted(c, fq1, fq2)) {
        if (process_bgzf_gpu(c, fq1, out, skip_units)) return;
        quiet_nseq = true;
    }
    // a pair of BGZF files: both inflated on the device and paired there (process_bgzf_gpu_pair); same rule for what it hands back
    if (!is_pack_container(fq1) && bgzf_pair_gpu_wanted(c, fq1, fq2)) {
        if (process_bgzf_gpu_pair(c, fq1, fq2, out, skip_units)) return;
        quiet_nseq = true;
    }
    // a pair of plain files: both as text on the device, mates by record index (process_text_gpu_pair); same rule for what it hands back
    else if (!is_pack_container(fq1) && pair_gpu_wanted(c, fq1, fq2)) {
        if (process_text_gpu_pair(c, fq1, fq2, out, skip_units)) return;
        quiet_nseq = true;
    }
    // a pre-packed read container (`bonsai pack`): no parser and no packer -- every chunk goes from the file into the page-locked
    // buffers of the GPU call (load_packed_chunk, several loader threads per device: one pread stream is ~6 GB/s)
    const bool packed_in = is_pack_container(fq1);
    int pfd = -1;
    bool has_names = false;
    if (packed_in) {
        if (fq2) die("a read container holds both mates of a pair: give the one file");
        if (c.get_emit_fastq()) die("FASTQ-style output needs bases and qualities, which a read container does not hold (classify the FASTQ itself, or use -F)");
        pfd = ::open(fq1, O_RDONLY);
        if (pfd < 0) die(std::string("Could not open ") + fq1 + " for reading.");
        PackFileHeader fh;
        pread_all(pfd, &fh, sizeof(fh), 0, "file header");
        if (fh.version != 1) die("read container: unknown version");
        is_paired = (fh.flags & 1u) ? 1 : 0;
        has_names = (fh.flags & 2u) != 0;
    }
    // A pipeline of 2 + 2 G threads, G = devices (classifier.h:296-337 has one loop; its kt_forpool fan-out is the GPU call here):
    //   reader      assembles chunks (kseq semantics) and numbers them;
    //   G packers   one per device: takes the next WHOLE chunk and packs its sequences (2 bits per base, -p / G threads) into the
    //               page-locked buffers that travel with the chunk's result;
    //   G callers   one per device, each with its own context: the GPU call on the packed chunk, while the packer is on the next
    //               one -- no device waits for another (round 2 split every chunk G ways and joined all devices per chunk);
    //   formatter   takes finished chunks IN INPUT ORDER, turns results into text on -p threads and writes it.
    // At most 4 G chunks are in flight (read but not yet written).
    const unsigned G = (unsigned)c.ctxs_.size();
    struct Job { u64 seq = 0; std::unique_ptr<ReadChunk> seqs; std::unique_ptr<ChunkResult> res; u64 off = 0; PackChunkHeader hdr{}; u64 first_unit = 0; };
    u64 units_read = 0;                                        // (the reader's: units in the chunks numbered so far)
    std::vector<std::unique_ptr<ReadChunk>> seq_pool;          // container input: recycled record arrays (under mu)
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> todo;                                      // read, not yet taken by a device
    std::map<u64, Job> done;                                   // classified, waiting for their turn at the formatter
    std::vector<std::unique_ptr<ChunkResult>> spare;           // recycled result buffers
    u64 n_read = 0, n_written = 0;                             // chunks numbered so far / chunks the formatter is done with
    unsigned callers_left = G;
    bool reader_done = false, cancel = false;
    std::string error;
    auto fail_with = [&](const std::string &what) {            // (called with mu held)
        if (error.empty()) error = what;
        cancel = true;
        cv.notify_all();
    };
    std::unique_ptr<ChunkSource> source_p;
    if (!packed_in) source_p.reset(new ChunkSource(fq1, fq2, chunk_size, parser_threads, segment_bytes, nullptr, text_begin));
    // BNS_CLI_TRACE=<file>: when each stage worked on each chunk (stage, chunk, begin, end in seconds since the start), one line each
    struct Ev { char stage; u64 seq; double t0, t1; };
    std::vector<Ev> trace;
    std::mutex trace_mu;
    const char *trace_path = std::getenv("BNS_CLI_TRACE");
    const double t_origin = tnow();
    auto mark = [&](char stage, u64 seq, double t0) {
        if (!trace_path) return;
        const double t1 = tnow();
        std::lock_guard<std::mutex> lk(trace_mu);
        trace.push_back(Ev{stage, seq, t0 - t_origin, t1 - t_origin});
    };
    std::thread reader([&] {
        try {
            u64 pack_at = sizeof(PackFileHeader);
            const u64 pack_size = packed_in ? (u64)::lseek(pfd, 0, SEEK_END) : 0;
            for (;;) {
                const double tr0 = tnow();
                std::unique_ptr<ReadChunk> seqs;
                u64 off = 0;
                PackChunkHeader hdr{};
                if (packed_in) {                                     // walk the chunk headers; the payload is read by the loader threads
                    if (pack_at + sizeof(PackChunkHeader) > pack_size) break;
                    pread_all(pfd, &hdr, sizeof(hdr), pack_at, "chunk header");
                    if (hdr.magic != PACK_CHUNK_MAGIC || pack_at + sizeof(hdr) + hdr.payload_bytes > pack_size) die("read container: damaged chunk header");
                    off = pack_at;
                    pack_at += sizeof(hdr) + hdr.payload_bytes;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (!seq_pool.empty()) { seqs = std::move(seq_pool.back()); seq_pool.pop_back(); }
                    }
                    if (!seqs) seqs = std::make_unique<ReadChunk>();
                } else seqs = source_p->next();
                if (!seqs) break;
                mark('R', n_read, tr0);
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return n_read - n_written < 4ull * G + 4 || cancel; });     // (chunks in flight: two packers and a caller per device, two formatters)
                if (cancel) break;
                Job j; j.seq = n_read++; j.seqs = std::move(seqs); j.off = off; j.hdr = hdr;
                j.first_unit = units_read;
                units_read += packed_in ? hdr.n_reads / (is_paired ? 2u : 1u) : j.seqs->recs.size() / (is_paired ? 2u : 1u);
                todo.push_back(std::move(j));
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
        std::lock_guard<std::mutex> lk(mu);
        reader_done = true;
        cv.notify_all();
    });
    std::fflush(out);                                          // (what the caller may have put into the FILE goes first)
    auto write_all = [&](const char *p, size_t n) {
        for (size_t off = 0; off < n;) {
            const ssize_t w = ::write(fd, p + off, n - off);
            if (w <= 0) die("write failed");
            off += (size_t)w;
        }
    };
    // the writer: write(2) of one chunk's text while the formatter works on the next chunk's
    // (NF formatter threads take alternate chunks -- with Kraken lines the formatter was the longest stage once the packers were two --
    // into 2 NF sets of buffers; the writer takes the sets in chunk order, the raw taxon file (-b) with them)
    constexpr unsigned NF = 2, NSETS = 2 * NF;
    std::vector<ClassifierGeneric::Work::Part> out_sets[NSETS];
    std::vector<u32> w_taxa[NSETS];
    std::mutex wmu;
    std::condition_variable wcv;
    bool w_pending[NSETS] = {}, w_stop = false, w_failed = false;
    unsigned w_parts[NSETS] = {};
    u64 w_seq[NSETS] = {}, w_next = 0;
    std::thread writer([&] {
        try {
            for (;;) {
                unsigned set, n_parts;
                {
                    std::unique_lock<std::mutex> lk(wmu);
                    wcv.wait(lk, [&] { return (w_pending[w_next % NSETS] && w_seq[w_next % NSETS] == w_next) || w_stop; });
                    if (!(w_pending[w_next % NSETS] && w_seq[w_next % NSETS] == w_next)) break;
                    set = (unsigned)(w_next % NSETS); n_parts = w_parts[set];
                }
                const double tw = tnow();
                for (unsigned t = 0; t < n_parts; ++t) {
                    const ClassifierGeneric::Work::Part &part = out_sets[set][t];
                    write_all(part.p, part.n);
                    write_all(part.s.data(), part.s.size());
                }
                if (c.taxon_out_ && !w_taxa[set].empty())
                    if (std::fwrite(w_taxa[set].data(), 4, w_taxa[set].size(), c.taxon_out_) != w_taxa[set].size()) die("write failed (taxon file)");
                c.work_.t_write += tnow() - tw;
                mark('W', w_next, tw);
                std::lock_guard<std::mutex> lk(wmu);
                w_pending[set] = false; ++w_next;
                wcv.notify_all();
            }
        } catch (const std::exception &e) {
            { std::lock_guard<std::mutex> lk(wmu); w_failed = true; wcv.notify_all(); }
            std::lock_guard<std::mutex> lk(mu);
            fail_with(e.what());
        }
    });
    auto formatter_fn = [&](unsigned f) {
        try {
            for (u64 next = f;; next += NF) {
                Job job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return done.count(next) || cancel || callers_left == 0; });
                    if (cancel || !done.count(next)) break;         // (every caller has finished and chunk `next` is not there: it never will be)
                    job = std::move(done[next]);
                    done.erase(next);
                }
                if (job.seq == 0 && !quiet_nseq) std::fprintf(stderr, "nseq: %i\n", (int)job.res->n);
                // text of chunk n goes into buffer set n % NSETS, which the writer thread must be done with (chunk n - NSETS)
                const unsigned set = (unsigned)(job.seq % NSETS);
                {
                    std::unique_lock<std::mutex> lk(wmu);
                    wcv.wait(lk, [&] { return !w_pending[set] || w_failed; });
                    if (w_failed) break;
                }
                const double tf0 = tnow();
                const unsigned n_units_job = job.res->n / (is_paired ? 2u : 1u);
                const unsigned skip_here = (unsigned)std::min<u64>(n_units_job, skip_units > job.first_unit ? skip_units - job.first_unit : 0);
                const unsigned n_parts = format_chunk_parts(c, job.seqs->recs.data(), *job.res, &out_sets[set], skip_here);
                w_taxa[set].clear();
                if (c.taxon_out_ && job.res->n) w_taxa[set].assign(job.res->taxon.data() + skip_here, job.res->taxon.data() + n_units_job);
                mark('F', job.seq, tf0);
                {
                    std::lock_guard<std::mutex> lk(wmu);
                    w_pending[set] = true; w_parts[set] = n_parts; w_seq[set] = job.seq;
                    wcv.notify_all();
                }
                if (!packed_in) source_p->recycle(std::move(job.seqs));   // (the chunk's text blocks go back before the reader is woken)
                std::lock_guard<std::mutex> lk(mu);
                if (packed_in) { job.seqs->clear(); seq_pool.push_back(std::move(job.seqs)); }
                spare.push_back(std::move(job.res));
                ++n_written;
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
    };
    std::vector<std::thread> formatters;
    for (unsigned f = 0; f < NF; ++f) formatters.emplace_back(formatter_fn, f);
    // per device: a packer thread (takes the next whole chunk, packs it into the result's page-locked buffers) and a caller thread
    // (the GPU call); one packed chunk may wait between them
    std::vector<std::deque<Job>> packed(G);
    // (FASTQ input: the packer's own thread spends as long outside bns_pack_reads_ptrs -- gathering the records' pointers and
    // lengths out of 64 bytes per record, resizing, recycling the text blocks -- as inside it, and with one packer that thread was
    // the pipeline's longest stage: two take alternate chunks; BNS_CLI_PACKERS overrides)
    unsigned packers_per_dev = packed_in ? 4u : 2u;
    if (const char *e = std::getenv("BNS_CLI_PACKERS")) packers_per_dev = (unsigned)std::max(1, std::min(8, std::atoi(e)));
    std::vector<unsigned> packers_left(G, packers_per_dev);
    auto packer = [&](unsigned g) {
        try {
            for (;;) {
                Job job;
                {
                    const double tq = tnow();
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return (!todo.empty() && packed[g].empty()) || (todo.empty() && reader_done) || cancel; });
                    if (g == 0) c.work_.t_wait += tnow() - tq;
                    if (cancel || todo.empty()) break;
                    job = std::move(todo.front());
                    todo.pop_front();
                    if (!spare.empty()) { job.res = std::move(spare.back()); spare.pop_back(); }
                }
                if (!job.res) job.res = std::make_unique<ChunkResult>();
                const double tp0 = tnow();
                if (packed_in) load_packed_chunk(c, c.ctxs_[g], pfd, job.off, job.hdr, is_paired != 0, has_names, *job.seqs, *job.res);
                else {
                    unsigned n = (unsigned)job.seqs->recs.size();
                    n -= n % (is_paired ? 2u : 1u);
                    pack_chunk(c, c.ctxs_[g], job.seqs->recs.data(), n, is_paired, *job.res, (unsigned)std::max(1, c.nt_ / (int)G));
                }
                mark('P', job.seq, tp0);
                std::lock_guard<std::mutex> lk(mu);
                packed[g].push_back(std::move(job));
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
        std::lock_guard<std::mutex> lk(mu);
        --packers_left[g];
        cv.notify_all();
    };
    auto caller = [&](unsigned g) {
        double t_gpu = 0;
        try {
            for (;;) {
                Job job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !packed[g].empty() || packers_left[g] == 0 || cancel; });
                    if (cancel || packed[g].empty()) break;
                    job = std::move(packed[g].front());
                    packed[g].pop_front();
                    cv.notify_all();                                 // (the packer may take the next chunk)
                }
                const double t0 = tnow();
                call_chunk(c.ctxs_[g], *job.res);
                mark('G', job.seq, t0);
                t_gpu += tnow() - t0;
                std::lock_guard<std::mutex> lk(mu);
                c.work_.t_pack += job.res->t_pack; c.work_.t_call += job.res->t_call; c.work_.t_copy += job.res->t_copy;
                const u64 seq = job.seq;
                done[seq] = std::move(job);
                cv.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); }
        std::lock_guard<std::mutex> lk(mu);
        c.work_.t_gpu += t_gpu;
        --callers_left;
        cv.notify_all();
    };
    std::vector<std::thread> workers;
    for (unsigned g = 0; g < G; ++g) for (unsigned t = 0; t < packers_per_dev; ++t) workers.emplace_back(packer, g);
    for (unsigned g = 1; g < G; ++g) workers.emplace_back(caller, g);
    caller(0);                                                 // (this thread is device 0's caller)
    for (auto &t : workers) t.join();
    for (auto &t : formatters) t.join();
    { std::lock_guard<std::mutex> lk(wmu); w_stop = true; wcv.notify_all(); }
    writer.join();                                             // (writes what is still pending first)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!error.empty()) cancel = true;
        cv.notify_all();
    }
    reader.join();                                             // (after a cancel it stops at the end of the chunk it is parsing)
    if (!error.empty()) die(error);
    if (n_read == 0) std::fprintf(stderr, "Could not get any sequences from file, fyi.\n");
    if (trace_path)
        if (std::FILE *tf = std::fopen(trace_path, "w")) {
            for (const Ev &e : trace) std::fprintf(tf, "%c\t%llu\t%.6f\t%.6f\n", e.stage, (unsigned long long)e.seq, e.t0, e.t1);
            std::fclose(tf);
        }
    if (pfd >= 0) ::close(pfd);
    if (std::getenv("BNS_CLI_TIMING") && !packed_in)
        std::fprintf(stderr, "[timing] reader: bseq_read %.3f s%s, of which waiting for file blocks %.3f s\n", source_p->parse_seconds(),
                     source_p->stretches() > 1 ? (" summed over the parser threads (" + std::to_string(source_p->stretches()) + " stretches)").c_str() : "",
                     source_p->blocked_seconds());
    if (std::getenv("BNS_CLI_TIMING"))
        std::fprintf(stderr, "[timing] wait-for-reader %.3f s  pack + gpu call (sum over %u devices) %.3f = pack %.3f + call %.3f + copy-out %.3f  format %.3f  write %.3f\n",
                     c.work_.t_wait, G, c.work_.t_gpu, c.work_.t_pack, c.work_.t_call, c.work_.t_copy, c.work_.t_format, c.work_.t_write);
}

// ---------------------------------------------------------------------------------------------- db construction
std::vector<std::pair<std::string, tax_t>> build_name_hash(const char *fn)
{
    std::ifstream is(fn);
    if (!is) die(std::string("Could not open seq2tax map ") + fn);
    std::vector<std::pair<std::string, tax_t>> v;
    std::string line;
    while (std::getline(is, line)) {
        if (line.empty() || line[0] == '#') continue;                   // util.h:706
        const size_t tab = line.find('\t');
        const std::string name = line.substr(0, tab);
        const tax_t id = tab == std::string::npos ? 0 : (tax_t)std::atoi(line.c_str() + tab + 1);
        v.emplace_back(name, id);
    }
    // later lines overwrite earlier ones (util.h:709-719): keep the last occurrence of every name
    std::stable_sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    std::vector<std::pair<std::string, tax_t>> out;
    for (size_t i = 0; i < v.size(); ++i)
        if (i + 1 == v.size() || v[i + 1].first != v[i].first) out.push_back(v[i]);
    return out;
}

std::string genome_name(const std::string &line)
{
    if (line.find('|') != std::string::npos) {                          // util.h:913-919
        const size_t last = line.rfind('|');
        size_t q = last;
        while (q > 0 && line[--q] != '|') {}
        const size_t start = line[q] == '|' ? q + 1 : q;
        return line.substr(start, line.find('|', start) - start);
    }
    size_t e = 0;
    while (e < line.size() && !std::isspace((unsigned char)line[e])) ++e;
    return line.substr(0, e);
}

tax_t get_taxid(const char *path, const std::vector<std::pair<std::string, tax_t>> &names)
{
    gzFile fp = gzopen(path, "rb");
    if (!fp) die(std::string("Could not read from file ") + path);
    char buf[2048];
    char *line = gzgets(fp, buf, sizeof(buf));
    gzclose(fp);
    if (!line) die(std::string("zlib error reading ") + path);
    std::string l(line + 1);                                             // skip '>' (util.h:909)
    while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
    const std::string name = genome_name(l);
    auto it = std::lower_bound(names.begin(), names.end(), name, [](const auto &a, const std::string &b) { return a.first < b; });
    return (it != names.end() && it->first == name) ? it->second : 1u;  // util.h:924: unknown -> 1
}

namespace {
struct DevMem {
    bns_ctx *ctx; void *p = nullptr;
    DevMem(bns_ctx *c, size_t n) : ctx(c) { chk(c, bns_dev_alloc(c, n, &p), "bns_dev_alloc"); }
    ~DevMem() { if (p) bns_dev_free(ctx, p); }
    DevMem(const DevMem &) = delete; DevMem &operator=(const DevMem &) = delete;
};
}  // namespace

Database lca_map(const std::vector<std::string> &paths, const std::vector<u32> &parent, const char *seq2tax_path,
                 const BuildOptions &opt)
{
    const auto names = build_name_hash(seq2tax_path);
    // every FASTA record of every genome file is one sequence under the genome's taxid (feature_min.h:67-82)
    std::string bases;
    std::vector<u64> offsets{0};
    std::vector<u32> taxids;
    for (const std::string &path : paths) {
        const tax_t tx = get_taxid(path.c_str(), names);
        SeqReader rd(path.c_str());
        bseq1_t rec;
        while (rd.read(rec) >= 0) {
            bases += rec.seq;
            offsets.push_back(bases.size());
            taxids.push_back(tx);
        }
    }
    if (taxids.empty()) die("Need input files from command line or file. See usage.");
    const unsigned k = opt.k;
    spvec_t gaps = opt.spacing.empty() ? spvec_t(k - 1, 0) : opt.spacing;
    if (gaps.size() + 1 != k) die("Error: input vector must have size 1 less than k.");       // spacer.h:65-68
    unsigned c = k;
    for (u16 g : gaps) c += g;
    const unsigned w = std::max<int>((int)c, std::max<int>(opt.wsz, (int)k));                 // bonsai.cpp:217 + spacer.h:61
    const unsigned span = w > c ? w : c;

    bns_ctx *ctx = nullptr;
    chk(nullptr, bns_create(opt.device, &ctx), "bns_create");
    struct Guard { bns_ctx *c; ~Guard() { bns_destroy(c); } } guard{ctx};
    chk(ctx, bns_set_encoder(ctx, k, gaps.data(), opt.canon ? 1 : 0, 1), "bns_set_encoder");
    chk(ctx, bns_set_window(ctx, w, opt.entropy ? BNS_SCORE_ENTROPY_PATH : BNS_SCORE_LEX), "bns_set_window");
    chk(ctx, bns_load_taxonomy(ctx, parent.data(), (u32)parent.size()), "bns_load_taxonomy");

    u64 upper = 0;                                                                             // emitted values <= this
    for (size_t i = 0; i + 1 < offsets.size(); ++i) {
        const u64 L = offsets[i + 1] - offsets[i];
        if (L >= span) upper += L - span + 1;
        else if (L >= c) upper += 1;        // the emitted-stream modes (-C -w, real-entropy) flush one minimum for a sequence
    }                                       // that never fills a window (encoder.h:304-305)
    bases.resize(bases.size() + 8, 'N');                                                       // 4-byte readable tail
    DevMem d_bases(ctx, bases.size()), d_off(ctx, offsets.size() * 8), d_tx(ctx, taxids.size() * 4);
    chk(ctx, bns_dev_upload(ctx, d_bases.p, bases.data(), bases.size()), "upload");
    chk(ctx, bns_dev_upload(ctx, d_off.p, offsets.data(), offsets.size() * 8), "upload");
    chk(ctx, bns_dev_upload(ctx, d_tx.p, taxids.data(), taxids.size() * 4), "upload");

    auto pow2_for = [](u64 keys) { u64 nb = 4; while ((u64)(nb * 0.77 + 0.5) <= keys) nb <<= 1; return nb; };
    // windowed dbs keep roughly 2/(ws+1) of the positions; start there and grow on BNS_ERR_TABLE
    const u64 ws = w - c + 1;
    u64 nb = pow2_for(ws > 1 ? std::max<u64>(1024, upper * 3 / (ws + 1)) : upper);
    Database db;
    for (int attempt = 0; attempt < 40; ++attempt) {
        DevMem d_f(ctx, (nb < 16 ? 1 : nb >> 4) * 4), d_k(ctx, nb * 8), d_v(ctx, nb * 4);
        u64 hdr[4] = {0, 0, 0, 0};
        const int rc = bns_build_table_device(ctx, (const char *)d_bases.p, (const u64 *)d_off.p, taxids.size(),
                                              offsets.back(), (const u32 *)d_tx.p, nb, (u32 *)d_f.p, (u64 *)d_k.p, (u32 *)d_v.p,
                                              hdr, nullptr);
        if (rc == BNS_ERR_TABLE) { nb <<= 1; continue; }                                       // load factor would exceed 0.77
        chk(ctx, rc, "bns_build_table_device");
        const u64 want = pow2_for(hdr[2]);
        if (want < nb) { nb = want; continue; }                                                // compact to khash's own size
        db.k_ = k; db.w_ = w; db.s_ = gaps;
        db.db_.n_buckets = hdr[0]; db.db_.n_occupied = hdr[1]; db.db_.size = hdr[2]; db.db_.upper_bound = hdr[3];
        db.db_.flags.resize(nb < 16 ? 1 : nb >> 4); db.db_.keys.resize(nb); db.db_.vals.resize(nb);
        chk(ctx, bns_dev_download(ctx, db.db_.flags.data(), d_f.p, db.db_.flags.size() * 4), "download");
        chk(ctx, bns_dev_download(ctx, db.db_.keys.data(), d_k.p, nb * 8), "download");
        chk(ctx, bns_dev_download(ctx, db.db_.vals.data(), d_v.p, nb * 4), "download");
        return db;
    }
    die("could not size the hash table");
}

// ---------------------------------------------------------------------------------------------- Encoder
namespace {
// The reference's path overloads read .xz / .bz2 / .zst through `xz|bzip2|zstd -dc <path>` (encoder.h:516-523, 826-833); so
// does this: the child's stdout is opened by SeqReader as /dev/fd/N.  Everything else goes to SeqReader as it is (plain, gzip,
// BGZF).
struct PathInput {
    std::FILE *pfp = nullptr;
    std::unique_ptr<SeqReader> reader;
    explicit PathInput(const char *path)
    {
        const std::string p(path);
        const bool xz = ends_with(p, ".xz"), bz = ends_with(p, ".bz2"), zst = ends_with(p, ".zst");
        if (xz || bz || zst) {
            if (::access(path, R_OK) != 0) die(std::string("Could not open file at ") + path);
            std::string quoted = "'";
            for (char c : p) { if (c == '\'') quoted += "'\\''"; else quoted += c; }
            quoted += "'";
            const std::string cmd = std::string(xz ? "xz" : (bz ? "bzip2" : "zstd")) + " -dc " + quoted;
            pfp = ::popen(cmd.c_str(), "r");
            if (!pfp) die("Failed to open popen call: " + cmd);
            reader.reset(new SeqReader(("/dev/fd/" + std::to_string(::fileno(pfp))).c_str()));
        } else {
            reader.reset(new SeqReader(path));
        }
    }
    ~PathInput() { reader.reset(); if (pfp) ::pclose(pfp); }
};

// records of `in`, batches of whole records (<= batch_bases of sequence per device call): call(bases, offsets, n_records)
template <typename Call>
void for_record_batches(SeqReader &in, size_t batch_bases, const Call &call)
{
    std::string bases;
    std::vector<u64> offsets{0};
    bseq1_t rec;
    for (;;) {
        const int rc = in.read(rec);
        if (rc >= 0) {
            bases.append(rec.seq.data(), rec.seq.size());
            offsets.push_back(bases.size());
        }
        if ((rc < 0 && offsets.size() > 1) || bases.size() >= batch_bases) {
            call(bases.data(), offsets.data(), (u64)offsets.size() - 1);
            bases.clear();
            offsets.assign(1, 0);
        }
        if (rc < 0) break;                                     // -1 end of input, -2 truncated record: kseq_read < 0 ends the reference's loop too
    }
}
constexpr size_t PATH_BATCH_BASES = 32u << 20;
}  // namespace

Encoder::Encoder(unsigned k, const spvec_t &gaps, bool canonicalize, int device, unsigned w, int score)
    : k_(k), w_(w), score_(score), canon_(canonicalize), spaced_(false), gaps_(gaps)
{
    for (u16 g : gaps) spaced_ |= g != 0;
    if (spaced_) canon_ = false;                                // encoder.h:148-150
    chk(nullptr, bns_create(device, &ctx_), "bns_create");
    configure(false, canon_);
}

Encoder::~Encoder() { if (ctx_) bns_destroy(ctx_); }

// string rules: the reference's for_each(func, str, len), SURVEY F7 for a spaced seed included (it emits nothing) and the string
// form of the entropy score; path rules: what the path overloads dispatch to (for_each_uncanon_spaced; the path form, F8)
void Encoder::configure(bool path_rules, bool canon)
{
    const int want = (path_rules ? 2 : 0) | (canon ? 1 : 0);
    if (configured_ == want) return;
    chk(ctx_, bns_set_encoder(ctx_, k_, gaps_.empty() ? nullptr : gaps_.data(), canon ? 1 : 0, path_rules ? 1 : 0), "bns_set_encoder");
    if (w_) {
        int score = score_;
        if (path_rules && score == BNS_SCORE_ENTROPY_STRING) score = BNS_SCORE_ENTROPY_PATH;
        if (!path_rules && score == BNS_SCORE_ENTROPY_PATH) score = BNS_SCORE_ENTROPY_STRING;
        chk(ctx_, bns_set_window(ctx_, w_, score), "bns_set_window");
    }
    configured_ = want;
}

void Encoder::fetch(const char *str, u64 l)
{
    configure(false, canon_);
    const u64 offsets[2] = {0, l};
    kmers_.assign(l + 1, 0);
    u32 n = 0;
    chk(ctx_, bns_encode_batch(ctx_, str, offsets, 1, kmers_.data(), &n), "bns_encode_batch");
    kmers_.resize(n);
}

void Encoder::fetch_hash(const char *str, u64 l, unsigned k, const u64 *table256)
{
    configure(false, canon_);
    const u64 offsets[2] = {0, l};
    kmers_.assign(l + 1, 0);
    u32 n = 0;
    chk(ctx_, bns_for_each_hash_batch(ctx_, str, offsets, 1, k, -1, table256, kmers_.data(), &n), "bns_for_each_hash_batch");
    kmers_.resize(n);
}

void Encoder::each_path(const char *path, PathMode mode, const Sink &sink)
{
    PathInput in(path);
    // encoder.h:524-525: canonicalize_ picks for_each_canon / for_each_uncanon; a spaced seed is never canonical (:148-150)
    const bool canon = mode == PATH_AUTO || mode == PATH_HASH ? canon_ : (mode == PATH_CANON && !spaced_);
    if (mode == PATH_HASH) configure(false, canon_); else configure(true, canon);
    std::vector<u32> cnt;
    for_record_batches(*in.reader, PATH_BATCH_BASES, [&](const char *bases, const u64 *offsets, u64 n) {
        kmers_.resize(offsets[n] + 1);
        cnt.resize(n);
        if (mode == PATH_HASH) chk(ctx_, bns_for_each_hash_batch(ctx_, bases, offsets, n, 0, -1, nullptr, kmers_.data(), cnt.data()), "bns_for_each_hash_batch");
        else chk(ctx_, bns_encode_batch(ctx_, bases, offsets, n, kmers_.data(), cnt.data()), "bns_encode_batch");
        for (u64 r = 0; r < n; ++r)
            if (cnt[r]) sink(kmers_.data() + offsets[r], cnt[r]);
    });
    kmers_.clear();
}

// ---------------------------------------------------------------------------------------------- RollingHasher
namespace detail {
RollingCore::RollingCore(unsigned bits, unsigned k, bool canon, int enc, long long wsz, u64 seed1, u64 seed2, int device)
    : bits_(bits), k_(k), canon_(canon), seed1_(seed1), seed2_(seed2)
{
    if (enc != 0) die("RollingHasher: only the DNA alphabet is supported (protein alphabets are outside this path)");
    if (bits != 64 && bits != 128) die("RollingHasher: 64- or 128-bit words");
    window(wsz);
    chk(nullptr, bns_create(device, &ctx_), "bns_create");
    const size_t words = bits == 64 ? 256 : 512;
    fwd_.resize(words); rc_.resize(words);
    // encoder.h:682-683: hasher_.seed(seed1, seed2); rchasher_.seed(seed2 * seed1, seed2 ^ seed1)
    if (bits == 64) chk(ctx_, bns_rolling_tables(seed1, seed2, fwd_.data(), rc_.data()), "bns_rolling_tables");
    else chk(ctx_, bns_rolling_tables128(seed1, seed2, fwd_.data(), rc_.data()), "bns_rolling_tables128");
}

RollingCore::~RollingCore() { if (ctx_) bns_destroy(ctx_); }

void RollingCore::run(const char *bases, const u64 *offsets, u64 n, bool canon, const Sink &sink)
{
    const bool windowed = w_ > (long long)k_;
    const u64 per_base = (windowed && canon) ? 2 : 1;          // the canonical windowed form queues both strands: two values per base at most
    const u64 words = bits_ == 64 ? 1 : 2;
    out_.resize(per_base * words * offsets[n] + 2);
    cnt_.resize(n);
    int rc;
    if (bits_ == 64)
        rc = windowed ? bns_rolling_hash_windowed_batch(ctx_, bases, offsets, n, k_, canon ? 1 : 0, (u32)w_, fwd_.data(), rc_.data(), out_.data(), cnt_.data())
                      : bns_rolling_hash_batch(ctx_, bases, offsets, n, k_, canon ? 1 : 0, fwd_.data(), rc_.data(), out_.data(), cnt_.data());
    else
        rc = windowed ? bns_rolling_hash128_windowed_batch(ctx_, bases, offsets, n, k_, canon ? 1 : 0, (u32)w_, fwd_.data(), rc_.data(), out_.data(), cnt_.data())
                      : bns_rolling_hash128_batch(ctx_, bases, offsets, n, k_, canon ? 1 : 0, fwd_.data(), rc_.data(), out_.data(), cnt_.data());
    chk(ctx_, rc, "bns_rolling_hash_batch");
    for (u64 r = 0; r < n; ++r)
        if (cnt_[r]) sink(out_.data() + per_base * words * offsets[r], cnt_[r]);
}

void RollingCore::each_str(const char *s, size_t l, bool canon, const Sink &sink)
{
    const u64 offsets[2] = {0, l};
    run(s, offsets, 1, canon, sink);
}

void RollingCore::each_path(const char *path, bool canon, const Sink &sink)
{
    PathInput in(path);
    for_record_batches(*in.reader, PATH_BATCH_BASES, [&](const char *bases, const u64 *offsets, u64 n) { run(bases, offsets, n, canon, sink); });
}
}  // namespace detail

}  // namespace bns
