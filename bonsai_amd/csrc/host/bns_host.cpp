// bns_host.cpp -- host side of the classify path (see bns_host.hpp for the reference map).
#include "bns_host_internal.hpp"

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
void *big_alloc(size_t bytes)
{
    if (bytes < (1u << 20)) { void *p = std::malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return p; }
    const size_t want = (bytes + (2u << 20) - 1) & ~size_t((2u << 20) - 1);
    void *p = ::mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) throw std::bad_alloc();
    static const bool huge = [] { const char *e = std::getenv("BNS_DB_HUGE"); return !e || e[0] != '0'; }();      // (BNS_DB_HUGE=0: 4 KiB pages -- the read 0.25-0.6 s instead of 0.05, the upload no steadier)
    if (huge) (void)::madvise(p, want, MADV_HUGEPAGE);
    return p;
}
void big_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes < (1u << 20)) std::free(p);
    else ::munmap(p, (bytes + (2u << 20) - 1) & ~size_t((2u << 20) - 1));
}

namespace {
// A bns.db that is not compressed (what `bonsai build` writes unless its name ends in .gz), read where it lies: the header by one
// pread, the three arrays by several threads into memory nobody has touched yet -- 6.6 GB (2^29 buckets) in ~0.4 s out of the page
// cache where zlib's transparent gzread took 1.4 s behind a second of zero-filling.  -> false: not this width / not a plain bns.db
bool read_plain_db(const char *path, int width, unsigned &k_out, unsigned &w_out, spvec_t &sp_out, KhashC &t)
{
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) die(std::string("Could not open ") + path + " for reading.");
    struct Closer { int fd; ~Closer() { ::close(fd); } } closer{fd};
    const u64 fsize = (u64)::lseek(fd, 0, SEEK_END);
    unsigned char head[8 + 2 * 32 + 32];
    const ssize_t got = ::pread(fd, head, sizeof(head), 0);
    if (got < 8 + 32 || (head[0] == 0x1f && head[1] == 0x8b)) return false;
    u32 k, w;
    std::memcpy(&k, head, 4); std::memcpy(&w, head + 4, 4);
    if (k < 1 || k > 32) return false;
    const size_t sp_bytes = (size_t)(k - 1) * (size_t)width, hdr_at = 8 + sp_bytes;
    if ((size_t)got < hdr_at + 32) return false;
    spvec_t sp(k - 1);
    for (u32 i = 0; i + 1 < k; ++i) {
        if (width == 1) sp[i] = head[8 + i];
        else { u16 v; std::memcpy(&v, head + 8 + 2 * i, 2); sp[i] = v; }
    }
    u64 hdr[4];
    std::memcpy(hdr, head + hdr_at, 32);
    const u64 nb = hdr[0];
    if (!(nb && !(nb & (nb - 1)) && hdr[2] <= hdr[1] && hdr[1] <= nb && hdr[3] == (u64)(nb * 0.77 + 0.5))) return false;
    const u64 nf = nb < 16 ? 1 : nb >> 4;
    const u64 at_flags = hdr_at + 32, at_keys = at_flags + nf * 4, at_vals = at_keys + nb * 8, end = at_vals + nb * 4;
    if (end != fsize) return false;                              // (the payload ends exactly at the end of the file)
    t.n_buckets = nb; t.n_occupied = hdr[1]; t.size = hdr[2]; t.upper_bound = hdr[3];
    t.flags.resize(nf); t.keys.resize(nb); t.vals.resize(nb);
    struct Part { char *dst; u64 at, n; };
    const Part parts[3] = {{reinterpret_cast<char *>(t.flags.data()), at_flags, nf * 4}, {reinterpret_cast<char *>(t.keys.data()), at_keys, nb * 8},
                           {reinterpret_cast<char *>(t.vals.data()), at_vals, nb * 4}};
    const u64 PIECE = 16u << 20;
    std::vector<Part> pieces;
    for (const Part &p : parts) for (u64 o = 0; o < p.n; o += PIECE) pieces.push_back(Part{p.dst + o, p.at + o, std::min(PIECE, p.n - o)});
    const unsigned nt = (unsigned)std::max(1, std::min<int>(12, std::min<int>(usable_cpus(), (int)pieces.size())));
    std::atomic<size_t> next{0};
    std::atomic<bool> bad{false};
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; ++i)
        th.emplace_back([&] {
            for (size_t j; (j = next.fetch_add(1)) < pieces.size();) {
                const Part &p = pieces[j];
                for (u64 done = 0; done < p.n;) {
                    const ssize_t r = ::pread(fd, p.dst + done, (size_t)(p.n - done), (off_t)(p.at + done));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { bad = true; return; }
                    done += (u64)r;
                }
            }
        });
    for (auto &x : th) x.join();
    if (bad) die(std::string("Error: short read in ") + path);
    k_out = k; w_out = w; sp_out = std::move(sp);
    return true;
}
}  // namespace

Database::Database(const char *path)
{
    // (a plain file: read where it lies, on several threads; either spacing width, as below)
    for (int width = 1; width <= 2; ++width) {
        KhashC t;
        if (read_plain_db(path, width, k_, w_, s_, t)) { db_ = std::move(t); spacing_width_ = width; return; }
    }
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

// ---------------------------------------------------------------------------------------------- formatting


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
        // one PCIe upload of the db, RCCL broadcast over xGMI to the other devices, one re-hash per device.  (The arrays go up from where
        // they lie, pageable: page-locking 6.6 GB of them first -- bns_host_register -- took 0.0-1.8 s depending on how the kernel had backed
        // the pages, against ~0.25 s that the staged copy costs; measured and not kept, tools/r06_db_load.sh)
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
    // (round 6: anything large is memory of our own, registered -- 2 ms per 96 MiB instead of 15-45 under the runtime's lock, and copies
    // from it run at the same 56.7 GB/s once its mapping has been used: tools/micro/pin_bench.hip.  BNS_PIN_MALLOC=1: hipHostMalloc as before)
    static const bool pin_malloc = std::getenv("BNS_PIN_MALLOC") != nullptr;
    if (!pin_malloc && std::max(bytes, 2 * cap) >= (1u << 20)) return reserve_registered(c, bytes);
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

char *PinnedBuf::reserve_registered(bns_ctx *c, size_t bytes)
{
    if (bytes <= cap) return p;
    const size_t want = ((std::max(bytes, 2 * cap) + (2u << 20) - 1) >> 21) << 21;      // (whole 2 MiB pages)
    release();
    ctx = c;
    const double t0 = tnow();
    void *q = ::mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q == MAP_FAILED) die("out of host memory");
    static const bool huge = !std::getenv("BNS_PIN_NOHUGE");
    if (huge) (void)::madvise(q, want, MADV_HUGEPAGE);
    for (size_t o = 0; o < want; o += 4096) static_cast<volatile char *>(q)[o] = 0;       // (resident before it is registered)
    const double t1 = tnow();
    mapped = true;
    pinned = bns_host_register(c, q, want) == BNS_OK;                                      // (not registered: pageable, copies staged by the runtime)
    const double t2 = tnow();
    if (t2 - t0 > 0.05 && std::getenv("BNS_CLI_TIMING"))
        std::fprintf(stderr, "[timing] page-locking %zu MiB took %.3f s (map + touch %.3f, register %.3f)\n", want >> 20, t2 - t0, t1 - t0, t2 - t1);
    p = static_cast<char *>(q); cap = want;
    return p;
}

void PinnedBuf::release()
{
    if (p) {
        if (mapped) { if (pinned) bns_host_unregister(ctx, p); ::munmap(p, cap); }
        else if (pinned) bns_host_free(ctx, p);
        else std::free(p);
    }
    p = nullptr; cap = 0; pinned = false; mapped = false;
}

PinnedBuf::~PinnedBuf() { release(); }


// First half of classify_seqs: gather the chunk's sequences into one buffer and make the ONE C-ABI call that replaces the
// kt_forpool fan-out of classifier.h:275 (with the hit stream already run-length encoded on the device when the output
// prints it).  Everything the formatter needs ends up in r.
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

namespace {
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
            // (a reader that cannot be made must not leave the child behind: the destructor does not run for a constructor that throws)
            try { reader.reset(new SeqReader(("/dev/fd/" + std::to_string(::fileno(pfp))).c_str())); }
            catch (...) { ::pclose(pfp); pfp = nullptr; throw; }
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
