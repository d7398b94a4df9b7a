// bns_host.cpp -- host side of the classify path (see bns_host.hpp for the reference map).
#include "bns_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
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

void append_taxa_runs(tax_t taxon, const std::vector<tax_t> &taxa, std::string &s)   // classifier.h:45-61 (+30-42)
{
    if (!taxon) { s += "0:0\n"; return; }
    size_t i = 0;
    while (i < taxa.size()) {
        size_t j = i;
        while (j < taxa.size() && taxa[j] == taxa[i]) ++j;
        if (taxa[i] == 0) s.push_back('U');
        else if (taxa[i] == (tax_t)-1) s.push_back('A');
        else put_unsigned(s, taxa[i]);
        s.push_back(':'); put_unsigned(s, (u32)(j - i)); s.push_back('\t');
        i = j;
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
    // empty / deleted slots are written as zeros (util.h:282-284)
    std::vector<u64> keys(db_.keys);
    std::vector<u32> vals(db_.vals);
    for (u64 i = 0; i < db_.n_buckets; ++i) if (!db_.exists(i)) { keys[i] = 0; vals[i] = 0; }
    put(db_.flags.data(), db_.flags.size() * 4); put(keys.data(), keys.size() * 8); put(vals.data(), vals.size() * 4);
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
SeqReader::SeqReader(const char *path) : buf_(1 << 18)
{
    fp_ = gzopen(path, "rb");
    if (!fp_) die(std::string("Could not open ") + path + " for reading.");
    gzbuffer(static_cast<gzFile>(fp_), 1 << 18);
}

SeqReader::~SeqReader() { if (fp_) gzclose(static_cast<gzFile>(fp_)); }

int SeqReader::getc_()
{
    if (begin_ >= end_) {
        if (eof_) return -1;
        const int n = gzread(static_cast<gzFile>(fp_), buf_.data(), (unsigned)buf_.size());
        if (n <= 0) { eof_ = true; return -1; }
        begin_ = 0; end_ = (size_t)n;
    }
    return buf_[begin_++];
}

// refill the buffer when it is exhausted; false at end of stream
bool SeqReader::fill_()
{
    if (begin_ < end_) return true;
    if (eof_) return false;
    const int n = gzread(static_cast<gzFile>(fp_), buf_.data(), (unsigned)buf_.size());
    if (n <= 0) { eof_ = true; return false; }
    begin_ = 0; end_ = (size_t)n;
    return true;
}

// append up to (not including) the next '\n' to dst, consume the newline; one trailing '\r' is stripped when the
// accumulated string is longer than one character (kseq's KS_SEP_LINE rule).
// Returns 1 when a newline ended the line, 0 when the stream ended first, -1 when nothing was left to read.
int SeqReader::read_line_(std::string &dst)
{
    int rc = -1;
    for (;;) {
        if (!fill_()) break;
        rc = 0;
        const unsigned char *p = buf_.data() + begin_;
        const void *nl = std::memchr(p, '\n', end_ - begin_);
        const size_t len = nl ? (size_t)((const unsigned char *)nl - p) : end_ - begin_;
        dst.append(reinterpret_cast<const char *>(p), len);
        begin_ += len;
        if (nl) { ++begin_; rc = 1; break; }
    }
    if (dst.size() > 1 && dst.back() == '\r') dst.pop_back();
    return rc;
}

int SeqReader::read(bseq1_t &rec)
{
    int c;
    if (last_char_ == 0) {                                   // jump to the next header line
        while ((c = getc_()) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return -1;
        last_char_ = c;
    }
    rec.name.clear(); rec.comment.clear(); rec.seq.clear(); rec.qual.clear();
    // name = first whitespace-delimited token; comment = rest of the header line
    bool got = false;
    c = -1;
    for (;;) {
        if (!fill_()) { c = -1; break; }
        const unsigned char *p = buf_.data() + begin_;
        size_t i = 0, n = end_ - begin_;
        while (i < n && !std::isspace(p[i])) ++i;
        if (i) { rec.name.append(reinterpret_cast<const char *>(p), i); got = true; }
        begin_ += i;
        if (i < n) { c = buf_[begin_++]; break; }
    }
    if (c < 0 && !got) return -1;
    if (c >= 0 && c != '\n') read_line_(rec.comment);
    while ((c = getc_()) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        rec.seq.push_back((char)c);
        read_line_(rec.seq);
    }
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') { if (c < 0) last_char_ = 0; return (int)rec.seq.size(); }            // FASTA
    { std::string skip; if (read_line_(skip) != 1) return -2; }                         // rest of the '+' line; EOF here = no quality
    while (rec.qual.size() < rec.seq.size()) { if (read_line_(rec.qual) < 0) break; }
    last_char_ = 0;
    if (rec.qual.size() != rec.seq.size()) return -2;
    return (int)rec.seq.size();
}

static void trim_readno(std::string &s)                        // kseq_declare.h:106-110
{
    const size_t l = s.size();
    if (l > 2 && s[l - 2] == '/' && std::isdigit((unsigned char)s[l - 1])) s.resize(l - 2);
}

int bseq_read(int chunk_size, SeqReader &r1, SeqReader *r2, std::vector<bseq1_t> &out)
{
    out.clear();
    long size = 0;
    bseq1_t a, b;
    while (r1.read(a) >= 0) {
        if (r2 && r2->read(b) < 0) { std::fprintf(stderr, "[W::bseq_read] the 2nd file has fewer sequences.\n"); break; }
        trim_readno(a.name);
        size += a.l_seq();
        out.push_back(std::move(a));
        if (r2) { trim_readno(b.name); size += b.l_seq(); out.push_back(std::move(b)); }
        if (size >= chunk_size && (out.size() & 1) == 0) break;
    }
    if (size == 0 && r2 && r2->read(b) >= 0) std::fprintf(stderr, "[W::bseq_read] the 1st file has fewer sequences.\n");
    return (int)out.size();
}

// ---------------------------------------------------------------------------------------------- formatting
void append_kraken_classification(const std::vector<tax_t> &taxa, tax_t taxon, u32 ambig_count, u32 missing_count,
                                  const bseq1_t &bs, std::string &bks)
{
    bks.push_back(taxon ? 'C' : 'U'); bks.push_back('\t');
    bks += bs.name; bks.push_back('\t');
    put_unsigned(bks, taxon); bks.push_back('\t');
    put_signed(bks, bs.l_seq()); bks.push_back('\t');
    append_counts(missing_count, 'M', bks);
    append_counts(ambig_count, 'A', bks);
    append_taxa_runs(taxon, taxa, bks);
}

void append_fastq_classification(const std::vector<tax_t> &taxa, tax_t taxon, u32 ambig_count, u32 missing_count,
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
    if (verbose) append_taxa_runs(taxon, taxa, bks); else bks.back() = '\n';
    const size_t cme = bks.size();
    bks += bs->seq; bks += "\n+\n"; bks += bs->qual.empty() ? bs->seq : bs->qual; bks.push_back('\n');
    if (is_paired) {
        const bseq1_t *m2 = bs + 1;
        bks += m2->name; bks.push_back(' ');
        bks.append(bks, cms, cme - cms); bks.push_back('\n');
        bks += m2->seq; bks += "\n+\n"; bks += m2->qual.empty() ? m2->seq : m2->qual; bks.push_back('\n');
    }
}

// ---------------------------------------------------------------------------------------------- classifier
ClassifierGeneric::ClassifierGeneric(const Database &db, const std::vector<u32> &parent, int device, int num_threads,
                                     bool emit_all, bool emit_fastq, bool emit_kraken, bool canonicalize, int layout)
    : k_(db.k_), nt_(num_threads > 0 ? num_threads : 1)
{
    if (emit_all) output_flag_ |= EMIT_ALL;
    if (emit_fastq) output_flag_ |= FASTQ;
    if (emit_kraken) output_flag_ |= KRAKEN;
    c_ = k_;
    for (u16 g : db.s_) c_ += g;
    chk(nullptr, bns_create(device, &ctx_), "bns_create");
    // bin/bonsai.cpp:152: Spacer(db.k_, wsz = db.k_, db.s_): classify looks up every k-mer.
    // A spaced seed takes the intended for_each_uncanon_spaced path (deviation from SURVEY F7, see README).
    chk(ctx_, bns_set_encoder(ctx_, db.k_, db.s_.empty() ? nullptr : db.s_.data(), canonicalize ? 1 : 0, 1), "bns_set_encoder");
    chk(ctx_, bns_load_table(ctx_, db.db_.n_buckets, db.db_.flags.data(), db.db_.keys.data(), db.db_.vals.data(), layout), "bns_load_table");
    chk(ctx_, bns_load_taxonomy(ctx_, parent.data(), (u32)parent.size()), "bns_load_taxonomy");
}

ClassifierGeneric::~ClassifierGeneric() { if (ctx_) bns_destroy(ctx_); }

void classify_seqs(ClassifierGeneric &c, bseq1_t *bs, std::string &cks, unsigned n, int is_paired)
{
    const unsigned inc = is_paired ? 2 : 1;
    n -= n % inc;
    if (!n) return;
    std::vector<u64> offsets(n + 1, 0);
    for (unsigned i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + bs[i].seq.size();
    std::string bases;
    bases.resize(offsets[n] + 8, 'N');
    const unsigned n_units = n / inc;
    const unsigned nt = (unsigned)std::max(1, std::min<int>(c.nt_, (int)(n_units / 4096 + 1)));
    auto parallel = [&](auto &&fn) {                        // static split of [0, n_units) over nt host threads (-p)
        if (nt == 1) { fn(0u, n_units, 0u); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] { fn((unsigned)((u64)n_units * t / nt), (unsigned)((u64)n_units * (t + 1) / nt), t); });
        for (auto &x : th) x.join();
    };
    parallel([&](unsigned lo, unsigned hi, unsigned) {
        for (unsigned i = lo * inc; i < hi * inc; ++i) std::memcpy(&bases[offsets[i]], bs[i].seq.data(), bs[i].seq.size());
    });
    std::vector<u32> taxon(n_units), missing(n_units), ambig(n_units), n_hits(n_units), hits;
    const bool want_runs = c.get_emit_kraken() != 0;          // run strings are only printed in Kraken / verbose FASTQ mode
    if (want_runs) hits.resize(offsets[n] + 1);
    // the one call that replaces the kt_forpool fan-out of classifier.h:275
    chk(c.ctx_, bns_classify_batch(c.ctx_, bases.data(), offsets.data(), n, is_paired, taxon.data(), missing.data(),
                                   ambig.data(), n_hits.data(), want_runs ? hits.data() : nullptr), "bns_classify_batch");
    std::vector<std::string> parts(nt);
    std::vector<u64> ncls(nt * 2, 0);
    parallel([&](unsigned lo, unsigned hi, unsigned t) {
        std::vector<tax_t> taxa;
        std::string &out = parts[t];
        for (unsigned u = lo; u < hi; ++u) {
            bseq1_t &b = bs[u * inc];
            ++ncls[t * 2 + (taxon[u] == 0)];
            if (!(c.get_emit_all() || taxon[u])) continue;
            taxa.clear();
            if (want_runs) taxa.assign(hits.begin() + offsets[u * inc], hits.begin() + offsets[u * inc] + n_hits[u]);
            if (c.get_emit_fastq())
                append_fastq_classification(taxa, taxon[u], ambig[u], missing[u], &b, out, c.get_emit_kraken(), is_paired);
            else if (c.get_emit_kraken())
                append_kraken_classification(taxa, taxon[u], ambig[u], missing[u], b, out);
        }
    });
    for (unsigned t = 0; t < nt; ++t) { cks += parts[t]; c.classified_[0] += ncls[t * 2]; c.classified_[1] += ncls[t * 2 + 1]; }
}

void process_dataset(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, unsigned chunk_size)
{
    SeqReader r1(fq1);
    std::unique_ptr<SeqReader> r2(fq2 ? new SeqReader(fq2) : nullptr);
    const int is_paired = fq2 != nullptr;
    const int fd = fileno(out);
    // Two-stage pipeline: a reader thread parses chunk i+1 (kseq semantics, single stream: gz inflate is the bound)
    // while this thread classifies chunk i on the GPU and formats it.
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::vector<bseq1_t>> queue;
    bool done = false;
    std::string reader_error;
    std::thread reader([&] {
        try {
            for (;;) {
                std::vector<bseq1_t> seqs;
                if (bseq_read((int)chunk_size, r1, r2.get(), seqs) <= 0) break;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return queue.size() < 2; });
                queue.push_back(std::move(seqs));
                cv.notify_all();
            }
        } catch (const std::exception &e) { reader_error = e.what(); }
        std::lock_guard<std::mutex> lk(mu);
        done = true;
        cv.notify_all();
    });
    auto flush = [&](std::string &cks) {
        std::fflush(out);
        for (size_t off = 0; off < cks.size();) {
            const ssize_t w = ::write(fd, cks.data() + off, cks.size() - off);
            if (w <= 0) die("write failed");
            off += (size_t)w;
        }
        cks.clear();
    };
    std::string cks;
    bool first = true;
    try {
        for (;;) {
            std::vector<bseq1_t> seqs;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !queue.empty() || done; });
                if (queue.empty()) break;
                seqs = std::move(queue.front());
                queue.pop_front();
                cv.notify_all();
            }
            classify_seqs(c, seqs.data(), cks, (unsigned)seqs.size(), is_paired);
            if (first) { std::fprintf(stderr, "nseq: %i\n", (int)seqs.size()); first = false; }
            if (cks.size() > (1ull << 16)) flush(cks);
        }
    } catch (...) {
        { std::lock_guard<std::mutex> lk(mu); queue.clear(); done = true; }
        cv.notify_all();
        // let the reader run off the end of its current chunk; it exits on its own
        reader.detach();
        throw;
    }
    reader.join();
    if (!reader_error.empty()) die(reader_error);
    if (first) std::fprintf(stderr, "Could not get any sequences from file, fyi.\n");
    flush(cks);
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
    }
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
Encoder::Encoder(unsigned k, const spvec_t &gaps, bool canonicalize, int device) : k_(k), canon_(canonicalize)
{
    bool spaced = false;
    for (u16 g : gaps) spaced |= g != 0;
    if (spaced) canon_ = false;                                 // encoder.h:148-150
    chk(nullptr, bns_create(device, &ctx_), "bns_create");
    // string for_each semantics of the reference, including SURVEY F7 for a spaced seed
    chk(ctx_, bns_set_encoder(ctx_, k, gaps.empty() ? nullptr : gaps.data(), canon_ ? 1 : 0, 0), "bns_set_encoder");
}

Encoder::~Encoder() { if (ctx_) bns_destroy(ctx_); }

void Encoder::fetch(const char *str, u64 l)
{
    const u64 offsets[2] = {0, l};
    kmers_.assign(l + 1, 0);
    u32 n = 0;
    chk(ctx_, bns_encode_batch(ctx_, str, offsets, 1, kmers_.data(), &n), "bns_encode_batch");
    kmers_.resize(n);
}

}  // namespace bns
