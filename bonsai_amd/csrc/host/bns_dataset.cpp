// bns_dataset.cpp -- process_dataset (classifier.h:296-337) and the pre-packed read container (host side of the classify path; see bns_host.hpp for the reference map).
#include "bns_host_internal.hpp"

namespace bns {
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
    if (!is_pack_container(fq1) && bgzf_gpu_wanted(c, fq1, fq2)) {
        if (process_bgzf_gpu(c, fq1, out, skip_units)) return;
    }
    // one plain gzip file: the stream entered at block headers found on the device, inflated, parsed and classified there (process_gz_gpu);
    // same rule for what it hands back -- and for a stream the device decoder does not take
    else if (!is_pack_container(fq1) && gz_gpu_wanted(c, fq1, fq2)) {
        if (process_gz_gpu(c, fq1, out, skip_units)) return;
    }
    // a pair of plain gzip files: both streams inflated on the device, mates paired there (process_gz_gpu_pair); same rule
    else if (!is_pack_container(fq1) && gz_pair_gpu_wanted(c, fq1, fq2)) {
        if (process_gz_gpu_pair(c, fq1, fq2, out, skip_units)) return;
    }
    // a pair of BGZF files: both inflated on the device and paired there (process_bgzf_gpu_pair); same rule for what it hands back
    if (!is_pack_container(fq1) && bgzf_pair_gpu_wanted(c, fq1, fq2)) {
        if (process_bgzf_gpu_pair(c, fq1, fq2, out, skip_units)) return;
    }
    // a pair of plain files: both as text on the device, mates by record index (process_text_gpu_pair); same rule for what it hands back
    else if (!is_pack_container(fq1) && pair_gpu_wanted(c, fq1, fq2)) {
        if (process_text_gpu_pair(c, fq1, fq2, out, skip_units)) return;
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
                if (job.seq == 0 && !c.nseq_printed_) std::fprintf(stderr, "nseq: %i\n", (int)job.res->n);
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


}  // namespace bns
