// bns_text_pipeline.cpp -- FASTA / FASTQ text parsed on the device: the pipelines that feed bns_classify_text from plain files, BGZF files and pairs of either (host side of the classify path; see bns_host.hpp for the reference map).
#include "bns_host_internal.hpp"

namespace bns {
// ---------------------------------------------------------------------------------------------- text on the device
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

// ---- finished blocks -> text, in block order (formatter threads taking alternate blocks, one writer that keeps the order): the one
// ordered writer under every device-text pipeline below
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
                if (j->seq == 0 && j->n_records) { std::fprintf(stderr, "nseq: %i\n", (int)j->n_records); c_.nseq_printed_ = true; }
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
    std::map<u64, std::unique_ptr<TextJob>> loading, loaded, done;
    u64 next_load = 0, next_verify = 0;
    u64 verified_end = 0;                                      // where the first record of block next_verify starts
    std::map<u64, u64> end_of;                                 // block -> where it stopped (as far as known)
    std::deque<std::unique_ptr<TextJob>> redo;                 // blocks whose guessed start was wrong
    bool cancel = false, stop_loading = false;
    u64 resume_at = fsize;                                     // the host parser's share starts here (fsize: nothing)
    std::string error;
    double t_read = 0, t_call = 0, t_alloc = 0;
    u64 n_guess = 0, n_redo = 0, n_ahead = 0;
    double t_idle = 0;                                         // callers waiting for a block to be read
    auto fail_with = [&](const std::string &w) { if (error.empty()) error = w; cancel = true; cv.notify_all(); };
    // ---- formatters and the writer (file order): verified blocks go to the sink under their block number
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { std::lock_guard<std::mutex> lk(mu); spare.push_back(std::move(j)); cv.notify_all(); });

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
            sink.submit(std::move(it->second));                // (its seq is its block number: the sink prints in that order)
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

    std::vector<std::thread> readers, callers;
    for (unsigned r = 0; r < R; ++r) readers.emplace_back(reader);
    for (unsigned g = 0; g < G; ++g) callers.emplace_back(caller, g);
    for (auto &t : callers) t.join();
    u64 n_final;
    {
        std::lock_guard<std::mutex> lk(mu);
        n_final = next_verify;                                 // every block up to here is verified (or the path has handed over there)
        stop_loading = true;
        cv.notify_all();
    }
    if (error.empty()) sink.finish(n_final); else sink.finish(0, true);
    { std::lock_guard<std::mutex> lk(mu); cancel = true; cv.notify_all(); }
    for (auto &t : readers) t.join();
    for (bns_ctx *cx : c.ctxs_) (void)bns_text_prefetch(cx, nullptr, nullptr, 0);      // (blocks uploaded ahead of a call that never came: handed over, or failed)
    if (!error.empty()) die(error);
    if (timing)
        std::fprintf(stderr, "[timing] text on the device: %llu blocks of %llu MiB on %u device(s), %u readers: page-lock %.3f s, pread %.3f (summed), calls %.3f (summed), format %.3f, write %.3f; "
                             "callers waited %.3f s for blocks, %llu uploads started ahead of their call; %llu guessed starts, %llu classified again%s\n",
                     (unsigned long long)next_verify, (unsigned long long)(B >> 20), G, R, t_alloc, t_read, t_call, sink.t_format, sink.t_write, t_idle, (unsigned long long)n_ahead,
                     (unsigned long long)n_guess, (unsigned long long)n_redo, resume_at != fsize ? "; the host parser takes the rest" : "");
    return resume_at;
}
bool bgzf_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    if (fq2 || c.get_emit_fastq()) return false;
    if (const char *e = std::getenv("BNS_TEXT_GPU")) if (e[0] == '0') return false;
    struct stat st;
    if (::stat(fq1, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    return is_bgzf_file(fq1) && !std::getenv("BNS_NO_BGZF");
}

// ---- blocks of one input over several devices, cut in order -------------------------------------------------------------------
// Block b goes to device b % G -- its bytes read, uploaded or inflated there ahead of time, side by side with the other devices' -- but
// where a block's first record starts (and, for a pair of files, which record of the other file is its mate) is only known when the
// block in front has been parsed.  So the blocks are CUT in order: the thread of block b waits for its turn, parses
// (bns_classify_text with BNS_TEXT_DEFER: the records and where the call stopped are known after ~0.15 ms per 64 MiB), hands the
// turn on with what block b + 1 has to know, and only then classifies (bns_text_finish) -- while the next device parses.  Nothing is
// guessed and nothing is classified twice; records, their order and the pairing are those of one device by construction
// (classifier.h:296-337 reads its chunks in order, too).
class Turns {
public:
    // block b's turn (false: the chain has stopped -- text handed back to the host parser, or a failure)
    bool wait(u64 b) { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return stop_ || turn_ == b; }); return !stop_; }
    void pass() { std::lock_guard<std::mutex> lk(mu_); ++turn_; cv_.notify_all(); }
    void halt() { std::lock_guard<std::mutex> lk(mu_); stop_ = true; cv_.notify_all(); }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    u64 turn_ = 0;
    bool stop_ = false;
};

// recycled result buffers (their page-locked arrays with them)
class JobPool {
public:
    std::unique_ptr<TextJob> get()
    {
        { std::lock_guard<std::mutex> lk(mu_); if (!spare_.empty()) { auto j = std::move(spare_.back()); spare_.pop_back(); return j; } }
        return std::make_unique<TextJob>();
    }
    void put(std::unique_ptr<TextJob> j) { std::lock_guard<std::mutex> lk(mu_); spare_.push_back(std::move(j)); }
private:
    std::mutex mu_;
    std::vector<std::unique_ptr<TextJob>> spare_;
};

// the result arrays of one bns_classify_text call, sized for `cap` records (names_cap / runs_cap bytes / runs)
void size_text_job(bns_ctx *ctx, TextJob &j, bns_text_out &o, bool taxon_only, u64 cap, u64 names_cap, u64 runs_cap)
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

// The library calls on ONE block's text.  As a rule one call in two halves: parse() under the turn, finish() behind it.  A call that
// stops at BNS_TEXT_CAP (records of a few bytes, long names: the arrays are sized for ~160 bytes of text per record) is finished at once,
// what it took printed as a job of its own, and the next call goes on from there ON THE SAME TEXT with arrays twice the size -- only the
// truly unfinished last record is left for the block behind.
struct BlockCalls {
    ClassifierGeneric &c;
    bns_ctx *ctx;
    TextSink &sink;
    JobPool &pool;
    u64 &n_jobs;                                               // the chain's job counter (touched under the turn only)
    int n_streams = 1;
    const char *tp[2] = {nullptr, nullptr};                    // the block's text (host or device) and its size
    u64 tb[2] = {0, 0};
    u64 limit = ~0ULL;                                         // stream 0: records that start in front of this offset only
    int flags = 0;                                             // BNS_TEXT_DEVICE / BNS_TEXT_FINAL / BNS_TEXT_TRIM_READNO
    // results
    u64 used[2] = {0, 0};                                      // consumed, all calls together
    int status = BNS_TEXT_OK;                                  // of the last call
    u64 units = 0;                                             // units handed to the sink
    double ms_parse = 0, ms_classify = 0;

    BlockCalls(ClassifierGeneric &c_, bns_ctx *ctx_, TextSink &sink_, JobPool &pool_, u64 &n_jobs_) : c(c_), ctx(ctx_), sink(sink_), pool(pool_), n_jobs(n_jobs_) {}

    void parse()
    {
        taxon_only_ = !c.get_emit_kraken();
        cap_ = (tb[0] + tb[1]) / 160 + 4096; names_cap_ = cap_ * 24; runs_cap_ = cap_ * 4;
        for (;;) {
            if (limit != ~0ULL && used[0] >= limit) { status = BNS_TEXT_OK; pending_ = false; return; }     // (everything in front of the limit went with the calls so far)
            j_ = pool.get();
            bns_text_out o{};
            size_text_job(ctx, *j_, o, taxon_only_, cap_, names_cap_, runs_cap_);
            for (int s = 0; s < n_streams; ++s) { cp_[s] = tp[s] + used[s]; cb_[s] = tb[s] - used[s]; }
            lim_ = limit == ~0ULL ? ~0ULL : limit - used[0];
            chk(ctx, bns_classify_text(ctx, cp_, cb_, n_streams, lim_, flags | BNS_TEXT_DEFER, cap_, &o, &first_), "bns_classify_text");
            ms_parse += first_.ms_parse;
            if (first_.status != BNS_TEXT_CAP) break;
            // the arrays are full: this call is finished here (under the turn), the next one goes on behind it
            bns_text_info fin{};
            chk(ctx, bns_text_finish(ctx, &fin), "bns_text_finish");
            ms_classify += fin.ms_classify;
            for (int s = 0; s < n_streams; ++s) used[s] += fin.consumed[s];
            if (fin.n_records) submit(fin.n_records); else pool.put(std::move(j_));
            cap_ *= 2; names_cap_ *= 2; runs_cap_ *= 2;
        }
        // the block's last call: its second half waits.  (its job's number is taken now: the jobs are printed in this order)
        for (int s = 0; s < n_streams; ++s) used[s] += first_.consumed[s];
        status = first_.status;
        j_->seq = n_jobs++;
        pending_ = true;
    }

    void finish()
    {
        if (!pending_) return;
        pending_ = false;
        bns_text_info fin{};
        chk(ctx, bns_text_finish(ctx, &fin), "bns_text_finish");
        ms_classify += fin.ms_classify;
        while (fin.n_records != first_.n_records) {
            // the hit runs did not fit the job's arrays (the first half cannot know how many there will be): the same call once more, in
            // one piece, with room -- the text is still where it was, the records and where the call stops are the same
            if (fin.status != BNS_TEXT_CAP) die("bns_text_finish: fewer records than the first half of the call reported");
            runs_cap_ *= 2;
            bns_text_out o{};
            size_text_job(ctx, *j_, o, taxon_only_, cap_, names_cap_, runs_cap_);
            chk(ctx, bns_classify_text(ctx, cp_, cb_, n_streams, lim_, flags, cap_, &o, &fin), "bns_classify_text");
            ms_classify += fin.ms_classify;
            if (fin.n_records == first_.n_records && (fin.consumed[0] != first_.consumed[0] || fin.consumed[1] != first_.consumed[1]))
                die("bns_classify_text: the same text parsed differently the second time");
        }
        const u64 seq = j_->seq;
        submit(fin.n_records, &seq);
    }
    bool has_pending() const { return pending_; }

private:
    void submit(u64 n_records, const u64 *seq = nullptr)
    {
        j_->seq = seq ? *seq : n_jobs++;
        j_->n_records = n_records; j_->mates = (unsigned)n_streams;
        units += n_records / (u64)n_streams;
        sink.submit(std::move(j_));
    }
    std::unique_ptr<TextJob> j_;
    bns_text_info first_{};
    const char *cp_[2] = {nullptr, nullptr};
    u64 cb_[2] = {0, 0}, lim_ = ~0ULL;
    u64 cap_ = 0, names_cap_ = 0, runs_cap_ = 0;
    bool taxon_only_ = false, pending_ = false;
};

// A BGZF file as text in DEVICE memory, batch by batch in file order: compressed members up (pread into page-locked memory,
// bns_inflate_members_device: one member per wavefront, thousands per batch, two batches side by side on inflater handles of their
// own), their text left in HBM behind HEAD bytes of room (for what the caller could not finish of the batch in front: the record that
// straddles two batches).  Batch b is inflated on device b % G (round 6: the members of a BGZF file are independent, so every device
// inflates its own batches into its own memory; one set of readers and one header walk feed them all).
class BgzfDeviceSource {
public:
    struct Item { u64 seq = 0; int tbuf = -1; u64 text_bytes = 0; bool last = false; };
    u64 HEAD = 0, TEXT_MAX = 0;
    unsigned R = 0, NI = 0, G = 1, n_handles = 0;
    // (what the timing line prints)
    double t_read = 0, t_inflate = 0, t_kernel = 0, t_split = 0, t_pin = 0, t_wait_inf = 0, t_wait_next = 0, t_wait_walk = 0, t_first_inflated = 0;
    u64 n_members = 0, text_total = 0;

    // range_scale: the ranges of compressed bytes (= batches) of THIS file against the default size (a pair of files: the second file's
    // ranges scaled by the files' sizes, so that batch b of either file holds about the same records)
    BgzfDeviceSource(ClassifierGeneric &c, const char *path, double range_scale = 1.0)
    {
        G = (unsigned)c.ctxs_.size();
        dev_.resize(G);
        for (unsigned g = 0; g < G; ++g) { dev_[g].ctx = c.ctxs_[g]; dev_[g].device = c.devices_[g]; dev_[g].next_inflate = dev_[g].next_out = g; }
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) die(std::string("Could not open ") + path + " for reading.");
        fsize_ = (u64)::lseek(fd_, 0, SEEK_END);
        auto env_num = [](const char *name, u64 dflt) { const char *e = std::getenv(name); return e && std::atol(e) > 0 ? (u64)std::atol(e) : dflt; };
        MEMB_ = std::min<u64>(env_num("BNS_BGZF_BATCH_MEMBERS", 16384), 1u << 20);   // members per batch
        HEAD = std::min<u64>(env_num("BNS_BGZF_HEAD_MB", 64) << 20, 256ull << 20);   // room in front of a batch's text for what the batch before left
        if (const char *e = std::getenv("BNS_BGZF_HEAD_BYTES")) HEAD = (u64)std::max(4096L, std::min(256L << 20, std::atol(e)));       // (tests: windows of a few records)
        TEXT_MAX = std::min<u64>(MEMB_ * 65536ull, (2047ull << 20) - HEAD);           // (a call takes less than 2^31 bytes of text, what the batch in front left included)
        NI = (unsigned)std::max<u64>(1, std::min<u64>(8, env_num("BNS_BGZF_GPU_THREADS", 2)));
        // (inflater handles are a DEVICE's: two are what keeps one busy -- 4 k members in flight --, so contexts that share a device share them;
        // with `-g 0,0` four handles ran four inflate kernels beside each other and page-locked twice the slots for nothing)
        n_handles = 0;
        for (unsigned g = 0; g < G; ++g) {
            unsigned same = 0;
            for (unsigned q = 0; q < G; ++q) same += dev_[q].device == dev_[g].device ? 1u : 0u;
            dev_[g].ni = std::max(1u, NI / same);
            n_handles += dev_[g].ni;
        }
        R = (unsigned)std::max(2, std::min<int>(6 + 2 * ((int)G - 1), usable_cpus() / 3));
        // The file is read in RANGES of CB compressed bytes at nominal offsets (plus one member's worth of slack), side by side and ahead;
        // a walker goes over the ranges in file order and finds the members in the bytes that were just read -- no page of a mapping
        // is touched (walking the headers over a mapping was a page fault per member: 1.8-2.5 s per 460 k members, the longest stage).
        // A batch = the members that START in a range (the one that straddles its end included: hence the slack).
        // CB: 96 MiB = ~3 k members a batch.  The member-per-wavefront inflate kernel is at its rate from ~4 k members in flight (two
        // handles work side by side), and a slot is page-locked before its first use, 0.45 ms per MiB with the other threads' HIP calls
        // waiting behind it: with 384 MiB ranges (what the member-per-lane kernel wanted) the GPU stood idle for the first 0.3 s of a
        // file (profiles/r05_bgzf_trace.txt: 64 M reads 1.35 s with 384 MiB, 0.92 with 128, 0.88 with 96 and with 64, 1.04 with 48).
        const u64 CB = std::max<u64>(1u << 20, (u64)((double)(env_num("BNS_BGZF_RANGE_MB", 96) << 20) * range_scale));
        // (the FIRST range is short: a slot is page-locked before it is read -- 0.45 ms per MiB -- and nothing is inflated until the first one
        // is; one short range only: every size step re-allocates the inflaters' device buffers and the result arrays, a drained device each)
        range_off_.push_back(0);
        for (u64 ramp : {CB / 12}) if (ramp >= (1u << 20) && range_off_.back() + ramp < fsize_) range_off_.push_back(range_off_.back() + ramp);
        while (range_off_.back() + CB < fsize_) range_off_.push_back(range_off_.back() + CB);
        range_off_.push_back(std::max<u64>(fsize_, range_off_.back()));
        n_ranges_ = range_off_.size() - 1;
        NS_ = n_handles + 3;
        // device text buffers, HEAD + TEXT_MAX each, per device: one per inflater, one inflated and waiting, and two with the callers (a
        // batch's buffer is let go when the batch behind it has taken what was left AND its own classify call is through)
        try {
            for (Dev &d : dev_) {
                d.tbufs.assign(d.ni + 3, nullptr);
                for (auto &p : d.tbufs) chk(d.ctx, bns_dev_alloc(d.ctx, (size_t)(HEAD + TEXT_MAX) + 4096, &p), "bns_dev_alloc");
                for (unsigned i = 0; i < d.tbufs.size(); ++i) d.free_t.push_back((int)i);
                // (the handles are made HERE, before a reader page-locks its first slot: a stream created behind five hipHostMallocs waited 0.3 s)
                d.handles.assign(d.ni, nullptr);
                for (auto &h : d.handles) if (bns_inflater_create(d.device, &h) != BNS_OK) die("BGZF input: could not open an inflater on the GPU");
            }
        } catch (...) { free_all(); throw; }
        t_begin_ = tnow();
        splitter_ = std::thread([this] { split_loop(); });
        for (unsigned r = 0; r < R; ++r) readers_.emplace_back([this] { read_loop(); });
        for (unsigned g = 0; g < G; ++g) for (unsigned i = 0; i < dev_[g].ni; ++i) inflaters_.emplace_back([this, g, i] { inflate_loop(g, dev_[g].handles[i]); });
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

    // device g's next batch (batches g, g + G, ...) in file order; false: there is none (the file is done, cancel() was called, or a
    // thread failed: error())
    bool next(unsigned g, Item &it)
    {
        Dev &d = dev_[g];
        std::unique_lock<std::mutex> lk(mu_);
        const double tw = tnow();
        cv_.wait(lk, [&] { return cancel_ || inflated_.count(d.next_out) || d.next_out >= n_batches_; });
        t_wait_next += tnow() - tw;
        if (d.next_out == 0) t_first_inflated = tnow() - t_begin_;
        if (cancel_ || !inflated_.count(d.next_out)) return false;
        std::unique_ptr<Batch> b = std::move(inflated_[d.next_out]); inflated_.erase(d.next_out);
        it.seq = d.next_out; it.tbuf = b->tbuf; it.text_bytes = b->text_bytes; it.last = b->last;
        d.next_out += G;
        return true;
    }
    // true once the walker knows that the file has no batch `seq`
    bool no_batch(u64 seq) { std::lock_guard<std::mutex> lk(mu_); return seq >= n_batches_; }
    // a free text buffer of device g (its HEAD room: for a side of a pair that has no batch of its own left); -1: cancelled
    int acquire(unsigned g)
    {
        Dev &d = dev_[g];
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return cancel_ || !d.free_t.empty(); });
        if (cancel_) return -1;
        const int t = d.free_t.back(); d.free_t.pop_back();
        return t;
    }
    char *buf(unsigned g, int t) const { return static_cast<char *>(dev_[g].tbufs[(size_t)t]); }
    bns_ctx *ctx(unsigned g) const { return dev_[g].ctx; }
    void release(unsigned g, int t) { std::lock_guard<std::mutex> lk(mu_); dev_[g].free_t.push_back(t); cv_.notify_all(); }
    void cancel() { std::lock_guard<std::mutex> lk(mu_); cancel_ = true; cv_.notify_all(); }
    std::string error() { std::lock_guard<std::mutex> lk(mu_); return error_; }

private:
    struct Dev {
        bns_ctx *ctx = nullptr;
        int device = 0;
        std::vector<void *> tbufs;
        std::vector<int> free_t;
        std::vector<bns_inflater *> handles;
        u64 next_inflate = 0, next_out = 0;
        unsigned ni = 1;                                        // inflater handles of this context
    };
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
        for (Dev &d : dev_) {
            for (bns_inflater *h : d.handles) if (h) bns_inflater_destroy(h);
            d.handles.clear();
            for (void *p : d.tbufs) if (p) bns_dev_free(d.ctx, p);
            d.tbufs.clear();
        }
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
                            // (page-locked, portable: whichever device inflates it.  Registered memory of our own: hipHostMalloc was 0.45 ms per
                            // MiB with every other thread's HIP calls waiting behind it -- the GPU idled through most of a file's first 0.15 s while
                            // five slots were made; BNS_PIN_MALLOC=1: as before -- PinnedBuf::reserve)
                            sl->comp.reserve(dev_[sl->seq % G].ctx, sl->bytes + 256);
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
    // ---- inflaters: a handle each; device g's batches in file order, each into a free text buffer of that device (behind HEAD bytes of room)
    void inflate_loop(unsigned g, bns_inflater *h)
    {
        Dev &d = dev_[g];
        try {
            for (;;) {
                std::unique_ptr<Batch> b;
                int tb = -1;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    const double tw = tnow();
                    cv_.wait(lk, [&] { return cancel_ || (loaded_.count(d.next_inflate) && !d.free_t.empty()) || d.next_inflate >= n_batches_; });
                    t_wait_inf += tnow() - tw;
                    if (cancel_ || !loaded_.count(d.next_inflate)) break;
                    b = std::move(loaded_[d.next_inflate]); loaded_.erase(d.next_inflate); d.next_inflate += G;
                    tb = d.free_t.back(); d.free_t.pop_back();
                }
                const size_t n = b->in_off.size();
                b->crc.assign(n, 0); b->status.assign(n, 0);
                const double t0 = tnow();
                if (n) {
                    const int rc = bns_inflate_members_device(h, reinterpret_cast<const uint8_t *>(b->slot->comp.p), b->slot->bytes, b->in_off.data(), b->in_len.data(),
                                                              b->out_off.data(), b->out_len.data(), n, static_cast<char *>(d.tbufs[(size_t)tb]) + HEAD, b->text_bytes,
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

    std::vector<Dev> dev_;
    int fd_ = -1;
    u64 fsize_ = 0, MEMB_ = 0, n_ranges_ = 0;
    unsigned NS_ = 0;
    std::vector<u64> range_off_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Slot *> spare_, all_slots_;                    // (slots go back to spare_ when the last batch that points into them lets go)
    std::deque<Piece> pieces_;
    std::map<u64, std::shared_ptr<Slot>> reading_, read_done_;
    std::map<u64, std::unique_ptr<Batch>> loaded_, inflated_;
    u64 next_range_ = 0, n_batches_ = ~0ULL;
    bool cancel_ = false;
    std::string error_;
    double t_begin_ = 0;
    std::thread splitter_;
    std::vector<std::thread> readers_, inflaters_;
};

// A batch's text buffer is let go when both are done with it: the block behind it (has taken the unfinished rest) and the batch's own
// classify call (which reads the text again when the hit runs did not fit the first time).
struct TextHold {
    BgzfDeviceSource *src = nullptr;
    unsigned dev = 0;
    int tbuf = -1;
    std::atomic<int> left{2};
    void drop() { if (left.fetch_sub(1) == 1 && src && tbuf >= 0) src->release(dev, tbuf); }
};

// One side of the chain's hand-over for device text: what the block in front left unfinished, in ITS device buffer
struct Rest {
    std::shared_ptr<TextHold> hold;
    u64 off = 0, len = 0;
};
// ... copied into the room in front of the next block's text (src.buf(g, tbuf) + HEAD - len): one device or two
static void take_rest(BgzfDeviceSource &src, unsigned g, int tbuf, Rest &rest)
{
    if (rest.len)
        chk(src.ctx(g), bns_dev_copy_peer(src.ctx(g), src.buf(g, tbuf) + src.HEAD - rest.len, src.ctx(rest.hold->dev), src.buf(rest.hold->dev, rest.hold->tbuf) + rest.off, (size_t)rest.len),
            "bns_dev_copy_peer");
    if (rest.hold) { rest.hold->drop(); rest.hold.reset(); }
}

// A BGZF file whose text never leaves the devices: BgzfDeviceSource's batches (batch b inflated on device b % G), what the batch in front
// could not finish copied in front of the next one's text, bns_classify_text on it where it lies -- cut in file order (Turns), classified
// side by side --, names and results down.
// -> true: the whole file was classified.  false: the kernels handed text back (not in their regular form) after `units_done`
// units had been printed: the caller reads the file with the host parser and leaves those out.
bool process_bgzf_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    const unsigned G = (unsigned)c.ctxs_.size();
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) for (bns_ctx *cx : c.ctxs_) (void)bns_set_timing(cx, 1);      // (HIP events around the parse and classify kernels: the sums in the timing line)
    JobPool pool;
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { pool.put(std::move(j)); });
    BgzfDeviceSource src(c, fq1);
    const u64 HEAD = src.HEAD;
    Turns turns;
    // what the turn carries from block to block
    Rest rest;
    u64 n_jobs = 0;
    std::mutex mu;                                             // (the sums below, and the first failure)
    double t_gpu_parse = 0, t_gpu_cls = 0, t_call = 0;
    u64 units = 0;
    bool handed_back = false;
    std::string failure;

    auto worker = [&](unsigned g) {
        try {
            bns_ctx *ctx = c.ctxs_[g];
            for (u64 b = g;; b += G) {
                BgzfDeviceSource::Item it;
                if (!src.next(g, it)) { if (!src.no_batch(b)) turns.halt(); break; }       // (the file is done -- or the source has stopped: nobody waits for this block's turn)
                auto hold = std::make_shared<TextHold>();
                hold->src = &src; hold->dev = g; hold->tbuf = it.tbuf;
                if (!turns.wait(b)) { src.release(g, it.tbuf); break; }
                // ---- this block's turn
                const double t0 = tnow();
                if (rest.len > HEAD) {                              // (a record longer than HEAD: the host parser's)
                    std::lock_guard<std::mutex> lk(mu);
                    handed_back = true; turns.halt(); src.release(g, it.tbuf);
                    break;
                }
                const u64 tail_len = rest.len;
                take_rest(src, g, it.tbuf, rest);
                BlockCalls calls(c, ctx, sink, pool, n_jobs);
                calls.tp[0] = src.buf(g, it.tbuf) + HEAD - tail_len;
                calls.tb[0] = tail_len + it.text_bytes;
                calls.flags = BNS_TEXT_DEVICE | BNS_TEXT_TRIM_READNO | (it.last ? BNS_TEXT_FINAL : 0);
                calls.parse();
                // (a batch without one complete record is not an error as long as more text follows: all of it waits in front of the next one)
                const bool ok = (calls.status == BNS_TEXT_OK || (calls.status == BNS_TEXT_NO_RECORD && !it.last)) && (!it.last || calls.used[0] == calls.tb[0]);
                rest.hold = hold; rest.off = (HEAD - tail_len) + calls.used[0]; rest.len = calls.tb[0] - calls.used[0];
                if (ok) turns.pass();
                else { std::lock_guard<std::mutex> lk(mu); handed_back = true; turns.halt(); }
                // ---- behind the turn: classify, results down, the job to the formatters
                calls.finish();
                hold->drop();
                std::lock_guard<std::mutex> lk(mu);
                t_call += tnow() - t0; t_gpu_parse += calls.ms_parse * 1e-3; t_gpu_cls += calls.ms_classify * 1e-3;
                units += calls.units;
                if (!ok) break;
            }
        } catch (const std::exception &e) {
            { std::lock_guard<std::mutex> lk(mu); if (failure.empty()) failure = e.what(); }
            turns.halt(); src.cancel();
        }
    };
    std::vector<std::thread> th;
    for (unsigned g = 1; g < G; ++g) th.emplace_back(worker, g);
    worker(0);
    for (auto &t : th) t.join();
    src.stop();
    if (failure.empty()) failure = src.error();
    if (!failure.empty()) { sink.finish(0, true); die(failure); }
    sink.finish(n_jobs);
    units_done = units;
    if (timing)
        std::fprintf(stderr, "[timing] BGZF text on the device: %llu jobs on %u device(s), %llu members, %.2f GB of text; header walk %.3f s, pread %.3f (summed over %u readers), inflate calls %.3f (summed over %u handles) of which kernel %.3f, "
                             "classify calls %.3f (their kernels: text %.3f, classify %.3f), format %.3f, write %.3f; page-lock %.3f (summed), first batch inflated after %.3f s, waits: walker for bytes %.3f, inflaters for batches / buffers %.3f (summed), classify for text %.3f%s\n",
                     (unsigned long long)n_jobs, G, (unsigned long long)src.n_members, src.text_total / 1e9, src.t_split, src.t_read, src.R, src.t_inflate, src.n_handles, src.t_kernel, t_call, t_gpu_parse, t_gpu_cls,
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

// A PAIR of BGZF files on SEVERAL devices: batch b of either file is inflated on device b % G, the second file's ranges scaled by the
// files' sizes so that batch b of either holds about the same records (both files hold the same number; what process_text_gpu_pair does
// with its blocks).  Call b = what call b - 1 left of either file + batch b of either, mates paired record for record on the device;
// cut in order (Turns), classified side by side.  A side whose batches have run out goes on with what is left of it (in a buffer taken
// from its source on the call's device).  What one side runs ahead of the other stays in front of its next batch: when that is more than
// HEAD bytes (files whose record sizes drift apart within the files, not just between them) this path stops and the host parser takes over.
static bool process_bgzf_gpu_pair_multi(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    const unsigned G = (unsigned)c.ctxs_.size();
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) for (bns_ctx *cx : c.ctxs_) (void)bns_set_timing(cx, 1);
    struct stat st0, st1;
    if (::stat(fq1, &st0) != 0 || ::stat(fq2, &st1) != 0) die("Could not stat the input files.");
    JobPool pool;
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { pool.put(std::move(j)); });
    BgzfDeviceSource src0(c, fq1), src1(c, fq2, (double)std::max<off_t>(1, st1.st_size) / (double)std::max<off_t>(1, st0.st_size));
    BgzfDeviceSource *srcs[2] = {&src0, &src1};
    const u64 HEAD = src0.HEAD;
    Turns turns;
    Rest rest[2];
    bool exhausted[2] = {false, false};                        // (under the turn) the side's last batch has been taken
    u64 n_jobs = 0;
    std::mutex mu;
    double t_gpu_parse = 0, t_gpu_cls = 0, t_call = 0;
    u64 units = 0;
    bool handed_back = false, done = false;
    std::string failure;

    auto worker = [&](unsigned g) {
        try {
            bns_ctx *ctx = c.ctxs_[g];
            for (u64 b = g;; b += G) {
                // this call's batches, inflated on this device ahead of the turn (a side without a batch b: its file is done)
                BgzfDeviceSource::Item it[2];
                bool have[2];
                for (int s = 0; s < 2; ++s) {
                    have[s] = srcs[s]->next(g, it[s]);
                    if (!have[s] && !srcs[s]->no_batch(b)) { turns.halt(); return; }
                }
                if (!have[0] && !have[1]) break;                   // (both files ended in front of batch b: the call that took the later of their last batches was the final one)
                if (!turns.wait(b)) { for (int s = 0; s < 2; ++s) if (have[s]) srcs[s]->release(g, it[s].tbuf); break; }
                // ---- this call's turn
                const double t0 = tnow();
                if (done) { for (int s = 0; s < 2; ++s) if (have[s]) srcs[s]->release(g, it[s].tbuf); turns.halt(); break; }
                std::shared_ptr<TextHold> hold[2];
                BlockCalls calls(c, ctx, sink, pool, n_jobs);
                calls.n_streams = 2;
                bool too_long = false;
                for (int s = 0; s < 2; ++s) {
                    if (rest[s].len > HEAD) too_long = true;
                    if (!have[s]) { it[s].tbuf = srcs[s]->acquire(g); it[s].text_bytes = 0; it[s].last = true; if (it[s].tbuf < 0) { turns.halt(); return; } }
                    hold[s] = std::make_shared<TextHold>();
                    hold[s]->src = srcs[s]; hold[s]->dev = g; hold[s]->tbuf = it[s].tbuf;
                }
                if (too_long) {
                    std::lock_guard<std::mutex> lk(mu);
                    handed_back = true; turns.halt();
                    for (int s = 0; s < 2; ++s) srcs[s]->release(g, it[s].tbuf);
                    break;
                }
                for (int s = 0; s < 2; ++s) {
                    const u64 tail_len = rest[s].len;
                    take_rest(*srcs[s], g, it[s].tbuf, rest[s]);
                    calls.tp[s] = srcs[s]->buf(g, it[s].tbuf) + HEAD - tail_len;
                    calls.tb[s] = tail_len + it[s].text_bytes;
                    if (it[s].last) exhausted[s] = true;
                    rest[s].off = HEAD - tail_len;                  // (+ what the call uses, below)
                }
                const bool final_call = exhausted[0] && exhausted[1];
                calls.flags = BNS_TEXT_DEVICE | BNS_TEXT_TRIM_READNO | (final_call ? BNS_TEXT_FINAL : 0);
                calls.parse();
                for (int s = 0; s < 2; ++s) { rest[s].hold = hold[s]; rest[s].off += calls.used[s]; rest[s].len = calls.tb[s] - calls.used[s]; }
                bool ok = calls.status == BNS_TEXT_OK || (calls.status == BNS_TEXT_NO_RECORD && !final_call);
                if (ok && final_call) {
                    done = true;
                    if (rest[0].len || rest[1].len)                 // kseq_declare.h:116-120 / 134-137: one file holds more records than the other
                        std::fprintf(stderr, "[W::%s] the %s file has fewer sequences.\n", "bseq_read", rest[0].len ? "2nd" : "1st");
                }
                if (ok) turns.pass();
                else { std::lock_guard<std::mutex> lk(mu); handed_back = true; turns.halt(); }
                // ---- behind the turn
                calls.finish();
                for (int s = 0; s < 2; ++s) hold[s]->drop();
                std::lock_guard<std::mutex> lk(mu);
                t_call += tnow() - t0; t_gpu_parse += calls.ms_parse * 1e-3; t_gpu_cls += calls.ms_classify * 1e-3;
                units += calls.units;
                if (!ok || done) break;
            }
        } catch (const std::exception &e) {
            { std::lock_guard<std::mutex> lk(mu); if (failure.empty()) failure = e.what(); }
            turns.halt(); src0.cancel(); src1.cancel();
        }
    };
    std::vector<std::thread> th;
    for (unsigned g = 1; g < G; ++g) th.emplace_back(worker, g);
    worker(0);
    for (auto &t : th) t.join();
    src0.stop(); src1.stop();
    if (failure.empty()) failure = src0.error();
    if (failure.empty()) failure = src1.error();
    if (!failure.empty()) { sink.finish(0, true); die(failure); }
    sink.finish(n_jobs);
    units_done = units;
    if (timing)
        std::fprintf(stderr, "[timing] pair of BGZF files, text on the device: %llu calls on %u devices, %llu + %llu members, %.2f + %.2f GB of text; pread %.3f s (summed), inflate calls %.3f of which kernel %.3f (summed over %u handles), "
                             "classify calls %.3f (their kernels: text %.3f, classify %.3f), format %.3f, write %.3f; first batches inflated after %.3f / %.3f s, classify waited %.3f s for text%s\n",
                     (unsigned long long)n_jobs, G, (unsigned long long)src0.n_members, (unsigned long long)src1.n_members, src0.text_total / 1e9, src1.text_total / 1e9, src0.t_read + src1.t_read,
                     src0.t_inflate + src1.t_inflate, src0.t_kernel + src1.t_kernel, src0.n_handles + src1.n_handles, t_call, t_gpu_parse, t_gpu_cls, sink.t_format, sink.t_write,
                     src0.t_first_inflated, src1.t_first_inflated, src0.t_wait_next + src1.t_wait_next, handed_back ? "; the host parser takes the rest" : "");
    return !handed_back;
}

// A PAIR of BGZF files, both inflated into device memory (a BgzfDeviceSource each) and paired there: bns_classify_text with two streams
// of device text -- record i of the one file and record i of the other are mates (kseq_declare.h:116-131).  The two files' batches do
// not end at the same record, so each side keeps a WINDOW: what its last call left, with the next batch behind it (the rest copied
// into the room in front of the new batch's text, device to device) whenever less than LOW bytes are left; a call takes the pairs
// both windows hold and says where it stopped in either.  One device: calls in file order (several: process_bgzf_gpu_pair_multi).
// -> true: everything was classified; false: text handed back after `units_done` pairs (the host parser reads both files and leaves
// those out)
bool process_bgzf_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done)
{
    if (c.ctxs_.size() > 1) return process_bgzf_gpu_pair_multi(c, fq1, fq2, out, units_done);
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    bns_ctx *ctx = c.ctxs_[0];
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) (void)bns_set_timing(ctx, 1);
    const bool want_runs = c.get_emit_kraken() != 0, taxon_only = !want_runs;
    JobPool pool;
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { pool.put(std::move(j)); });
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
                    if (!d.src->next(0, it)) {
                        const std::string e = d.src->error();
                        if (!e.empty()) die(e);
                        d.exhausted = true;
                        break;
                    }
                    char *base = d.src->buf(0, it.tbuf);
                    if (d.len) chk(ctx, bns_dev_copy(ctx, base + HEAD - d.len, d.src->buf(0, d.t) + d.off, (size_t)d.len), "bns_dev_copy");
                    if (d.t >= 0) d.src->release(0, d.t);
                    d.t = it.tbuf; d.off = HEAD - d.len; d.len += it.text_bytes;
                    if (it.last) d.exhausted = true;
                }
            }
            const bool final_call = side[0].exhausted && side[1].exhausted;
            if (final_call && side[0].len == 0 && side[1].len == 0) break;
            std::unique_ptr<TextJob> j = pool.get();
            const double t0 = tnow();
            const char *tp[2] = {side[0].t >= 0 ? side[0].src->buf(0, side[0].t) + side[0].off : nullptr, side[1].t >= 0 ? side[1].src->buf(0, side[1].t) + side[1].off : nullptr};
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
                     src0.t_inflate + src1.t_inflate, src0.t_kernel + src1.t_kernel, src0.n_handles + src1.n_handles, t_call, t_gpu_parse, t_gpu_cls, sink.t_format, sink.t_write,
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
// says where it stopped in both.  Block b goes to device b % G (its buffers page-locked for it, its upload started ahead:
// bns_text_prefetch); the blocks are cut in order (Turns: a block's turn is its parse) and classified side by side.  Where file 2 drifts
// out of its buffer (mates whose sizes differ more in one stretch of the files than the room allows), or the kernels hand text back, this
// path stops: the caller reads both files with the host parser and leaves out the units that were printed.  -> true: everything was classified
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
    const unsigned G = (unsigned)c.ctxs_.size();
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
    const unsigned max_jobs = 3 * G + 2;                        // per device: in its call, uploaded ahead of it, being read
    struct Piece { PairJob *j; int s; size_t off, len; };
    std::deque<Piece> pieces;
    std::map<u64, std::unique_ptr<PairJob>> loading, loaded;
    u64 next_load = 0;
    bool cancel = false;
    std::string error;
    double t_read = 0, t_call = 0, t_idle = 0;
    u64 n_ahead = 0;
    auto fail_with = [&](const std::string &w) { if (error.empty()) error = w; cancel = true; cv.notify_all(); };
    JobPool pool;
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { pool.put(std::move(j)); });

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
                            bns_ctx *cx = c.ctxs_[b % G];
                            jp->text[0].reserve(cx, (size_t)std::max<u64>(B + SLACK, jp->bytes[0]) + 256);
                            jp->text[1].reserve(cx, (size_t)std::max<u64>(B2 + 2 * ROOM, jp->bytes[1]) + 256);
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

    Turns turns;
    // what the turn carries from block to block
    u64 pos[2] = {0, 0};                                       // where the next call starts in either file
    u64 n_jobs = 0;
    bool handed_back = false;
    u64 units = 0;

    auto worker = [&](unsigned g) {
        try {
            bns_ctx *ctx = c.ctxs_[g];
            for (u64 b = g; b < n_blocks; b += G) {
                std::unique_ptr<PairJob> j;
                PairJob *ahead = nullptr;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    const double tw = tnow();
                    cv.wait(lk, [&] { return cancel || loaded.count(b); });
                    t_idle += tnow() - tw;
                    if (cancel) { turns.halt(); break; }
                    j = std::move(loaded[b]); loaded.erase(b);
                    auto nx = loaded.find(b + G);                    // this device's next block: its upload may start now
                    if (nx != loaded.end() && !nx->second->prefetched) { ahead = nx->second.get(); ahead->prefetched = true; ++n_ahead; }
                }
                if (ahead) {
                    const char *tp[2] = {ahead->text[0].p, ahead->text[1].p};
                    const u64 tb[2] = {ahead->bytes[0], ahead->bytes[1]};
                    chk(ctx, bns_text_prefetch(ctx, tp, tb, 2), "bns_text_prefetch");
                }
                if (!turns.wait(b)) break;
                // ---- this block's turn
                const double t0 = tnow();
                // both starts inside their buffers?  (file 1: always, by the limit rule; file 2: as long as it has not drifted by more than ROOM)
                if (pos[0] < j->off[0] || pos[0] > j->off[0] + j->bytes[0] || pos[1] < j->off[1] || pos[1] > j->off[1] + j->bytes[1]) {
                    std::lock_guard<std::mutex> lk(mu);
                    handed_back = true; turns.halt();
                    break;
                }
                BlockCalls calls(c, ctx, sink, pool, n_jobs);
                calls.n_streams = 2;
                for (int s = 0; s < 2; ++s) { calls.tp[s] = j->text[s].p + (pos[s] - j->off[s]); calls.tb[s] = j->off[s] + j->bytes[s] - pos[s]; }
                calls.limit = j->last ? ~0ULL : (j->off[0] + B) - pos[0];
                calls.flags = (j->last ? BNS_TEXT_FINAL : 0) | BNS_TEXT_TRIM_READNO;
                calls.parse();
                pos[0] += calls.used[0]; pos[1] += calls.used[1];
                // done with the block: file 1 handed over everything that starts in it (the last block: whatever pairs there were)
                const bool ok = calls.status == BNS_TEXT_OK && (j->last || pos[0] >= j->off[0] + B);
                if (ok && b + 1 == n_blocks && (pos[0] < fsize[0] || pos[1] < fsize[1])) {
                    // kseq_declare.h:116-120 / 134-137: one file holds more records than the other
                    std::fprintf(stderr, "[W::%s] the %s file has fewer sequences.\n", "bseq_read", pos[0] < fsize[0] ? "2nd" : "1st");
                }
                if (ok) turns.pass();
                else { std::lock_guard<std::mutex> lk(mu); handed_back = true; turns.halt(); }
                // ---- behind the turn (the block's buffers are needed once more only when the hit runs did not fit: kept until then)
                calls.finish();
                std::lock_guard<std::mutex> lk(mu);
                t_call += tnow() - t0;
                units += calls.units;
                spare.push_back(std::move(j));
                cv.notify_all();
                if (!ok) break;
            }
        } catch (const std::exception &e) { { std::lock_guard<std::mutex> lk(mu); fail_with(e.what()); } turns.halt(); }
    };
    std::vector<std::thread> th;
    for (unsigned g = 1; g < G; ++g) th.emplace_back(worker, g);
    worker(0);
    for (auto &t : th) t.join();
    { std::lock_guard<std::mutex> lk(mu); cancel = true; cv.notify_all(); }
    for (auto &t : readers) t.join();
    for (bns_ctx *cx : c.ctxs_) (void)bns_text_prefetch(cx, nullptr, nullptr, 0);          // (blocks uploaded ahead of a call that never came)
    if (!error.empty()) { sink.finish(0, true); die(error); }
    sink.finish(n_jobs);
    units_done = units;
    if (timing)
        std::fprintf(stderr, "[timing] pair of files, text on the device: %llu jobs, blocks of %llu + %llu MiB on %u device(s), %u readers: pread %.3f s (summed), calls %.3f (summed), format %.3f, write %.3f; "
                             "waited %.3f s for blocks, %llu uploads started ahead of their call%s\n",
                     (unsigned long long)n_jobs, (unsigned long long)(B >> 20), (unsigned long long)(B2 >> 20), G, R, t_read, t_call, sink.t_format, sink.t_write, t_idle,
                     (unsigned long long)n_ahead, handed_back ? "; the host parser takes the rest" : "");
    return !handed_back;
}

}  // namespace bns
