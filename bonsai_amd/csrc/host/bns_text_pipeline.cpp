// bns_text_pipeline.cpp -- FASTA / FASTQ text parsed on the device: the pipelines that feed bns_classify_text from plain files, BGZF files and pairs of either (host side of the classify path; see bns_host.hpp for the reference map).
#include "bns_text_pipeline.hpp"

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


bool process_device_text(ClassifierGeneric &c, DeviceTextSource &src, std::FILE *out, u64 &units_done, const char *what)
{
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    const unsigned G = src.devices();
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) for (unsigned g = 0; g < G; ++g) (void)bns_set_timing(src.ctx(g), 1);      // (HIP events around the parse and classify kernels: the sums in the timing line)
    JobPool pool;
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { pool.put(std::move(j)); });
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
            bns_ctx *ctx = src.ctx(g);
            for (u64 b = g;; b += G) {
                DeviceTextSource::Item it;
                if (!src.next(g, it)) { if (!src.no_batch(b)) turns.halt(); break; }       // (the input is done -- or the source has stopped: nobody waits for this block's turn)
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
    if (src.gave_up()) handed_back = true;
    if (timing)
        std::fprintf(stderr, "[timing] %s on the device: %llu jobs on %u device(s); %s; classify calls %.3f (their kernels: text %.3f, classify %.3f), format %.3f, write %.3f%s\n",
                     what, (unsigned long long)n_jobs, G, src.timing_line().c_str(), t_call, t_gpu_parse, t_gpu_cls, sink.t_format, sink.t_write,
                     handed_back ? "; the host parser takes the rest" : "");
    return !handed_back;
}


// A PAIR of inputs whose text lies in device memory (a source each, one device) paired there: bns_classify_text with two streams of device
// text -- record i of the one and record i of the other are mates (kseq_declare.h:116-131).  The two sources' batches do not end at the
// same record, so each side keeps a WINDOW: what its last call left, with the next batch behind it (the rest copied into the room in
// front of the new batch's text, device to device) whenever less than LOW bytes are left; a call takes the pairs both windows hold and
// says where it stopped in either.  Calls in input order.
// -> true: everything was classified; false: text handed back after `units_done` pairs (the host parser reads both inputs and leaves
// those out)
bool process_device_text_pair(ClassifierGeneric &c, DeviceTextSource &src0, DeviceTextSource &src1, std::FILE *out, u64 &units_done, const char *what)
{
    units_done = 0;
    std::fflush(out);
    const int ofd = fileno(out);
    bns_ctx *ctx = src0.ctx(0);
    const bool timing = std::getenv("BNS_CLI_TIMING") != nullptr;
    if (timing) (void)bns_set_timing(ctx, 1);
    const bool want_runs = c.get_emit_kraken() != 0, taxon_only = !want_runs;
    JobPool pool;
    TextSink sink(c, ofd, [&](std::unique_ptr<TextJob> j) { pool.put(std::move(j)); });
    struct Side { DeviceTextSource *src; int t = -1; u64 off = 0, len = 0; bool exhausted = false; } side[2] = {{&src0}, {&src1}};
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
                    DeviceTextSource::Item it;
                    if (!d.src->next(0, it)) {
                        const std::string e = d.src->error();
                        if (!e.empty()) die(e);
                        if (d.src->gave_up()) handed_back = true;       // (a stream the device does not take: what was classified so far is good)
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
            if (handed_back) break;
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
        std::fprintf(stderr, "[timing] %s, text on the device: %llu calls; first file: %s; second file: %s; classify calls %.3f (their kernels: text %.3f, classify %.3f), format %.3f, write %.3f%s\n",
                     what, (unsigned long long)n_calls, src0.timing_line().c_str(), src1.timing_line().c_str(), t_call, t_gpu_parse, t_gpu_cls, sink.t_format, sink.t_write,
                     handed_back ? "; the host parser takes the rest" : "");
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
