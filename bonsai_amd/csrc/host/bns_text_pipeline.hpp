// bns_text_pipeline.hpp -- what the device-text pipelines share (bns_text_pipeline.cpp: plain files; bns_bgzf_pipeline.cpp: BGZF files;
// bns_gz_pipeline.cpp: one plain gzip stream): the result buffers of a bns_classify_text call and their ordered writer, the turn chain
// that cuts an input's blocks in order, the library calls on one block, and the interface of a source of text that lies in device memory.
#pragma once
#include "bns_host_internal.hpp"

namespace bns {
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

unsigned format_text_job(ClassifierGeneric &c, const TextJob &j, std::vector<ClassifierGeneric::Work::Part> &parts);

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
void size_text_job(bns_ctx *ctx, TextJob &j, bns_text_out &o, bool taxon_only, u64 cap, u64 names_cap, u64 runs_cap);

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
            const double tc0 = tnow();
            chk(ctx, bns_classify_text(ctx, cp_, cb_, n_streams, lim_, flags | BNS_TEXT_DEFER, cap_, &o, &first_), "bns_classify_text");
            if (tnow() - tc0 > 0.3 && std::getenv("BNS_CLI_TIMING"))
                std::fprintf(stderr, "[timing] a bns_classify_text call (first half) took %.3f s: %llu bytes of text, %llu records, its parse kernels %.1f ms\n", tnow() - tc0,
                             (unsigned long long)(cb_[0] + cb_[1]), (unsigned long long)first_.n_records, first_.ms_parse);
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
        const double tf0 = tnow();
        chk(ctx, bns_text_finish(ctx, &fin), "bns_text_finish");
        if (tnow() - tf0 > 0.3 && std::getenv("BNS_CLI_TIMING"))
            std::fprintf(stderr, "[timing] a bns_text_finish call took %.3f s: %llu records, its classify kernels %.1f ms\n", tnow() - tf0, (unsigned long long)fin.n_records, fin.ms_classify);
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

// A source of text that LIES IN DEVICE MEMORY, batch by batch in input order (BGZF members or one gzip stream inflated there).  Batch b
// is left on device b % devices() in one of the source's text buffers, behind HEAD bytes of room (for what the caller could not finish
// of the batch in front: the record that straddles two batches).
class DeviceTextSource {
public:
    struct Item { u64 seq = 0; int tbuf = -1; u64 text_bytes = 0; bool last = false; };
    u64 HEAD = 0;
    virtual ~DeviceTextSource() {}
    // device g's next batch (batches g, g + G, ...) in input order; false: there is none (the input is done, cancel() was called, the
    // source gave up, or a thread failed: error())
    virtual bool next(unsigned g, Item &it) = 0;
    // true once the source knows that the input has no batch `seq`
    virtual bool no_batch(u64 seq) = 0;
    virtual char *buf(unsigned g, int t) const = 0;
    virtual bns_ctx *ctx(unsigned g) const = 0;
    virtual void release(unsigned g, int t) = 0;
    virtual void cancel() = 0;
    virtual void stop() = 0;                                    // everybody home (the source's figures are final after this)
    virtual std::string error() = 0;
    virtual unsigned devices() const = 0;
    // the source could not go on with the device (text that inflates beyond its room, ...): what it delivered is good, the host reader takes the input
    virtual bool gave_up() { return false; }
    virtual std::string timing_line() { return std::string(); }
};

// A batch's text buffer is let go when both are done with it: the block behind it (has taken the unfinished rest) and the batch's own
// classify call (which reads the text again when the hit runs did not fit the first time).
struct TextHold {
    DeviceTextSource *src = nullptr;
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
inline void take_rest(DeviceTextSource &src, unsigned g, int tbuf, Rest &rest)
{
    if (rest.len)
        chk(src.ctx(g), bns_dev_copy_peer(src.ctx(g), src.buf(g, tbuf) + src.HEAD - rest.len, src.ctx(rest.hold->dev), src.buf(rest.hold->dev, rest.hold->tbuf) + rest.off, (size_t)rest.len),
            "bns_dev_copy_peer");
    if (rest.hold) { rest.hold->drop(); rest.hold.reset(); }
}

// Text that never leaves the devices: the source's batches, what the batch in front could not finish copied in front of the next one's
// text, bns_classify_text on it where it lies -- cut in input order (Turns), classified side by side --, names and results down.
// -> true: the whole input was classified.  false: text was handed back (not in the kernels' regular form, or the source gave up) after
// `units_done` units had been printed: the caller reads the input with the host parser and leaves those out.
bool process_device_text(ClassifierGeneric &c, DeviceTextSource &src, std::FILE *out, u64 &units_done, const char *what);
// ... and a pair of such inputs on one device, mates paired record for record there (same contract)
bool process_device_text_pair(ClassifierGeneric &c, DeviceTextSource &src0, DeviceTextSource &src1, std::FILE *out, u64 &units_done, const char *what);
}  // namespace bns
