// bns_bgzf_pipeline.cpp -- BGZF files whose text never leaves the devices: the source that inflates their members there
// (BgzfDeviceSource), and the pipelines for one file and for a pair (host side of the classify path; see bns_host.hpp for the reference map).
#include "bns_text_pipeline.hpp"

namespace bns {
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
// straddles two batches).  Batch b is inflated on device b % G (round 6: the members of a BGZF file are independent, so every device
// inflates its own batches into its own memory; one set of readers and one header walk feed them all).
class BgzfDeviceSource : public DeviceTextSource {
public:
    u64 TEXT_MAX = 0;
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
    void stop() override
    {
        cancel();
        if (splitter_.joinable()) splitter_.join();
        for (auto &t : readers_) if (t.joinable()) t.join();
        for (auto &t : inflaters_) if (t.joinable()) t.join();
    }
    ~BgzfDeviceSource() override
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
    bool next(unsigned g, Item &it) override
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
    bool no_batch(u64 seq) override { std::lock_guard<std::mutex> lk(mu_); return seq >= n_batches_; }
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
    char *buf(unsigned g, int t) const override { return static_cast<char *>(dev_[g].tbufs[(size_t)t]); }
    bns_ctx *ctx(unsigned g) const override { return dev_[g].ctx; }
    unsigned devices() const override { return G; }
    void release(unsigned g, int t) override { std::lock_guard<std::mutex> lk(mu_); dev_[g].free_t.push_back(t); cv_.notify_all(); }
    void cancel() override { std::lock_guard<std::mutex> lk(mu_); cancel_ = true; cv_.notify_all(); }
    std::string error() override { std::lock_guard<std::mutex> lk(mu_); return error_; }

    std::string timing_line() override
    {
        char buf[640];
        std::snprintf(buf, sizeof(buf), "%llu members, %.2f GB of text; header walk %.3f s, pread %.3f (summed over %u readers), inflate calls %.3f (summed over %u handles) of which kernel %.3f; "
                                        "page-lock %.3f (summed), first batch inflated after %.3f s, waits: walker for bytes %.3f, inflaters for batches / buffers %.3f (summed), classify for text %.3f",
                      (unsigned long long)n_members, text_total / 1e9, t_split, t_read, R, t_inflate, n_handles, t_kernel, t_pin, t_first_inflated, t_wait_walk, t_wait_inf, t_wait_next);
        return buf;
    }

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

// A BGZF file whose text never leaves the devices (process_device_text over BgzfDeviceSource's batches: batch b inflated on device b % G).
bool process_bgzf_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out, u64 &units_done)
{
    BgzfDeviceSource src(c, fq1);
    return process_device_text(c, src, out, units_done, "BGZF text");
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

// A PAIR of BGZF files, both inflated into device memory (a BgzfDeviceSource each) and paired there.  One device: process_device_text_pair
// (bns_text_pipeline.cpp: a window per side, calls in file order); several: process_bgzf_gpu_pair_multi.
// -> true: everything was classified; false: text handed back after `units_done` pairs (the host parser reads both files and leaves
// those out)
bool process_bgzf_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done)
{
    if (c.ctxs_.size() > 1) return process_bgzf_gpu_pair_multi(c, fq1, fq2, out, units_done);
    BgzfDeviceSource src0(c, fq1), src1(c, fq2);
    return process_device_text_pair(c, src0, src1, out, units_done, "pair of BGZF files");
}

}  // namespace bns
