// bns_gz_pipeline.cpp -- ONE plain gzip file (not BGZF) whose text never leaves the device: compressed bytes up, the stream entered at
// block headers found on the device (bns_inflate_stream_device, csrc/bns_gzstream.hip), its text parsed and classified where it lies
// (host side of the classify path; the reference reads such a file through gzread under kseq: kseq_declare.h:112-145, one zlib inflate).
#include "bns_text_pipeline.hpp"
#include "pgzip.hpp"

namespace bns {

bool gz_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    if (fq2 || c.get_emit_fastq()) return false;
    if (const char *e = std::getenv("BNS_TEXT_GPU")) if (e[0] == '0') return false;
    if (const char *e = std::getenv("BNS_GZ_GPU")) if (e[0] == '0') return false;
    struct stat st;
    if (::stat(fq1, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) return false;
    unsigned char m[3] = {0, 0, 0};
    const int f = ::open(fq1, O_RDONLY);
    if (f < 0) return false;
    const bool gz = ::pread(f, m, 3, 0) == 3 && m[0] == 0x1f && m[1] == 0x8b && m[2] == 8;
    ::close(f);
    return gz && !is_bgzf_file(fq1);
}

// a pair of such files (R1.fastq.gz + R2.fastq.gz: what the reference's process_dataset is usually given, classifier.h:296-337)
bool gz_pair_gpu_wanted(const ClassifierGeneric &c, const char *fq1, const char *fq2)
{
    return fq2 && gz_gpu_wanted(c, fq1, nullptr) && gz_gpu_wanted(c, fq2, nullptr);
}

namespace {
// The file as text in DEVICE memory, batch by batch.  Readers pread the file into page-locked SLOTS (slot r = file bytes
// [r * P, (r + 1) * P + OVER): a call takes whole DEFLATE blocks only, so what it leaves of its slot's tail is in the next slot's head);
// ONE thread makes the calls -- a call starts where the one in front ended, with the 32 KiB behind that --, each into a free text
// buffer behind HEAD bytes of room.  A member's CRC-32 and ISIZE are checked when it ends; several members (cat a.gz b.gz) are taken
// one behind the other; bytes behind the last member that are no gzip header are ignored, as zlib does.  Whatever the device does not
// take (a first block that inflates beyond its chunk's room, a code the decoder rejects) makes the source GIVE UP: the text delivered
// so far is good, the host reader (pgzip / zlib) takes the file and reports what is wrong with it, if anything is.
class GzDeviceSource : public DeviceTextSource {
public:
    u64 TEXT_MAX = 0;
    double t_read = 0, t_calls = 0, t_kernel = 0, t_pin = 0, t_wait_slot = 0, t_wait_buf = 0, t_wait_next = 0, t_first_call = 0, t_first_kernel = 0;
    u64 n_calls = 0, n_chunks = 0, n_chained = 0, n_breaks = 0, text_total = 0, n_members = 0;

    GzDeviceSource(ClassifierGeneric &c, const char *path) : ctx_(c.ctxs_[0]), device_(c.devices_[0])
    {
        fd_ = ::open(path, O_RDONLY);
        if (fd_ < 0) die(std::string("Could not open ") + path + " for reading.");
        fsize_ = (u64)::lseek(fd_, 0, SEEK_END);
        auto env_num = [](const char *name, u64 dflt) { const char *e = std::getenv(name); return e && std::atol(e) > 0 ? (u64)std::atol(e) : dflt; };
        // P: 256 MiB of compressed bytes a call = ~4600 chunks of 64 KiB, what fills the device's 3584 decoder wavefronts (128 MiB: 1.10 s of
        // kernels per 20 GB of text, 256: 0.90, 64: 1.90 -- profiles/r06_gz.txt)
        P_ = std::min<u64>(std::max<u64>(env_num("BNS_GZ_PIECE_MB", 256) << 20, 1u << 20), 1024ull << 20);
        OVER_ = std::max<u64>(P_ / 8, 1u << 20);
        if (const char *e = std::getenv("BNS_GZ_PIECE_BYTES")) { P_ = (u64)std::max(65536L, std::atol(e)); OVER_ = std::max<u64>(P_ / 2, 65536); }     // (tests: many calls per file)
        HEAD = std::min<u64>(env_num("BNS_BGZF_HEAD_MB", 64) << 20, 256ull << 20);
        if (const char *e = std::getenv("BNS_BGZF_HEAD_BYTES")) HEAD = (u64)std::max(4096L, std::min(256L << 20, std::atol(e)));
        TEXT_MAX = std::min<u64>(std::max<u64>(env_num("BNS_GZ_TEXT_MB", 1536) << 20, 1u << 20), (2047ull << 20) - HEAD);
        if (const char *e = std::getenv("BNS_GZ_TEXT_BYTES")) TEXT_MAX = (u64)std::max(65536L, std::min(1L << 30, std::atol(e)));               // (tests: calls that stop at the text's room)
        n_slots_ = std::max<u64>(1, (fsize_ + P_ - 1) / P_);
        R_ = (unsigned)std::max(1, std::min<int>(4, usable_cpus() / 3));
        if (const char *e = std::getenv("BNS_GZ_PREFETCH")) prefetch_ = e[0] != '0';       // (measurements: every call copies its own bytes up first)
        if (const char *e = std::getenv("BNS_GZ_ROOM_RETRY")) room_retry_ = e[0] != '0';   // (tests: the give-up path)
        room_default_ = (unsigned)std::min<u64>(std::max<u64>(env_num("BNS_GZ_RATIO_CAP", 16), 2), 1024);
        const double t_alloc0 = tnow();
        try {
            tbufs_.assign(3, nullptr);
            for (auto &p : tbufs_) chk(ctx_, bns_dev_alloc(ctx_, (size_t)(HEAD + TEXT_MAX) + 4096, &p), "bns_dev_alloc");
            for (unsigned i = 0; i < tbufs_.size(); ++i) free_t_.push_back((int)i);
            chk(ctx_, bns_dev_alloc(ctx_, 32768, &d_window_), "bns_dev_alloc");
            if (bns_inflater_create(device_, &h_) != BNS_OK) die("gzip input: could not open an inflater on the GPU");
            // (the decoder's device buffers for the largest call now -- ~10 GB for 288 MiB: a buffer that grows between two calls is a free and an
            // allocation with the device drained, classify calls and all)
            if (bns_inflate_stream_reserve(h_, std::min<u64>(fsize_, P_ + OVER_)) != BNS_OK) die(std::string("gzip input: ") + bns_inflater_error(h_));
        } catch (...) { free_all(); throw; }
        if (tnow() - t_alloc0 > 0.1 && std::getenv("BNS_CLI_TIMING"))
            std::fprintf(stderr, "[timing] gzip source: device buffers (3 x %.2f GB of text, the decoder's for %.0f MiB calls) took %.3f s to allocate\n",
                         (double)(HEAD + TEXT_MAX) / 1e9, (double)std::min<u64>(fsize_, P_ + OVER_) / 1048576.0, tnow() - t_alloc0);
        for (unsigned r = 0; r < R_; ++r) readers_.emplace_back([this] { read_loop(); });
        caller_ = std::thread([this] { call_loop(); });
    }
    ~GzDeviceSource() override
    {
        const double t0 = tnow();
        stop();
        ready_.clear(); reading_.clear();
        const double t1 = tnow();
        for (Slot *p : all_slots_) delete p;
        const double t2 = tnow();
        free_all();
        if (tnow() - t0 > 0.1 && std::getenv("BNS_CLI_TIMING"))
            std::fprintf(stderr, "[timing] gzip source let go in %.3f s: threads %.3f, page-locked slots %.3f, device buffers %.3f\n", tnow() - t0, t1 - t0, t2 - t1, tnow() - t2);
    }
    GzDeviceSource(const GzDeviceSource &) = delete;
    GzDeviceSource &operator=(const GzDeviceSource &) = delete;

    bool next(unsigned, Item &it) override
    {
        std::unique_lock<std::mutex> lk(mu_);
        const double tw = tnow();
        cv_.wait(lk, [&] { return cancel_ || !out_.empty() || done_; });
        t_wait_next += tnow() - tw;
        if (cancel_ || out_.empty()) return false;
        it = out_.front(); out_.pop_front();
        ++next_out_;
        return true;
    }
    bool no_batch(u64 seq) override { std::lock_guard<std::mutex> lk(mu_); return done_ && !gave_up_ && seq >= n_batches_; }
    char *buf(unsigned, int t) const override { return static_cast<char *>(tbufs_[(size_t)t]); }
    bns_ctx *ctx(unsigned) const override { return ctx_; }
    unsigned devices() const override { return 1; }
    void release(unsigned, int t) override { std::lock_guard<std::mutex> lk(mu_); free_t_.push_back(t); cv_.notify_all(); }
    void cancel() override { std::lock_guard<std::mutex> lk(mu_); cancel_ = true; cv_.notify_all(); }
    std::string error() override { std::lock_guard<std::mutex> lk(mu_); return error_; }
    bool gave_up() override { std::lock_guard<std::mutex> lk(mu_); return gave_up_; }
    void stop() override
    {
        cancel();
        for (auto &t : readers_) if (t.joinable()) t.join();
        if (caller_.joinable()) caller_.join();
    }
    std::string timing_line() override
    {
        char b[640];
        std::snprintf(b, sizeof(b), "%llu member(s), %.2f GB of text in %llu calls (%llu chunks found a block header, %llu taken, %llu calls cut short by a false header, %llu asked again with more room); pread %.3f s (summed over %u readers), "
                                    "inflate calls %.3f of which kernels %.3f (the first call: %.3f / %.3f), page-lock %.3f; waits: caller for bytes %.3f, for a text buffer %.3f, classify for text %.3f%s",
                      (unsigned long long)n_members, text_total / 1e9, (unsigned long long)n_calls, (unsigned long long)n_chunks, (unsigned long long)n_chained, (unsigned long long)n_breaks, (unsigned long long)n_room_,
                      t_read, R_, t_calls, t_kernel, t_first_call, t_first_kernel, t_pin, t_wait_slot, t_wait_buf, t_wait_next, gave_up_ ? why_.c_str() : "");
        return b;
    }

private:
    struct Slot { PinnedBuf comp; u64 r = 0, file_off = 0; size_t bytes = 0; unsigned pieces_left = 0; };
    struct Piece { Slot *s; size_t off, len; };

    void free_all()
    {
        if (h_) { bns_inflater_destroy(h_); h_ = nullptr; }
        for (void *p : tbufs_) if (p) bns_dev_free(ctx_, p);
        tbufs_.clear();
        if (d_window_) { bns_dev_free(ctx_, d_window_); d_window_ = nullptr; }
        if (fd_ >= 0) { ::close(fd_); fd_ = -1; }
    }
    void fail_with(const std::string &w) { if (error_.empty()) error_ = w; cancel_ = true; cv_.notify_all(); }     // (mu_ held)

    // ---- readers: slots of the file into page-locked memory, piece by piece, at most three slots ahead of the caller
    void read_loop()
    {
        try {
            for (;;) {
                Piece pc{nullptr, 0, 0};
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    for (;;) {
                        if (cancel_ || done_) return;
                        if (!pieces_.empty()) { pc = pieces_.front(); pieces_.pop_front(); break; }
                        if (next_slot_ < n_slots_ && (!spare_.empty() || all_slots_.size() < 3)) {
                            Slot *sl;
                            if (!spare_.empty()) { sl = spare_.back(); spare_.pop_back(); }
                            else { sl = new Slot(); all_slots_.push_back(sl); }
                            sl->r = next_slot_++;
                            sl->file_off = sl->r * P_;
                            sl->bytes = (size_t)std::min<u64>(fsize_ - sl->file_off, P_ + OVER_);
                            reading_[sl->r] = sl;
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
                            if (!np) { ready_[sl->r] = sl; reading_.erase(sl->r); }
                            cv_.notify_all();
                            continue;
                        }
                        if (next_slot_ >= n_slots_ && reading_.empty()) return;
                        cv_.wait(lk);
                    }
                }
                const double t0 = tnow();
                pread_all(fd_, pc.s->comp.p + pc.off, pc.len, pc.s->file_off + pc.off, "gzip stream");
                const double t1 = tnow();
                std::lock_guard<std::mutex> lk(mu_);
                t_read += t1 - t0;
                if (--pc.s->pieces_left == 0) { ready_[pc.s->r] = pc.s; reading_.erase(pc.s->r); }
                cv_.notify_all();
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu_); fail_with(e.what()); }
    }

    // slot r, read (nullptr: cancelled); the slots in front of it go back to the readers
    Slot *slot(u64 r)
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (auto it = ready_.begin(); it != ready_.end() && it->first < r;) { spare_.push_back(it->second); it = ready_.erase(it); cv_.notify_all(); }
        const double tw = tnow();
        cv_.wait(lk, [&] { return cancel_ || ready_.count(r); });
        t_wait_slot += tnow() - tw;
        return cancel_ ? nullptr : ready_[r];
    }
    // slot r if it has been read already (no waiting)
    Slot *peek(u64 r) { std::lock_guard<std::mutex> lk(mu_); auto it = ready_.find(r); return it == ready_.end() ? nullptr : it->second; }
    // a slot's bytes to the device ahead of the calls on it (bns_inflate_stream_prefetch: on a stream of its own, under the kernels of the
    // calls on the slot in front); once per slot
    void bring_up(Slot *sl)
    {
        if (!prefetch_ || sl->r < next_up_) return;
        if (bns_inflate_stream_prefetch(h_, reinterpret_cast<const uint8_t *>(sl->comp.p), sl->bytes) != BNS_OK) die(std::string("bns_inflate_stream_prefetch: ") + bns_inflater_error(h_));
        next_up_ = sl->r + 1;
    }
    void give_up(const std::string &why)
    {
        std::lock_guard<std::mutex> lk(mu_);
        gave_up_ = true; done_ = true; why_ = "; gave up: " + why;
        cv_.notify_all();
    }
    void emit(int tbuf, u64 text_bytes, bool last)
    {
        std::lock_guard<std::mutex> lk(mu_);
        Item it; it.seq = n_emitted_++; it.tbuf = tbuf; it.text_bytes = text_bytes; it.last = last;
        out_.push_back(it);
        text_total += text_bytes;
        if (last) { n_batches_ = n_emitted_; done_ = true; }
        cv_.notify_all();
    }

    // ---- the caller: one call behind the other
    void call_loop()
    {
        try {
            u64 pos_bit = 0;                                   // where the stream goes on: a bit position in the FILE
            bool fresh = true;                                 // ... the first block of a member
            u32 crc = 0; u64 isize = 0;
            // Bytes a call is given: what it inflates beyond its text buffer's room is inflated again by the next call, so no more than the
            // room holds at the ratio seen so far -- and the FIRST call a quarter of a piece (its text starts the classify side, and says the ratio)
            double ratio = 0;
            // the first member's header
            {
                Slot *s0 = slot(0);
                if (!s0) return;
                const u64 he = pgz::gzip_header_end(reinterpret_cast<const uint8_t *>(s0->comp.p), s0->bytes, 0);
                if (!he || he >= fsize_) { give_up("no gzip header the device path knows"); return; }
                pos_bit = he * 8;
            }
            for (;;) {
                const u64 byte = pos_bit >> 3;
                const u64 r = std::min<u64>(byte / P_, n_slots_ - 1);
                Slot *sl = slot(r);
                if (!sl) return;
                bring_up(sl);
                if (Slot *nx = peek(r + 1)) bring_up(nx);
                const bool final_slot = sl->file_off + sl->bytes >= fsize_;
                const size_t off = (size_t)(byte - sl->file_off);
                if (off >= sl->bytes) { give_up("the stream ends inside a member"); return; }
                int tb = -1;
                {
                    std::unique_lock<std::mutex> lk(mu_);
                    const double tw = tnow();
                    cv_.wait(lk, [&] { return cancel_ || !free_t_.empty(); });
                    t_wait_buf += tnow() - tw;
                    if (cancel_) return;
                    tb = free_t_.back(); free_t_.pop_back();
                }
                bns_gz_result res{};
                const double t0 = tnow();
                const size_t avail = sl->bytes - off;
                size_t give;
                int rc;
                unsigned room = 0;                             // this call's room for symbols (0: the default)
                for (u64 at_least = 0;;) {
                    give = avail;
                    if (!final_slot || give > (8u << 20)) {
                        const u64 fit = ratio > 0 ? (u64)(0.85 * (double)TEXT_MAX / ratio) : std::max<u64>(P_ / 4, 65536);
                        give = (size_t)std::min<u64>(give, std::max<u64>(std::max<u64>(fit, at_least), std::min<u64>(8u << 20, P_ / 4)));
                    }
                    // (more room, fewer bytes: the decoder's buffers are chunks x room)
                    if (room) give = (size_t)std::min<u64>(give, std::max<u64>((P_ + OVER_) * room_default_ / room, 65536));
                    rc = bns_inflate_stream_device(h_, reinterpret_cast<const uint8_t *>(sl->comp.p) + off, give, pos_bit & 7u, fresh ? nullptr : d_window_,
                                                   static_cast<char *>(tbufs_[(size_t)tb]) + HEAD, TEXT_MAX, d_window_, &res);
                    // (a first block that does not end inside the bytes given: more of them, as long as the slot has more)
                    if (rc == BNS_OK && res.status == BNS_INF_IN_OVERRUN && give < avail) { at_least = 2 * (u64)give; continue; }
                    // a first block that inflates beyond its chunk's room (text that compresses 100:1): once more with eight times the room
                    if (rc == BNS_OK && res.status == BNS_INF_OUT_OVERFLOW && room_retry_ && (room ? room : room_default_) < 1024) {
                        room = std::min(1024u, (room ? room : room_default_) * 8);
                        (void)bns_inflate_stream_room(h_, room);
                        { std::lock_guard<std::mutex> lk(mu_); ++n_room_; }
                        continue;
                    }
                    break;
                }
                if (room) (void)bns_inflate_stream_room(h_, 0);
                const bool whole_tail = final_slot && give == avail;                 // (the call saw the file's last byte)
                const double t1 = tnow();
                if (rc != BNS_OK) die(std::string("bns_inflate_stream_device: ") + bns_inflater_error(h_));
                {
                    std::lock_guard<std::mutex> lk(mu_);
                    t_calls += t1 - t0; t_kernel += std::max(0.f, bns_inflater_last_kernel_ms(h_)) * 1e-3;
                    if (!n_calls) { t_first_call = t1 - t0; t_first_kernel = std::max(0.f, bns_inflater_last_kernel_ms(h_)) * 1e-3; }
                    ++n_calls; n_chunks += res.n_chunks; n_chained += res.n_chained; n_breaks += res.stop_why == 1 ? 1 : 0;          // (a chunk that did not end at the next one's header: that header was none)
                }
                if (res.status != BNS_INF_OK) {
                    release(0, tb);
                    give_up(res.status == BNS_INF_IN_OVERRUN ? std::string(whole_tail ? "the stream ends inside a block" : "a block longer than the bytes of a call")
                            : res.status == BNS_INF_OUT_OVERFLOW ? "a block inflates beyond its chunk's room"
                            : "the decoder rejects a block (BNS_INF code " + std::to_string(res.status) + " at byte " + std::to_string(byte) + ", bit " + std::to_string(pos_bit & 7u) + ")");
                    return;
                }
                if (res.end_bit > 8) ratio = std::max(ratio, (double)res.text_bytes / ((double)res.end_bit / 8.0));
                crc = bns_crc32_combine(crc, res.crc32, res.text_bytes);
                isize += res.text_bytes;
                pos_bit = (sl->file_off + off) * 8 + res.end_bit;
                fresh = false;
                bool last = false;
                if (res.member_end) {
                    // the trailer: CRC-32 and ISIZE at the next byte boundary; behind it the next member's header, or the end of the data
                    u64 tr = (pos_bit + 7) >> 3;
                    if (tr + 8 > fsize_) die("truncated gzip member (no trailer)");
                    Slot *ts = sl;
                    if (tr + 8 + 4096 > sl->file_off + sl->bytes && !final_slot) { ts = slot(r + 1); if (!ts) { release(0, tb); return; } }
                    const unsigned char *t = reinterpret_cast<const unsigned char *>(ts->comp.p) + (tr - ts->file_off);
                    const u32 want_crc = t[0] | ((u32)t[1] << 8) | ((u32)t[2] << 16) | ((u32)t[3] << 24);
                    const u32 want_isize = t[4] | ((u32)t[5] << 8) | ((u32)t[6] << 16) | ((u32)t[7] << 24);
                    if (crc != want_crc || (u32)isize != want_isize) die("gzip member does not inflate to its recorded checksum and size");
                    { std::lock_guard<std::mutex> lk(mu_); ++n_members; }
                    tr += 8;
                    // (a file of MANY SMALL members -- gzip members of a few hundred KB that are not BGZF -- is a call and a drained stream per
                    // member here, and 16 threads' worth of independent work for the host reader: its file)
                    if (n_members >= 8 && tr / n_members < (1u << 20) && tr + 18 <= fsize_) {
                        emit(tb, res.text_bytes, false);
                        give_up("many small members (" + std::to_string(n_members) + " in the first " + std::to_string(tr >> 10) + " KiB)");
                        return;
                    }
                    u64 he = 0;
                    if (tr + 18 <= fsize_) he = pgz::gzip_header_end(reinterpret_cast<const uint8_t *>(ts->comp.p), ts->bytes, tr - ts->file_off);
                    if (!he || ts->file_off + he >= fsize_) last = true;          // (nothing, or no gzip header, behind the member: the data ends here)
                    else { pos_bit = (ts->file_off + he) * 8; fresh = true; crc = 0; isize = 0; }
                }
                if (res.text_bytes || last) emit(tb, res.text_bytes, last);
                else release(0, tb);
                if (last) return;
            }
        } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(mu_); fail_with(e.what()); }
    }

    bns_ctx *ctx_;
    int device_;
    int fd_ = -1;
    u64 fsize_ = 0, P_ = 0, OVER_ = 0, n_slots_ = 0;
    unsigned R_ = 1;
    bns_inflater *h_ = nullptr;
    void *d_window_ = nullptr;
    std::vector<void *> tbufs_;
    std::vector<int> free_t_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Slot *> spare_, all_slots_;
    std::deque<Piece> pieces_;
    std::map<u64, Slot *> reading_, ready_;
    std::deque<Item> out_;
    u64 next_slot_ = 0, n_emitted_ = 0, next_out_ = 0, n_batches_ = ~0ULL;
    bool cancel_ = false, done_ = false, gave_up_ = false;
    bool prefetch_ = true, room_retry_ = true;
    unsigned room_default_ = 16;
    u64 n_room_ = 0;                                           // calls asked again with more room
    u64 next_up_ = 0;                                          // slots below this one have been brought up
    std::string error_, why_;
    std::vector<std::thread> readers_;
    std::thread caller_;
};
}  // namespace

// -> true: the whole file was classified.  false: after `units_done` units the device path stopped (text handed back, or a stream the
// device decoder does not take): the caller reads the file with the host reader and leaves those units out.
bool process_gz_gpu(ClassifierGeneric &c, const char *fq1, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    std::unique_ptr<GzDeviceSource> src;
    // (no room on the device for the decoder's buffers -- ~15 GB beside a table that fills the HBM --: the host reader's file, nothing printed yet)
    try { src = std::make_unique<GzDeviceSource>(c, fq1); }
    catch (const std::exception &e) { std::fprintf(stderr, "[gzip on the device] %s: the host reader takes the file\n", e.what()); return false; }
    return process_device_text(c, *src, out, units_done, "gzip text");
}

// A pair of plain gzip files: a source each (their inflate calls side by side on handles of their own), mates paired on the device.
bool process_gz_gpu_pair(ClassifierGeneric &c, const char *fq1, const char *fq2, std::FILE *out, u64 &units_done)
{
    units_done = 0;
    std::unique_ptr<GzDeviceSource> src0, src1;
    try { src0 = std::make_unique<GzDeviceSource>(c, fq1); src1 = std::make_unique<GzDeviceSource>(c, fq2); }
    catch (const std::exception &e) { std::fprintf(stderr, "[gzip on the device] %s: the host readers take the files\n", e.what()); return false; }
    return process_device_text_pair(c, *src0, *src1, out, units_done, "pair of gzip files");
}

}  // namespace bns
