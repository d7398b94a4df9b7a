// bns_reader.cpp -- FASTA / FASTQ text -> records on the host: SeqReader (kseq semantics on block views), BGZF members and one gzip stream inflated on many threads, bseq_read (host side of the classify path; see bns_host.hpp for the reference map).
#include "bns_host_internal.hpp"

namespace bns {
// ---------------------------------------------------------------------------------------------- FASTA/FASTQ
namespace {
constexpr size_t RAW_BLOCK = 4u << 20;

inline bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace() in the C locale
}  // namespace



BlockPool &block_pool() { static BlockPool *p = new BlockPool; return *p; }

// ---- BGZF (blocked gzip, what bgzip / htslib and many sequencing pipelines write): every gzip member is at most 64 KiB of text and
// carries its own compressed size in a 'BC' extra subfield, so members can be found without inflating and inflated side by side.
// (One plain gzip stream cannot: DEFLATE has no sync points -- that input keeps its one inflate thread, decoupled from the parser.)
namespace {
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
}  // namespace
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
namespace {
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
    static constexpr size_t HEAD = TEXT_BLOCK_HEAD;
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
            // (klib/kseq.h:135 -- not for a ONE-byte line that ends the input without a newline: ks_getc has taken the byte, ks_getuntil2
            // finds nothing behind it and returns -1 in front of the test)
            if (acc->size() > 1 && acc->back() == '\r' && !(!nl && e - p == 1)) acc->pop_back();
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
    // (klib/kseq.h:217: `while (ks_getuntil2(...) >= 0 && qual.l < seq.l);` -- at least ONE line is read, also behind an empty sequence:
    // a line there that is not empty -- the next record's header, say -- is quality that is too long, error -2)
    for (bool first = true; first || qual.size() < seq.size(); first = false) {
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
size_t SeqReader::raw_block_bytes() const { return impl_->raw_block; }
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


}  // namespace bns
