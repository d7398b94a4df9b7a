// bns_chunks.cpp -- where a plain file can be cut, and ChunkSource: bseq_read chunks of one or two files in input order on several parser threads (host side of the classify path; see bns_host.hpp for the reference map).
#include "bns_host_internal.hpp"

namespace bns {
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
long find_record_start(const char *b, size_t n, bool fastq)
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
                        auto tail = std::make_shared<TextBlock>(TEXT_BLOCK_HEAD + (size_t)c + 8);
                        tail->begin = TEXT_BLOCK_HEAD;
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
            m.stretch_blocks = (size_t)std::max<u64>(1, segment_bytes / m.feeder->raw_block_bytes());
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


}  // namespace bns
