// bns_host_capi.cpp -- plain-C exports of the host-side readers/formatters (libbns_host.so) so that Python
// hosts and the CPU test tier can reach them without a GPU.  No compute here: file formats and text only.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "bns_host.hpp"

using namespace bns;

namespace {
thread_local std::string g_err;
template <class F>
int guard(F &&f)
{
    try { f(); return 0; } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
}  // namespace

extern "C" {

const char *bnsh_last_error(void) { return g_err.c_str(); }

void *bnsh_db_open(const char *path)
{
    Database *db = nullptr;
    if (guard([&] { db = new Database(path); }) != 0) return nullptr;
    return db;
}
void bnsh_db_close(void *h) { delete static_cast<Database *>(h); }
void bnsh_db_info(const void *h, uint32_t *k, uint32_t *w, uint64_t *hdr4, int *spacing_width, uint16_t *gaps)
{
    const Database *db = static_cast<const Database *>(h);
    *k = db->k_; *w = db->w_; *spacing_width = db->spacing_width_;
    hdr4[0] = db->db_.n_buckets; hdr4[1] = db->db_.n_occupied; hdr4[2] = db->db_.size; hdr4[3] = db->db_.upper_bound;
    for (size_t i = 0; i < db->s_.size(); ++i) gaps[i] = db->s_[i];
}
const uint32_t *bnsh_db_flags(const void *h) { return static_cast<const Database *>(h)->db_.flags.data(); }
const uint64_t *bnsh_db_keys(const void *h) { return static_cast<const Database *>(h)->db_.keys.data(); }
const uint32_t *bnsh_db_vals(const void *h) { return static_cast<const Database *>(h)->db_.vals.data(); }
int bnsh_db_write(const void *h, const char *path, int spacing_width)
{
    return guard([&] { static_cast<const Database *>(h)->write(path, spacing_width); });
}
// build a Database from caller arrays (copied) so Python can write a bns.db
void *bnsh_db_from_arrays(uint32_t k, uint32_t w, const uint16_t *gaps, const uint64_t *hdr4, const uint32_t *flags,
                          const uint64_t *keys, const uint32_t *vals)
{
    Database *db = new Database();
    db->k_ = k; db->w_ = w;
    db->s_.assign(k ? k - 1 : 0, 0);
    if (gaps) for (uint32_t i = 0; i + 1 < k; ++i) db->s_[i] = gaps[i];
    db->db_.n_buckets = hdr4[0]; db->db_.n_occupied = hdr4[1]; db->db_.size = hdr4[2]; db->db_.upper_bound = hdr4[3];
    const uint64_t nb = hdr4[0];
    db->db_.flags.assign(flags, flags + (nb < 16 ? 1 : nb >> 4));
    db->db_.keys.assign(keys, keys + nb);
    db->db_.vals.assign(vals, vals + nb);
    return db;
}

int bnsh_parent_map(const char *path, uint32_t **out, uint32_t *n)
{
    return guard([&] {
        std::vector<u32> p = build_parent_map(path);
        *n = (uint32_t)p.size();
        *out = static_cast<uint32_t *>(std::malloc(p.size() * 4));
        std::memcpy(*out, p.data(), p.size() * 4);
    });
}
void bnsh_free(void *p) { std::free(p); }

int bnsh_parse_spacing(const char *s, unsigned k, uint16_t *out, int cap)
{
    const spvec_t v = parse_spacing(s, k);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

// Read every record of one (or two interleaved) FASTA/FASTQ(.gz) files in bseq_read chunks and serialise them
// as name \x1f comment \x1f seq \x1f qual \n.  chunks_out (optional) receives the number of bseq_read calls.
int bnsh_read_fastx_blk(const char *p1, const char *p2, int chunk_size, size_t block_bytes, char **blob, size_t *len, int *chunks_out)
{
    return guard([&] {
        SeqReader r1(p1, block_bytes);
        std::unique_ptr<SeqReader> r2(p2 ? new SeqReader(p2, block_bytes) : nullptr);
        ReadChunk seqs;
        std::string out;
        int chunks = 0;
        while (bseq_read(chunk_size, r1, r2.get(), seqs) > 0) {
            ++chunks;
            for (const bseq1_t &b : seqs.recs) {
                out += b.name; out.push_back('\x1f'); out += b.comment; out.push_back('\x1f');
                out += b.seq; out.push_back('\x1f'); out += b.qual; out.push_back('\n');
            }
        }
        *len = out.size();
        *blob = static_cast<char *>(std::malloc(out.size() + 1));
        std::memcpy(*blob, out.data(), out.size());
        if (chunks_out) *chunks_out = chunks;
    });
}

// The same serialisation through ChunkSource: one plain file parsed in stretches on parser_threads threads, or two files (p2) parsed
// side by side.  cuts (optional): the
// cut offsets to use instead of find_cut_points' (a test hands it a cut inside a record to see the sequential fallback).
// info_out[0] = stretches, [1] = 1 when a stretch did not end between two records and the rest was parsed sequentially.
int bnsh_read_fastx_par(const char *p1, const char *p2, int chunk_size, unsigned parser_threads, uint64_t segment_bytes, const uint64_t *cuts,
                        int n_cuts, char **blob, size_t *len, int *info_out)
{
    return guard([&] {
        std::vector<u64> forced;
        if (cuts) forced.assign(cuts, cuts + n_cuts);
        ChunkSource src(p1, p2, (unsigned)chunk_size, parser_threads, segment_bytes, cuts ? &forced : nullptr);
        std::string out;
        while (auto seqs = src.next()) {
            for (const bseq1_t &b : seqs->recs) {
                out += b.name; out.push_back('\x1f'); out += b.comment; out.push_back('\x1f');
                out += b.seq; out.push_back('\x1f'); out += b.qual; out.push_back('\n');
            }
            src.recycle(std::move(seqs));
        }
        *len = out.size();
        *blob = static_cast<char *>(std::malloc(out.size() + 1));
        std::memcpy(*blob, out.data(), out.size());
        if (info_out) { info_out[0] = (int)src.stretches(); info_out[1] = src.fell_back() ? 1 : 0; }
    });
}

int bnsh_find_cut_points(const char *path, uint64_t segment_bytes, uint64_t *out, int cap)
{
    const std::vector<u64> v = find_cut_points(path, segment_bytes);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

// readers opened from now on inflate BGZF input on this device as well (-1: CPU threads only)
void bnsh_set_bgzf_device(int device) { set_bgzf_device(device); }

int bnsh_read_fastx(const char *p1, const char *p2, int chunk_size, char **blob, size_t *len, int *chunks_out)
{
    return bnsh_read_fastx_blk(p1, p2, chunk_size, 0, blob, len, chunks_out);
}

// bns::Encoder(Spacer(k, w, gaps), canonicalize)::for_each(func, str, len) collected into out (test hook for the C++ class)
long bnsh_encoder_from_str(unsigned k, const uint16_t *gaps, int canon, unsigned w, int score, const char *str, uint64_t l,
                           uint64_t *out, uint64_t cap)
{
    try {
        spvec_t g;
        if (gaps) g.assign(gaps, gaps + (k - 1));
        Encoder enc(k, g, canon != 0, 0, w, score);
        uint64_t n = 0;
        enc.for_each([&](uint64_t km) { if (n < cap) out[n] = km; ++n; }, str, l);
        return (long)n;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// bns::Encoder(k, gaps, canon)::for_each_hash(func, str, len, hk) collected into out (test hook for the C++ class)
long bnsh_encoder_hash_from_str(unsigned k, const uint16_t *gaps, int canon, unsigned w, unsigned hk, const char *str, uint64_t l,
                                uint64_t *out, uint64_t cap)
{
    try {
        spvec_t g;
        if (gaps) g.assign(gaps, gaps + (k - 1));
        Encoder enc(k, g, canon != 0, 0, w, 0);
        uint64_t n = 0;
        enc.for_each_hash([&](uint64_t h) { if (n < cap) out[n] = h; ++n; }, str, l, hk);
        return (long)n;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

size_t bnsh_genome_name(const char *header, char *buf, size_t cap)
{
    const std::string n = genome_name(header);
    if (n.size() < cap) { std::memcpy(buf, n.data(), n.size()); buf[n.size()] = 0; }
    return n.size();
}

int bnsh_get_taxid(const char *genome_path, const char *seq2tax_path, uint32_t *out)
{
    return guard([&] { *out = get_taxid(genome_path, build_name_hash(seq2tax_path)); });
}

size_t bnsh_kraken_line(const char *name, int l_seq, uint32_t taxon, uint32_t missing, uint32_t ambig, const uint32_t *hits,
                        uint32_t n_hits, char *buf, size_t cap)
{
    bseq1_t b;
    const std::string filler((size_t)l_seq, 'A');            // only its length is printed
    b.name = name; b.seq = filler;
    std::string s;
    append_kraken_classification(std::vector<tax_t>(hits, hits + n_hits), taxon, ambig, missing, b, s);
    if (s.size() <= cap) std::memcpy(buf, s.data(), s.size());
    return s.size();
}

size_t bnsh_fastq_record(const char *name1, const char *seq1, const char *qual1, const char *name2, const char *seq2,
                         const char *qual2, uint32_t taxon, uint32_t missing, uint32_t ambig, const uint32_t *hits,
                         uint32_t n_hits, int verbose, char *buf, size_t cap)
{
    bseq1_t bs[2];
    bs[0].name = name1; bs[0].seq = seq1; bs[0].qual = qual1 ? qual1 : "";
    const int paired = name2 != nullptr;
    if (paired) { bs[1].name = name2; bs[1].seq = seq2; bs[1].qual = qual2 ? qual2 : ""; }
    std::string s;
    append_fastq_classification(std::vector<tax_t>(hits, hits + n_hits), taxon, ambig, missing, bs, s, verbose, paired);
    if (s.size() <= cap) std::memcpy(buf, s.data(), s.size());
    return s.size();
}

}  // extern "C"
