// bns_api.hip -- the C ABI of include/bonsai_amd.h: context, HBM residency, kernel launches.
// Single translation unit for the device library: the kernels are included so their templates are visible.
#include "../../include/bonsai_amd.h"
#include "bns_kernels.hip"

#include <dlfcn.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <rccl/rccl.h>          // types and declarations only: the library itself is dlopen()ed (see Rccl below)
#include <mutex>
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <chrono>
#include <type_traits>
#include <thread>
#include <vector>

using namespace bns;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// ctx->small: the few device words the entry points share with their kernels.  Every entry point zeroes ITS words on its own
// stream before use; nothing here persists across calls.
struct SmallLayout {
    u32 work_counter; u32 pad0[31];            // [0,128)   classify_kernel's chunk counter, on a line of its own (80 M atomics/s)
    u32 ovf_count, max_len; u32 pad1[30];      // [128,256) classify: units handed to the overflow kernel; longest read of the batch
    unsigned long long load_cnt[8];            // [256,320) table load / device build counters
    unsigned long long runs_cursor;            // [320,328) hit_runs_kernel's output cursor
};
constexpr size_t SMALL_BYTES = 512;
constexpr size_t SMALL_CLASSIFY_ZERO = offsetof(SmallLayout, max_len) + sizeof(u32);   // what a classify call zeroes, in one memset
static_assert(sizeof(SmallLayout) <= SMALL_BYTES, "ctx->small is allocated with SMALL_BYTES");
static_assert(offsetof(SmallLayout, load_cnt) >= SMALL_CLASSIFY_ZERO, "classify's memset must not reach the other entry points' words");

// bns_debug_set bits (tests and profiling; 0 in production)
constexpr int BNS_DBG_PLACE_FAIL = 0x100;           // every 61st bucket pretends to have no perfect hash (-> overflow table)
constexpr int BNS_DBG_SPACED_NOCLUSTER = 0x200;     // spaced seeds: m = k, every k-mer its own bucket
constexpr int BNS_DBG_PEXT_OFF = 0x400;             // spaced seeds: gather run by run instead of through the compress network
constexpr int BNS_DBG_FORCE_RCCL = 0x800;           // bns_load_table_multi with ONE context still broadcasts through RCCL (1-rank communicator)
constexpr int BNS_DBG_STREAM_LOAD = 0x1000;         // bns_load_table streams the host arrays even when they would fit
constexpr int BNS_DBG_OVC_OFF = 0x2000;             // classify: never the cooperative overflow lookup (A/B of the two forms)
constexpr int BNS_DBG_OVC_ON = 0x8000;              // classify: always the cooperative overflow lookup
constexpr int BNS_DBG_PLAIN_FILL = 0x10;            // bns_load_table: keys in arrival order even into a crowded table (A/B of the group-aware fill)
constexpr int BNS_DBG_GROUP_FILL = 0x20;            // bns_load_table: the group-aware fill even for a table with room (tests reach it on small tables)
constexpr int BNS_DBG_BATCH_TINY = 0x40;            // bns_classify_text: a classify launch per >= 64 records (many batches on small texts)
constexpr int BNS_DBG_SLICE_8K = 0x4000;            // bns_classify_batch uploads in 8 KiB slices (the slicing logic on small batches)
constexpr int BNS_DBG_STREAM_CHUNK_SHIFT = 16, BNS_DBG_STREAM_CHUNK_MASK = 0x1F << 16;   // log2 of the streamed chunk (0 = 27)
constexpr int BNS_DBG_SPACED_M_SHIFT = 24, BNS_DBG_SPACED_M_MASK = 0x1F << 24;           // spaced seeds: force the run minimizer's m

}  // namespace

struct bns_text_work;                                     // bns_ingest.hip: workspace of bns_classify_text
void text_work_free(struct bns_ctx *ctx);

struct bns_ctx {
    int device = 0;
    bns_text_work *text_work = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;                   // host-buffer entry points: uploads of the next slice while one is classified
    hipStream_t back_stream = nullptr;                   // ... and the copy-back of a classified slice's results, behind neither of the two
    hipEvent_t slice_ev[16] = {}, done_ev[16] = {};
    std::string err;
    int n_cu = 256;
    // encoder
    bool enc_set = false;
    u32 k = 0, c = 0;
    bool canon = true, spaced = false, spaced_intended = false;
    u16 pos[32] = {0};
    u32 n_runs = 0;
    u8 run_start[32] = {0}, run_len[32] = {0};
    u64 sample_mask = 0;
    u32 table_m = 0;            // minimizer length the MINBUCKET table was built with
    u32 min_span_req = 0;       // bns_set_minimizer_span: 0 = chosen from the db when the table is loaded
    u64 n_spilled = 0;          // keys of the MINBUCKET table that are not in their home bucket
    bool group_fill = false;    // the table was filled group by group (crowded tables: minbucket_tagcount_kernel)
    u32 table_len = 0, table_shift = 0, table_canon = 1;   // MinSpec of the loaded table (where in the key the minimizer lives)
    u32 sp_run_len = 0, sp_run_shift = 0;                  // spaced seeds: the mask's longest run of adjacent sampled bases
    u32 pext_on = 0, pext_n1 = 0, pext_steps[2] = {0, 0}, pext_top[2] = {0xFF, 0xFF};  // compress network for masks with many runs (ClassifyParams)
    u64 pext_mask[2] = {0, 0}, pext_mv[2][6] = {{0}};
    u32 win = 0;                // Spacer window in bases (0 / <= comb: unwindowed)
    int score = 0;              // BNS_SCORE_*
    // table
    int layout = -1;
    u64 kh_nb = 0;
    const u32 *kflags = nullptr;
    const u64 *kkeys = nullptr;
    const u32 *kvals = nullptr;
    bool own_khash = false;
    Slot *slots = nullptr;          // BUCKET: n_slots x 16 B; MINBUCKET: the same allocation viewed as MinBucket[n_slots / 8]
    u64 n_slots = 0;                // allocated 16-byte slots (MINBUCKET: 8 per bucket, n_mb + MINB_MAX_CHAIN - 1 buckets)
    u32 n_mb = 0;                   // MINBUCKET: buckets a key can call home (any count: the index is a multiply-high)
    Slot *ovf_slots = nullptr;      // MINBUCKET: plain-hashed overflow table for keys beyond MINB_MAX_CHAIN buckets
    u64 n_ovf_slots = 0, n_ovf_keys = 0;
    u64 n_keys = 0;
    u32 slots_log2_req = 0;
    u64 n_buckets_req = 0;          // bns_set_table_buckets: exact number of home buckets (0 = automatic)
    int fill_req = -1;              // bns_set_table_fill (and replicas taking the root's choice): -1 the loader decides, 0 arrival order, 1 group-aware fill
    int wide_req = -1;              // bns_set_minimizer_identity: -1 chosen from the key count, 0 narrow (32-bit), 1 wide (52-bit)
    bool table_wide = false;        // the loaded MINBUCKET table's identity
    u32 table_span = 0;             // the window candidate (MIN_CANDS span: 15 / 11 / 8) the loaded table was built with; 0: none (spaced seed)
    std::string warn;               // what the last table load has to say about the table it built (bns_table_warning)
    int dbg = 0;
    u32 table_k = 0;            // k the minimizer-clustered layout was built for
    // taxonomy
    TaxNode *nodes = nullptr;
    u32 n_nodes = 0;
    // workspace (grow-only)
    DevBuf words, nmask, ovf_list, scratch, small, records;      // small: ovf_count + misc counters
    DevBuf st_bases, st_offsets, st_out[4], st_hits, st_kmers, st_aux, st_runs[4], st_words, st_nmask, st_bad;   // st_words..: packed host batches   // st_runs: run_start, n_runs, run_tax, run_len
    u32 *h_run_tax = nullptr, *h_run_len = nullptr;      // host side of bns_classify_batch_runs (valid until the next call); page-locked:
    size_t h_run_cap = 0;                                // a copy into pageable memory is staged by the runtime at a fraction of the link rate
    void *peer_stage = nullptr; size_t peer_stage_cap = 0;   // bns_dev_copy_peer between two devices: page-locked staging of this (the receiving) context
    // timing
    bool timing = false;
    static constexpr int EV_RING = 64;
    hipEvent_t ev0[EV_RING] = {nullptr}, ev1[EV_RING] = {nullptr};
    int ev_head = 0, ev_count = 0;          // recorded-but-unsummarised pairs are the last ev_count before ev_head
};

namespace {

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                        \
            return _e == hipErrorOutOfMemory ? BNS_ERR_NOMEM : BNS_ERR_HIP;                        \
        }                                                                                         \
    } while (0)

int fail(bns_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

int ensure(bns_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return BNS_OK;
    // a buffer that GROWS grows by at least a quarter: a stream of batches whose sizes wander (the slices of a text: 207 k records, give
    // or take) would otherwise re-allocate at every new maximum, and a hipFree drains the whole device (6.7 ms each, 45 of them per 7.5 GB
    // of BGZF text in round 5's trace)
    const size_t grown = b.p ? b.cap + b.cap / 4 : 0;
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = (std::max(bytes, grown) + 255) & ~size_t(255);
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return BNS_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

inline unsigned grid_for(const bns_ctx *ctx, u64 items, unsigned per_block, unsigned blocks_per_cu = 8)
{
    const u64 need = (items + per_block - 1) / per_block;
    const u64 cap = (u64)ctx->n_cu * blocks_per_cu;
    return (unsigned)std::max<u64>(1, std::min(need, cap));
}

void fill_params(const bns_ctx *ctx, ClassifyParams &p)
{
    std::memset(&p, 0, sizeof(p));
    p.words = (const u64 *)ctx->words.p;
    p.nmask = (const u32 *)ctx->nmask.p;
    p.slots = ctx->layout == BNS_LAYOUT_MINBUCKET ? ctx->ovf_slots : ctx->slots;
    p.ovf_mask = ctx->n_ovf_slots ? ctx->n_ovf_slots / 4 - 1 : 0;
    p.minb = reinterpret_cast<const MinBucket *>(ctx->slots);
    p.bucket_mask = (ctx->n_slots && ctx->layout == BNS_LAYOUT_BUCKET) ? ctx->n_slots / 4 - 1 : 0;
    p.n_mb = ctx->layout == BNS_LAYOUT_MINBUCKET ? ctx->n_mb : 0u;
    p.min_wide = ctx->table_wide ? 1u : 0u;
    p.kflags = ctx->kflags; p.kkeys = ctx->kkeys; p.kvals = ctx->kvals; p.kh_nb = ctx->kh_nb;
    p.nodes = ctx->nodes; p.n_nodes = ctx->n_nodes;
    p.k = ctx->k; p.c = ctx->c; p.canon = ctx->canon ? 1 : 0; p.dbg = ctx->dbg;
    std::memcpy(p.pos, ctx->pos, sizeof(p.pos));
    p.n_runs = ctx->n_runs; p.sample_mask = ctx->sample_mask; p.m = ctx->table_m ? ctx->table_m : ctx->k;
    p.min_len = ctx->table_m ? ctx->table_len : ctx->k; p.min_shift = ctx->table_m ? ctx->table_shift : 0; p.min_canon = ctx->table_m ? ctx->table_canon : 1;
    p.w = ctx->win > ctx->c ? ctx->win : ctx->c; p.score = ctx->score;
    if (ctx->score == BNS_SCORE_ENTROPY_STRING) {                        // CircusEnt::value() terms, entropy.h:46-47 (qszinv_ = 1./qsz)
        const double qi = 1. / (double)ctx->k;
        for (u32 n = 1; n <= ctx->k && n <= 32; ++n) p.ent_tbl[n] = (double)n * qi * std::log((double)n * qi);
    }
    std::memcpy(p.run_start, ctx->run_start, sizeof(p.run_start)); std::memcpy(p.run_len, ctx->run_len, sizeof(p.run_len));
    p.pext_on = (ctx->dbg & BNS_DBG_PEXT_OFF) ? 0 : ctx->pext_on; p.pext_n1 = ctx->pext_n1;
    std::memcpy(p.pext_steps, ctx->pext_steps, sizeof(p.pext_steps)); std::memcpy(p.pext_mask, ctx->pext_mask, sizeof(p.pext_mask));
    std::memcpy(p.pext_top, ctx->pext_top, sizeof(p.pext_top));
    std::memcpy(p.pext_mv, ctx->pext_mv, sizeof(p.pext_mv));
}

// pack ASCII reads (device) into ctx->words / ctx->nmask
int pack_reads(bns_ctx *ctx, const char *d_bases, const u64 *d_offsets, u64 n_reads, u64 total_bases, hipStream_t st)
{
    const size_t n_words = (size_t)(total_bases >> 5) + n_reads + 2;
    int rc;
    if ((rc = ensure(ctx, ctx->words, n_words * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->nmask, n_words * 4)) != BNS_OK) return rc;
    if (n_reads == 0) return BNS_OK;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(ctx, n_reads, 4)), dim3(256), 0, st, (const u8 *)d_bases, d_offsets,
                       n_reads, (u64 *)ctx->words.p, (u32 *)ctx->nmask.p);
    HIPCHK(ctx, hipGetLastError());
    return BNS_OK;
}

// windows over the emitted stream (-C, real-entropy score) wider than 64 k-mers keep their queue image in global memory
int set_win_scratch(bns_ctx *ctx, ClassifyParams &p, unsigned grid)
{
    p.win_scratch = nullptr;
    const bool stream_windows = !ctx->spaced && (!ctx->canon || ctx->score == BNS_SCORE_ENTROPY_STRING);
    if (!stream_windows || ctx->win <= ctx->c || ctx->win - ctx->c + 1 <= 64) return BNS_OK;
    const size_t per_wave = (size_t)2 * (ctx->win - ctx->c + 65u) * 8;
    const int rc = ensure(ctx, ctx->scratch, (size_t)grid * 4 * per_wave);
    if (rc != BNS_OK) return rc;
    p.win_scratch = (u64 *)ctx->scratch.p;
    return BNS_OK;
}

template <class F>
void dispatch_sp_layout(bool spaced, int layout, F &&f)
{
    if (spaced) {
        if (layout == 2) f(std::true_type{}, std::integral_constant<int, 2>{});
        else if (layout == 1) f(std::true_type{}, std::integral_constant<int, 1>{});
        else f(std::true_type{}, std::integral_constant<int, 0>{});
    } else {
        if (layout == 2) f(std::false_type{}, std::integral_constant<int, 2>{});
        else if (layout == 1) f(std::false_type{}, std::integral_constant<int, 1>{});
        else f(std::false_type{}, std::integral_constant<int, 0>{});
    }
}

void free_table(bns_ctx *ctx)
{
    if (ctx->own_khash) {
        if (ctx->kflags) (void)hipFree((void *)ctx->kflags);
        if (ctx->kkeys) (void)hipFree((void *)ctx->kkeys);
        if (ctx->kvals) (void)hipFree((void *)ctx->kvals);
    }
    ctx->kflags = nullptr; ctx->kkeys = nullptr; ctx->kvals = nullptr; ctx->own_khash = false; ctx->kh_nb = 0;
    if (ctx->slots) (void)hipFree(ctx->slots);
    if (ctx->ovf_slots) (void)hipFree(ctx->ovf_slots);
    ctx->ovf_slots = nullptr; ctx->n_ovf_slots = 0; ctx->n_ovf_keys = 0;
    ctx->slots = nullptr; ctx->n_slots = 0; ctx->n_mb = 0; ctx->n_keys = 0; ctx->layout = -1; ctx->table_wide = false;
}

int ready(bns_ctx *ctx, bool need_table, bool need_tax)
{
    if (!ctx) return BNS_ERR_ARG;
    if (!ctx->enc_set) return fail(ctx, BNS_ERR_STATE, "encoder not configured (bns_set_encoder)");
    if (need_table && ctx->layout < 0) return fail(ctx, BNS_ERR_STATE, "no table loaded (bns_load_table)");
    if (need_table && ctx->layout == BNS_LAYOUT_MINBUCKET && ctx->table_k != ctx->k)
        return fail(ctx, BNS_ERR_STATE, "encoder k changed after a BNS_LAYOUT_MINBUCKET table was built; reload the table");
    if (need_tax && !ctx->nodes) return fail(ctx, BNS_ERR_STATE, "no taxonomy loaded (bns_load_taxonomy)");
    return BNS_OK;
}

}  // namespace

namespace {
// classify_kernel<false, MINBUCKET, KT, NM, SPAN, OVC, WIDE> for the (k, window) pairs the loader can produce with a common k.
// FULL: every form of the overflow lookup and the minimizer identity; otherwise the usual form only (the rest falls back to the
// generic kernel, which reads k from its arguments).
template <int KT, int SPAN, bool FULL>
bool launch_kt(const ClassifyParams &p, unsigned grid, hipStream_t st, bool ovc, bool wide, bool packed)
{
    if ((int)p.k != KT || KT - (int)p.m != SPAN || !p.canon) return false;     // (non-canonical contiguous seeds: the generic kernel)
    if (!FULL && (ovc || wide || packed)) return false;
    if (packed && (ovc || wide)) return false;                   // (packed input: the usual form only; the rest goes to the generic kernels)
    auto go = [&](auto nm) {
        constexpr int NM = decltype(nm)::value;
        if constexpr (FULL) {
            if (packed) { hipLaunchKernelGGL((classify_kernel<false, 2, KT, NM, SPAN, false, false, true>), dim3(grid), dim3(256), 0, st, p); return; }
            if (wide) {
                if (ovc) hipLaunchKernelGGL((classify_kernel<false, 2, KT, NM, SPAN, true, true>), dim3(grid), dim3(256), 0, st, p);
                else     hipLaunchKernelGGL((classify_kernel<false, 2, KT, NM, SPAN, false, true>), dim3(grid), dim3(256), 0, st, p);
                return;
            }
            if (ovc) { hipLaunchKernelGGL((classify_kernel<false, 2, KT, NM, SPAN, true, false>), dim3(grid), dim3(256), 0, st, p); return; }
        }
        hipLaunchKernelGGL((classify_kernel<false, 2, KT, NM, SPAN, false, false>), dim3(grid), dim3(256), 0, st, p);
    };
    if (p.nmates == 1) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 2>{});
    return true;
}
template <int KT, bool FULL>
bool launch_k(const ClassifyParams &p, unsigned grid, hipStream_t st, bool ovc, bool wide, bool packed)
{
    constexpr int S0 = KT - (int)minimizer_len(KT, MIN_CANDS[0]), S1 = KT - (int)minimizer_len(KT, MIN_CANDS[1]), S2 = KT - (int)minimizer_len(KT, MIN_CANDS[2]);
    return launch_kt<KT, S0, FULL>(p, grid, st, ovc, wide, packed) || launch_kt<KT, S1, FULL>(p, grid, st, ovc, wide, packed) ||
           launch_kt<KT, S2, FULL>(p, grid, st, ovc, wide, packed);
}
bool launch_fixed_k(const ClassifyParams &p, unsigned grid, hipStream_t st, bool ovc, bool wide, bool packed)
{
    return launch_k<31, true>(p, grid, st, ovc, wide, packed) || launch_k<21, false>(p, grid, st, ovc, wide, packed) ||
           launch_k<25, false>(p, grid, st, ovc, wide, packed) || launch_k<27, false>(p, grid, st, ovc, wide, packed) ||
           launch_k<32, false>(p, grid, st, ovc, wide, packed);
}
}  // namespace

extern "C" {

int bns_version(void) { return 104; }

int bns_device_pci_bus_id(int device, char *out, int cap)
{
    if (!out || cap < 16) return BNS_ERR_ARG;
    return hipDeviceGetPCIBusId(out, cap, device) == hipSuccess ? BNS_OK : BNS_ERR_HIP;
}

int bns_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* test / profiling aid, not part of the public header: BNS_DBG_* bits (above).  The ablation bits (1: no probe, 2: no vote,
 * 4: no minimizer window; BNS_ABLATION builds only) make results WRONG; the others select code paths that a small test could
 * not reach otherwise (streamed load, sliced upload, one-rank RCCL broadcast). */
int bns_debug_set(bns_ctx *ctx, int bits) { if (!ctx) return BNS_ERR_ARG; ctx->dbg = bits; return BNS_OK; }
#ifdef BNS_WAVE_TIMES
// measurement builds only: (start, end) wall-clock stamps of the 8192 wavefronts of the last classify_kernel launch
extern "C" int bns_debug_wave_times(bns_ctx *ctx, unsigned long long *out16384)
{
    if (!ctx || !out16384) return BNS_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return BNS_ERR_HIP;
    if (hipMemcpyFromSymbol(out16384, HIP_SYMBOL(bns::g_wave_times), 16384 * sizeof(unsigned long long)) != hipSuccess) return BNS_ERR_HIP;
    return BNS_OK;
}
#endif
#ifdef BNS_COUNT_FETCHES
__global__ void ovf_stats_kernel(const bns::Slot *ovf, unsigned long long n_slots, unsigned long long *out)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long occ = 0, full = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < n_slots / 4; b += stride) {
        unsigned c = 0;
        for (int q = 0; q < 4; ++q) c += ovf[b * 4 + q].occ ? 1u : 0u;
        occ += c; full += c == 4u;
    }
    atomicAdd(out, occ); atomicAdd(out + 1, full);
}
extern "C" int bns_debug_ovf_stats(bns_ctx *ctx, unsigned long long *out3)
{
    if (!ctx || !out3) return BNS_ERR_ARG;
    out3[0] = out3[1] = 0; out3[2] = ctx->n_ovf_slots;
    if (!ctx->ovf_slots) return BNS_OK;
    unsigned long long *d = ((SmallLayout *)ctx->small.p)->load_cnt;
    if (hipMemset(d, 0, 16) != hipSuccess) return BNS_ERR_HIP;
    hipLaunchKernelGGL(ovf_stats_kernel, dim3(1024), dim3(256), 0, 0, ctx->ovf_slots, (unsigned long long)ctx->n_ovf_slots, d);
    if (hipDeviceSynchronize() != hipSuccess) return BNS_ERR_HIP;
    if (hipMemcpy(out3, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return BNS_ERR_HIP;
    return BNS_OK;
}
extern "C" int bns_debug_ovf_copy(bns_ctx *ctx, void *host, unsigned long long bytes)
{
    if (!ctx || !host || !ctx->ovf_slots) return BNS_ERR_ARG;
    const unsigned long long have = ctx->n_ovf_slots * sizeof(bns::Slot);
    if (hipMemcpy(host, ctx->ovf_slots, bytes < have ? bytes : have, hipMemcpyDeviceToHost) != hipSuccess) return BNS_ERR_HIP;
    return BNS_OK;
}
// measurement builds only: {distinct 128-byte buckets fetched, probe passes, lanes sent to the overflow table, rounds with such lanes} since the last call (then reset)
extern "C" int bns_debug_fetch_count(bns_ctx *ctx, unsigned long long *out8)
{
    if (!ctx || !out8) return BNS_ERR_ARG;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return BNS_ERR_HIP;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(bns::g_fetch_count), sizeof(z)) != hipSuccess) return BNS_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(bns::g_fetch_count), z, sizeof(z)) != hipSuccess) return BNS_ERR_HIP;
    return BNS_OK;
}
#endif

const char *bns_strerror(int code)
{
    switch (code) {
        case BNS_OK: return "ok";
        case BNS_ERR_ARG: return "bad argument";
        case BNS_ERR_HIP: return "HIP runtime error";
        case BNS_ERR_NOMEM: return "out of memory";
        case BNS_ERR_STATE: return "context not ready";
        case BNS_ERR_TAX_CYCLE: return "taxonomy contains a cycle";
        case BNS_ERR_TAX_RANGE: return "taxid out of range for the flat parent array";
        case BNS_ERR_TABLE: return "inconsistent khash table";
        case BNS_ERR_NO_DEVICE: return "no usable GPU device";
        default: return "unknown error";
    }
}

const char *bns_last_error(const bns_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bns_create(int device, bns_ctx **out)
{
    if (!out) return BNS_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BNS_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BNS_ERR_NO_DEVICE;
    bns_ctx *ctx = new (std::nothrow) bns_ctx();
    if (!ctx) return BNS_ERR_NOMEM;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BNS_ERR_HIP; }
    for (int i = 0; i < bns_ctx::EV_RING; ++i)
        if (hipEventCreate(&ctx->ev0[i]) != hipSuccess || hipEventCreate(&ctx->ev1[i]) != hipSuccess) { delete ctx; return BNS_ERR_HIP; }
    if (hipMalloc(&ctx->small.p, SMALL_BYTES) != hipSuccess) { delete ctx; return BNS_ERR_NOMEM; }
    ctx->small.cap = SMALL_BYTES;
    *out = ctx;
    return BNS_OK;
}

void bns_destroy(bns_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_table(ctx);
    text_work_free(ctx);
    if (ctx->nodes) (void)hipFree(ctx->nodes);
    DevBuf *bufs[] = {&ctx->words, &ctx->nmask, &ctx->ovf_list, &ctx->scratch, &ctx->small, &ctx->records, &ctx->st_bases, &ctx->st_offsets,
                      &ctx->st_out[0], &ctx->st_out[1], &ctx->st_out[2], &ctx->st_out[3], &ctx->st_hits, &ctx->st_kmers, &ctx->st_aux,
                      &ctx->st_runs[0], &ctx->st_runs[1], &ctx->st_runs[2], &ctx->st_runs[3], &ctx->st_words, &ctx->st_nmask, &ctx->st_bad};
    for (DevBuf *b : bufs) release(*b);
    if (ctx->peer_stage) (void)hipHostFree(ctx->peer_stage);
    if (ctx->h_run_tax) (void)hipHostFree(ctx->h_run_tax);
    if (ctx->h_run_len) (void)hipHostFree(ctx->h_run_len);
    for (int i = 0; i < bns_ctx::EV_RING; ++i) {
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->back_stream) (void)hipStreamDestroy(ctx->back_stream);
    for (hipEvent_t e : ctx->slice_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->done_ev) if (e) (void)hipEventDestroy(e);
    delete ctx;
}

int bns_set_encoder(bns_ctx *ctx, uint32_t k, const uint16_t *gaps, int canonicalize, int spaced_intended)
{
    if (!ctx) return BNS_ERR_ARG;
    if (k < 1 || k > 32) return fail(ctx, BNS_ERR_ARG, "k must be in [1,32] (u64 k-mers)");
    u32 c = 1;
    ctx->pos[0] = 0;
    bool spaced = false;
    for (u32 i = 0; i + 1 < k; ++i) {
        const u32 g = gaps ? gaps[i] : 0;
        if (g) spaced = true;
        c += g + 1;                                   // Spacer ctor: offsets = gaps + 1 (spacer.h:64)
        if (c > 1024) return fail(ctx, BNS_ERR_ARG, "comb size > 1024 is not supported");
        ctx->pos[i + 1] = (u16)(c - 1);
    }
    // runs of adjacent sampled bases (fast spaced gather when the comb fits a 64-base window)
    ctx->n_runs = 0; ctx->sample_mask = 0;
    if (spaced && c <= 64) {
        u32 i = 0;
        while (i < k) {
            u32 j = i;
            while (j + 1 < k && ctx->pos[j + 1] == ctx->pos[j] + 1) ++j;
            ctx->run_start[ctx->n_runs] = (u8)ctx->pos[i];
            ctx->run_len[ctx->n_runs] = (u8)(j - i + 1);
            ++ctx->n_runs;
            i = j + 1;
        }
        for (u32 q = 0; q < k; ++q) ctx->sample_mask |= 1ULL << (63 - ctx->pos[q]);
    }
    // masks with many runs: the compress network of each 32-base half of the 64-base window (Hacker's Delight, fig. 7-4 "compress":
    // mv[i] = the bits that move right by 2^i in step i)
    ctx->pext_on = 0;
    if (spaced && c <= 64 && ctx->n_runs > 4) {
        u64 m2[2] = {0, 0};
        u32 cnt[2] = {0, 0};
        for (u32 q = 0; q < k; ++q) { const u32 pb = ctx->pos[q]; m2[pb >> 5] |= 3ULL << (62 - 2 * (pb & 31)); ++cnt[pb >> 5]; }
        for (int h = 0; h < 2; ++h) {
            u64 m = m2[h], mk = ~m << 1;
            ctx->pext_mask[h] = m2[h]; ctx->pext_steps[h] = 0; ctx->pext_top[h] = 0xFF;
            if (m == (cnt[h] ? ~0ULL << (64 - 2 * cnt[h]) : 0ULL)) { ctx->pext_top[h] = cnt[h]; std::memset(ctx->pext_mv[h], 0, sizeof(ctx->pext_mv[h])); continue; }
            for (int i = 0; i < 6; ++i) {
                u64 mp = mk ^ (mk << 1);
                mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16; mp ^= mp << 32;
                const u64 mv = mp & m;
                ctx->pext_mv[h][i] = mv;
                if (mv) ctx->pext_steps[h] |= 1u << i;
                m = (m ^ mv) | (mv >> (1u << i));
                mk &= ~mp;
            }
        }
        ctx->pext_n1 = cnt[1];
        ctx->pext_on = 1;
    }
    // longest run of adjacent sampled bases, in KEY coordinates (key base q, 0 = first = most significant, sits at bits
    // [2(k-1-q), 2(k-q))): the stretch neighbouring spaced k-mers share position by position (bns_device.hpp, MinSpec)
    ctx->sp_run_len = 0; ctx->sp_run_shift = 0;
    if (spaced) {
        u32 i = 0;
        while (i < k) {
            u32 j = i;
            while (j + 1 < k && ctx->pos[j + 1] == ctx->pos[j] + 1) ++j;
            if (j - i + 1 > ctx->sp_run_len) { ctx->sp_run_len = j - i + 1; ctx->sp_run_shift = 2u * (k - 1u - j); }
            i = j + 1;
        }
    }
    ctx->k = k; ctx->c = c; ctx->spaced = spaced;
    ctx->canon = canonicalize && !spaced;             // encoder.h:148-150
    ctx->spaced_intended = spaced_intended != 0;
    ctx->enc_set = true;
    ctx->win = 0; ctx->score = 0;                     // a new Spacer starts unwindowed
    return BNS_OK;
}

int bns_set_window(bns_ctx *ctx, uint32_t w, int score)
{
    if (!ctx) return BNS_ERR_ARG;
    if (!ctx->enc_set) return fail(ctx, BNS_ERR_STATE, "configure the encoder first (bns_set_encoder)");
    if (score != BNS_SCORE_LEX && score != BNS_SCORE_ENTROPY_PATH && score != BNS_SCORE_ENTROPY_STRING) return fail(ctx, BNS_ERR_ARG, "unknown score");
    if (score == BNS_SCORE_ENTROPY_STRING && ctx->spaced)
        return fail(ctx, BNS_ERR_ARG, "BNS_SCORE_ENTROPY_STRING is the contiguous-seed string overload (encoder.h:425,434); a spaced seed scores through the path rule");
    if (w > ctx->c) {
        // up to 1024 k-mers per window: one 128-entry LDS image up to 64, beyond that a running minimum over 64-entry
        // segments (position windows) or a queue image in global memory (windows over the emitted stream)
        if (w > 1920u) return fail(ctx, BNS_ERR_ARG, "window of more than 1920 bases is not supported (a wavefront works on 2048-base chunks)");
        if (w - ctx->c + 1 > 1024u) return fail(ctx, BNS_ERR_ARG, "window of more than 1024 k-mers is not supported");
    }
    ctx->win = w; ctx->score = score;
    return BNS_OK;
}

int bns_set_bucket_slots_log2(bns_ctx *ctx, uint32_t log2_slots)
{
    if (!ctx || (log2_slots && (log2_slots < 2 || log2_slots > 40))) return BNS_ERR_ARG;
    ctx->slots_log2_req = log2_slots;
    if (log2_slots) ctx->n_buckets_req = 0;
    return BNS_OK;
}

int bns_set_table_buckets(bns_ctx *ctx, uint64_t n_home_buckets)
{
    if (!ctx || n_home_buckets >= (1ULL << 31) - 8) return BNS_ERR_ARG;
    ctx->n_buckets_req = n_home_buckets;
    if (n_home_buckets) ctx->slots_log2_req = 0;
    return BNS_OK;
}

int bns_set_table_fill(bns_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return BNS_ERR_ARG;
    ctx->fill_req = mode - 1;                                    // 0 = the loader decides, 1 = arrival order, 2 = group-aware
    return BNS_OK;
}

int bns_set_minimizer_identity(bns_ctx *ctx, int bits)
{
    if (!ctx || (bits != 0 && bits != 32 && bits != 52)) return BNS_ERR_ARG;
    ctx->wide_req = bits == 0 ? -1 : (bits == 52 ? 1 : 0);
    return BNS_OK;
}

// Where the khash arrays are read from while a bucket table is built: resident on the device (one "chunk"), or on the host,
// streamed through a device staging buffer 2^27 slots at a time -- for dbs whose arrays and table do not fit the HBM together
// (8e9 keys: 210 GB of arrays + a 230 GB table).  The fill / overflow kernels see one chunk at a time.
namespace {
struct KhHost { const u32 *flags = nullptr; const u64 *keys = nullptr; const u32 *vals = nullptr; };
inline u64 stream_chunk(const bns_ctx *ctx)                     // slots per streamed chunk (tests run many small ones: debug bits 16-20)
{
    const int l = (ctx->dbg & BNS_DBG_STREAM_CHUNK_MASK) >> BNS_DBG_STREAM_CHUNK_SHIFT;
    return l >= 4 ? 1ULL << l : 1ULL << 27;
}
// a clustered table is filled to this fraction of its 10 keys per bucket unless the caller fixes its size: the knee of the
// load sweep (profiles/): kernel time within 2 % of a table eight times the size, a quarter of the memory of round 2's default
constexpr double MINB_TARGET_LOAD = 1.0 / 12.0;
}
static int load_table_impl(bns_ctx *ctx, uint64_t n_buckets, const uint32_t *d_flags, const uint64_t *d_keys,
                           const uint32_t *d_vals, const KhHost &host, int layout, void *stream);

int bns_load_table_device(bns_ctx *ctx, uint64_t n_buckets, const uint32_t *d_flags, const uint64_t *d_keys,
                          const uint32_t *d_vals, int layout, void *stream)
{
    if (!ctx || !d_flags || !d_keys || !d_vals) return BNS_ERR_ARG;
    return load_table_impl(ctx, n_buckets, d_flags, d_keys, d_vals, KhHost{}, layout, stream);
}

static int load_table_impl(bns_ctx *ctx, uint64_t n_buckets, const uint32_t *d_flags, const uint64_t *d_keys,
                           const uint32_t *d_vals, const KhHost &host, int layout, void *stream)
{
    const bool streamed = host.flags != nullptr;                 // (then the d_ pointers are null)
    if (n_buckets == 0 || (n_buckets & (n_buckets - 1))) return fail(ctx, BNS_ERR_TABLE, "n_buckets must be a power of two");
    if (layout != BNS_LAYOUT_KHASH && layout != BNS_LAYOUT_BUCKET && layout != BNS_LAYOUT_MINBUCKET) return BNS_ERR_ARG;
    if (layout == BNS_LAYOUT_MINBUCKET && !ctx->enc_set)
        return fail(ctx, BNS_ERR_STATE, "BNS_LAYOUT_MINBUCKET needs the encoder (k) configured before the table is loaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // keep caller-owned arrays alive across free_table when they are the same pointers
    const bool same = !streamed && (d_flags == ctx->kflags);
    if (!same) free_table(ctx);
    else {
        if (ctx->slots) (void)hipFree(ctx->slots);
        if (ctx->ovf_slots) (void)hipFree(ctx->ovf_slots);
        ctx->slots = nullptr; ctx->n_slots = 0; ctx->n_mb = 0; ctx->ovf_slots = nullptr; ctx->n_ovf_slots = 0; ctx->n_ovf_keys = 0;
    }
    ctx->warn.clear();

    if (layout == BNS_LAYOUT_KHASH) {
        if (streamed) return fail(ctx, BNS_ERR_ARG, "BNS_LAYOUT_KHASH probes the arrays themselves: they must be resident");
        ctx->kflags = d_flags; ctx->kkeys = d_keys; ctx->kvals = d_vals; ctx->kh_nb = n_buckets;
        ctx->layout = BNS_LAYOUT_KHASH;
        ctx->n_keys = 0;                              // unknown without a scan; bns_table_info reports 0
        return BNS_OK;
    }
    SmallLayout *sm = (SmallLayout *)ctx->small.p;
    unsigned long long *d_cnt = sm->load_cnt;
    // every pass over the khash arrays goes through here: fn(flags, keys, vals, n) once for resident arrays, once per chunk
    // (uploaded into the staging buffers, stream-ordered behind the previous chunk's kernel) for streamed ones.  flags_only: the
    // pass reads nothing else (the key count), so nothing else crosses PCIe; max_chunks: stop after that many (window sampling).
    u32 *sf = nullptr; u64 *sk = nullptr; u32 *sv = nullptr;
    struct Stage { u32 *&f; u64 *&k; u32 *&v; ~Stage() { if (f) (void)hipFree(f); if (k) (void)hipFree(k); if (v) (void)hipFree(v); } } stage{sf, sk, sv};
    const u64 STREAM_CHUNK = stream_chunk(ctx);
    if (streamed) {
        const u64 cn = std::min<u64>(n_buckets, STREAM_CHUNK);
        HIPCHK(ctx, hipMalloc((void **)&sf, std::max<u64>(1, cn >> 4) * 4));
        HIPCHK(ctx, hipMalloc((void **)&sk, cn * 8));
        HIPCHK(ctx, hipMalloc((void **)&sv, cn * 4));
    }
    auto for_chunks = [&](auto fn, bool flags_only = false, u64 max_chunks = ~0ULL, bool no_vals = false) -> int {
        if (!streamed) { fn(d_flags, d_keys, d_vals, (u64)n_buckets); return BNS_OK; }
        u64 done = 0;
        for (u64 o = 0; o < n_buckets && done < max_chunks; o += STREAM_CHUNK, ++done) {
            const u64 cn = std::min<u64>(STREAM_CHUNK, n_buckets - o);
            HIPCHK(ctx, hipMemcpyAsync(sf, host.flags + (o >> 4), std::max<u64>(1, cn >> 4) * 4, hipMemcpyHostToDevice, st));
            if (!flags_only) {
                HIPCHK(ctx, hipMemcpyAsync(sk, host.keys + o, cn * 8, hipMemcpyHostToDevice, st));
                if (!no_vals) HIPCHK(ctx, hipMemcpyAsync(sv, host.vals + o, cn * 4, hipMemcpyHostToDevice, st));
            }
            fn((const u32 *)sf, (const u64 *)sk, (const u32 *)sv, cn);
            HIPCHK(ctx, hipGetLastError());
        }
        return BNS_OK;
    };
#define BNS_RC(x) do { const int rc__ = (x); if (rc__ != BNS_OK) return rc__; } while (0)
    // present keys (what kh_size would say): the table is sized from them, not from the khash bucket count (a khash is
    // anywhere between 38 % and 77 % full)
    unsigned long long n_present = 0;
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(sm->load_cnt), st));
    BNS_RC(for_chunks([&](const u32 *cf, const u64 *, const u32 *, u64 cn) {
        hipLaunchKernelGGL(count_present_kernel, dim3(grid_for(ctx, std::max<u64>(1, cn >> 4), 256)), dim3(256), 0, st, cf, cn, d_cnt);
    }, true));
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(&n_present, d_cnt, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));

    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
    Slot *slots = nullptr;
    Slot *ovf = nullptr;
    // tens of GB each: released on every early return below (HIPCHK returns from the function), kept on success
    struct Release { Slot *&a; Slot *&b; bool keep = false; ~Release() { if (!keep) { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } } release{slots, ovf};
    u64 n_slots = 0, n_mb = 0, n_ovf_slots = 0, n_ovf_keys = 0;
    MinSpec table_spec{ctx->k, ctx->k, 0u, 1u, 0u};

    if (layout == BNS_LAYOUT_BUCKET) {
        // plain bucket layout: a power of two of 16-byte slots, 2x the khash bucket count by default; capacity must cover even a
        // khash with every slot present, or the fill could never terminate
        u32 lg = 0;
        while ((1ULL << lg) < n_buckets) ++lg;
        u32 want = ctx->slots_log2_req ? ctx->slots_log2_req : lg + 1;
        if (want < 4) want = 4;
        if (((size_t)16 << want) > free_b) return fail(ctx, BNS_ERR_NOMEM, "bucket table does not fit in free HBM");
        n_slots = 1ULL << want;
        if (n_slots <= n_buckets) return fail(ctx, BNS_ERR_ARG, "bucket_slots_log2 too small for this khash");
        HIPCHK(ctx, hipMalloc((void **)&slots, n_slots * sizeof(Slot)));
        HIPCHK(ctx, hipMemsetAsync(slots, 0, n_slots * sizeof(Slot), st));
        HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(sm->load_cnt), st));
        BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *cv, u64 cn) {
            hipLaunchKernelGGL(rebucket_kernel, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cv, cn, slots, n_slots / 4 - 1, d_cnt);
        }));
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
    } else {
        // ---- clustered layout: how many buckets.  Automatic: MINB_TARGET_LOAD, or what fits in three quarters of the free HBM
        // (the overflow table and the caller's batches need room too); any count will do, the bucket index is a multiply-high.
        const u64 mem_cap = (u64)(free_b / 4 * 3 / sizeof(MinBucket));
        const u64 hard_cap = (1ULL << 31) - 8;                    // bucket indices are 31-bit
        u64 want = ctx->n_buckets_req ? ctx->n_buckets_req
                 : ctx->slots_log2_req ? std::max<u64>(1, (1ULL << ctx->slots_log2_req) / 8)
                 : std::max<u64>(64, (u64)((double)n_present / (MINB_CAP * MINB_TARGET_LOAD)));
        if (!ctx->n_buckets_req && !ctx->slots_log2_req) want = std::min(want, mem_cap);
        if (want > hard_cap) {
            if (ctx->n_buckets_req || ctx->slots_log2_req) return fail(ctx, BNS_ERR_ARG, "more than 2^31 buckets: bucket indices are 31-bit");
            want = hard_cap;
        }
        n_mb = want;
        const u64 n_alloc = n_mb + MINB_MAX_CHAIN - 1;             // spill-only buckets behind the last home: chains never wrap
        if (n_alloc * sizeof(MinBucket) > free_b) return fail(ctx, BNS_ERR_NOMEM, "bucket table does not fit in free HBM");
        if ((double)n_mb * MINB_CAP * 0.97 < (double)n_present) return fail(ctx, BNS_ERR_TABLE, "bucket table too small for the key count");
        n_slots = n_alloc * 8;
        HIPCHK(ctx, hipMalloc((void **)&slots, n_alloc * sizeof(MinBucket)));
        HIPCHK(ctx, hipMemsetAsync(slots, 0, n_alloc * sizeof(MinBucket), st));
        MinBucket *mb = reinterpret_cast<MinBucket *>(slots);
        // ---- where the minimizer comes from.  Contiguous seeds: canonical m-mers of the whole key.  Spaced seeds: plain m-mers
        // inside the mask's longest run of adjacent sampled bases, if there is one long enough -- m is then the smallest length
        // with 4^m >= the number of keys (so that minimizer groups rarely share a value: groups are 2-3 keys, a bucket holds 10),
        // at least run - 8 (a group must fit a bucket) and at most run - 1 (a window of two m-mers is the least that lets
        // neighbours share); otherwise m = k: every k-mer its own bucket.
        std::vector<MinSpec> cands;
        std::vector<u32> cand_span;
        if (ctx->spaced) {
            MinSpec ms{ctx->k, ctx->k, 0u, 1u, 0u};
            const u32 R = ctx->sp_run_len;
            if (R >= 12 && !(ctx->dbg & BNS_DBG_SPACED_NOCLUSTER)) {
                u32 m = 11;
                while (m < 32 && (1ULL << (2 * m)) < n_present) ++m;
                if (const int force = (ctx->dbg & BNS_DBG_SPACED_M_MASK) >> BNS_DBG_SPACED_M_SHIFT) m = (u32)std::max(8, force);   // profiling aid
                const bool forced = ((ctx->dbg & BNS_DBG_SPACED_M_MASK) >> BNS_DBG_SPACED_M_SHIFT) != 0;
                if (m + 8 < R) m = R - 8;
                // one base shorter is tried first: a window of one more m-mer means a third fewer bucket fetches per run, and a kernel
                // bound by its fetches gains from that as long as the groups still fit -- accepted below when fewer than 5 keys in
                // 100 miss their home bucket (configs[2]'s 7.8e8-key db: m = 14 instead of 15, 4.4 % outside, 14.28 -> 13.45 ms; a
                // 2.3e8-key db: m = 13 would leave 19 % outside, 12.1 -> 16.6 ms, and is refused)
                if (!forced && m >= 9 && m + 7 >= R && m <= R) { cands.push_back(MinSpec{m - 1, R, ctx->sp_run_shift, 0u, 0u}); cand_span.push_back(0u); }
                if (m + 1 <= R) ms = MinSpec{m, R, ctx->sp_run_shift, 0u, 0u};
            }
            cands.push_back(ms); cand_span.push_back(0u);
        } else {
            // Contiguous seeds: the widest minimizer window whose groups still fit their buckets (MIN_CANDS: k - m = 15, 11, 8; a
            // wider window means fewer bucket fetches per read -- 16 instead of 25 per 150-bp read -- but groups of up to k - m + 1
            // keys in buckets of 10).  A candidate is taken when fewer than 1 key in 100 misses its home bucket when the table is
            // filled with it (tools/span_calib.sh); the last one always is.  A db of every k-mer fails the wide windows at once and
            // ends at 8; a db of window minimizers (bonsai build -w 50: one k-mer in ten) takes 15.  bns_set_minimizer_span() fixes
            // the window, bns_set_minimizer_identity() the identity (default: wide from WIDE_MIN_KEYS keys on).
            // the identity: fixed by bns_set_minimizer_identity(), else both forms go into the trial (narrow ones first)
            for (u32 wide = 0; wide < 2; ++wide) {
                if (ctx->wide_req >= 0 && (u32)ctx->wide_req != wide) continue;
                for (int ci = 0; ci < 3; ++ci) {
                    const MinCand cand = MIN_CANDS[ci];
                    if (ctx->min_span_req && cand.span != ctx->min_span_req) continue;
                    const u32 m = minimizer_len(ctx->k, cand);
                    if (wide && m >= ctx->k) continue;           // (no window, nothing to carry through it)
                    if (!cands.empty() && cands.back().m == m && cands.back().wide == wide) continue;
                    cands.push_back(MinSpec{m, ctx->k, 0u, 1u, wide}); cand_span.push_back(cand.span);
                }
            }
            // (the wide identity asked for, but k so small that there is no window to carry it through: the narrow form it is)
            if (cands.empty()) { cands.push_back(MinSpec{minimizer_len(ctx->k, MIN_CANDS[2]), ctx->k, 0u, 1u, 0u}); cand_span.push_back(MIN_CANDS[2].span); }
        }
        // ---- choose.  One pass over the khash arrays tries all candidates at once on a sample of the BUCKETS (every bucket for a
        // small table, one in 2^j for a large one: whole minimizer groups, at the table's own load) and counts the keys that miss
        // their home bucket (minbucket_trial_kernel); then ONE fill with the winner.  Streamed arrays cross PCIe twice (flags and
        // keys for the trial, everything for the fill) whatever the number of candidates; round 2 filled the whole table once per
        // candidate.
        unsigned long long h2[5] = {0, 0, 0, 0, 0};
        size_t pick = cands.size() - 1;
        double trial_miss = 0.0;                                   // the chosen candidate's share of keys outside their home bucket
        if (cands.size() > 1) {
            u32 sample = (u32)n_mb;
            while (sample > (1u << 22)) sample >>= 1;               // at most 4 M sampled buckets: plenty of groups
            const size_t img = (size_t)(sample + MINB_MAX_CHAIN);
            u32 *d_img = nullptr;
            struct Img { u32 *&p; ~Img() { if (p) (void)hipFree(p); } } img_guard{d_img};
            HIPCHK(ctx, hipMalloc((void **)&d_img, cands.size() * img * sizeof(u32)));
            HIPCHK(ctx, hipMemsetAsync(d_img, 0, cands.size() * img * sizeof(u32), st));
            unsigned long long *d_trial = nullptr;
            struct Tr { unsigned long long *&p; ~Tr() { if (p) (void)hipFree(p); } } tr_guard{d_trial};
            HIPCHK(ctx, hipMalloc((void **)&d_trial, 18 * sizeof(unsigned long long)));
            HIPCHK(ctx, hipMemsetAsync(d_trial, 0, 18 * sizeof(unsigned long long), st));
            TrialCands tc;
            tc.n = (u32)cands.size();
            for (size_t ci = 0; ci < cands.size(); ++ci) tc.c[ci] = cands[ci];
            BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *, u64 cn) {
                hipLaunchKernelGGL(minbucket_trial_kernel, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cn, (u32)n_mb, sample, d_img, d_trial, ctx->k, tc);
            }, false, ~0ULL, true));
            HIPCHK(ctx, hipGetLastError());
            unsigned long long tr[18] = {0};
            HIPCHK(ctx, hipMemcpyAsync(tr, d_trial, sizeof(tr), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            auto missfrac = [&](size_t ci) { return tr[3 * ci] ? (double)(tr[3 * ci + 1] + tr[3 * ci + 2]) / (double)tr[3 * ci] : 0.0; };
            // per identity: the widest window with fewer than 1 key in 100 outside its home bucket, else the narrowest
            size_t best[2] = {cands.size(), cands.size()};
            for (u32 wide = 0; wide < 2; ++wide) {
                size_t last = cands.size();
                for (size_t ci = 0; ci < cands.size(); ++ci) {
                    if (cands[ci].wide != wide) continue;
                    last = ci;
                    if (missfrac(ci) < 0.01) break;
                }
                best[wide] = last;
            }
            // the wide identity costs ~7 % of kernel time (a 64-bit ring, v_min_f64 per window entry): worth it when the narrow
            // table leaves more than 2 keys in 100 outside their home bucket and the wide one at most 60 % of that (calibrated:
            // a 9e8-key db of window minimizers 0.4 % -> narrow, 6.49 vs 6.94 ms; every-k-mer dbs of 9e8 / 1.8e9 keys 6.9 % /
            // 12 % -> wide, 8.05 vs 8.26 and 8.44 vs 9.26 ms)
            pick = best[0] < cands.size() ? best[0] : best[1];
            if (best[0] < cands.size() && best[1] < cands.size() && missfrac(best[0]) >= 0.02 && missfrac(best[1]) <= 0.6 * missfrac(best[0])) pick = best[1];
            if (ctx->spaced) pick = missfrac(0) < 0.05 ? 0 : cands.size() - 1;       // (two candidates: the longer window when its groups fit)
            trial_miss = missfrac(pick);
        }
        table_spec = cands[pick];
        ctx->table_span = cand_span[pick];
        // ---- fill.  A table with room takes its keys in arrival order; a crowded one -- the trial left more than 1 key in 100
        // outside its home bucket, or (no trial: one candidate) the table is more than an eighth full -- is filled group by group
        // (minbucket_tagcount_kernel: count, decide, residents, the rest): two more passes over the arrays, and about half as many
        // of a read's runs need a second probe pass.  BNS_DBG_PLAIN_FILL / BNS_DBG_GROUP_FILL force one or the other (tests, A/B).
        const bool crowded = cands.size() > 1 ? trial_miss >= 0.01 : (double)n_present > 0.125 * (double)n_mb * MINB_CAP;
        const bool group_fill = ctx->fill_req >= 0 ? ctx->fill_req == 1 : (!(ctx->dbg & BNS_DBG_PLAIN_FILL) && (crowded || (ctx->dbg & BNS_DBG_GROUP_FILL)));
        ctx->group_fill = group_fill;
        {
            HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, sizeof(sm->load_cnt), st));   // [0] present, [1] chain exhausted, [2] moved by place, [3] error, [4] spilled, [5] [6] decide
            if (group_fill) {
                BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *, u64 cn) {
                    hipLaunchKernelGGL(minbucket_tagcount_kernel, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cn, mb, (u32)n_mb, ctx->k, table_spec);
                }, false, ~0ULL, true));
                hipLaunchKernelGGL(minbucket_decide_kernel, dim3(grid_for(ctx, n_mb, 256)), dim3(256), 0, st, mb, n_mb, d_cnt + 5);
                BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *cv, u64 cn) {
                    hipLaunchKernelGGL(minbucket_fill_kernel<1>, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cv, cn, mb, (u32)n_mb, d_cnt, ctx->k, table_spec);
                }));
                BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *cv, u64 cn) {
                    hipLaunchKernelGGL(minbucket_fill_kernel<2>, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cv, cn, mb, (u32)n_mb, d_cnt, ctx->k, table_spec);
                }));
            } else
            BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *cv, u64 cn) {
                hipLaunchKernelGGL(minbucket_fill_kernel<0>, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cv, cn, mb, (u32)n_mb, d_cnt, ctx->k, table_spec);
            }));
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(h2, d_cnt, 40, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
        }
        ctx->n_spilled = h2[4];
        if (h2[0] && (h2[4] + h2[1]) * 100ULL >= h2[0]) {
            char buf[256];
            std::snprintf(buf, sizeof(buf), "%llu of %llu keys (%.1f %%) are not in their home bucket (minimizer window %u, %.0f %% load): lookups of "
                          "this table take extra probe passes; a larger table (bns_set_table_buckets) or a narrower window helps",
                          (unsigned long long)(h2[4] + h2[1]), (unsigned long long)h2[0], 100.0 * (double)(h2[4] + h2[1]) / (double)h2[0],
                          ctx->k - table_spec.m, 100.0 * (double)h2[0] / ((double)n_mb * MINB_CAP));
            ctx->warn = buf;
        }
        n_ovf_keys = h2[1];
        n_ovf_slots = 64;
        // at most half full (+2048: room for the buckets minbucket_place_kernel may move here -- about one in 1e8; under the test
        // switch that fails every 61st bucket, room for all of those)
        const u64 place_room = 2048 + ((ctx->dbg & BNS_DBG_PLACE_FAIL) ? n_alloc / 61 * MINB_CAP + MINB_CAP : 0);
        while (n_ovf_slots < 2 * (n_ovf_keys + place_room)) n_ovf_slots <<= 1;
        HIPCHK(ctx, hipMalloc((void **)&ovf, n_ovf_slots * sizeof(Slot)));
        HIPCHK(ctx, hipMemsetAsync(ovf, 0, n_ovf_slots * sizeof(Slot), st));
        u32 *d_err = reinterpret_cast<u32 *>(d_cnt + 3);
        if (n_ovf_keys)
            BNS_RC(for_chunks([&](const u32 *cf, const u64 *ck, const u32 *cv, u64 cn) {
                hipLaunchKernelGGL(minbucket_overflow_kernel, dim3(grid_for(ctx, cn, 256)), dim3(256), 0, st, cf, ck, cv, cn,
                                   (const MinBucket *)mb, (u32)n_mb, ovf, n_ovf_slots / 4 - 1, ctx->k, table_spec, d_err);
            }));
        hipLaunchKernelGGL(minbucket_place_kernel, dim3(grid_for(ctx, n_alloc, 4)), dim3(256), 0, st, mb, n_alloc, (u32)n_mb, ovf, n_ovf_slots / 4 - 1, d_cnt + 2, d_err,
                           (u32)((ctx->dbg & BNS_DBG_PLACE_FAIL) ? 61 : 0), ctx->k, table_spec);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(h2, d_cnt, 32, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        n_ovf_keys += h2[2];
        if (h2[3]) return fail(ctx, BNS_ERR_TABLE, "overflow table full while placing bucket keys");
    }
    unsigned long long h_cnt = 0;
    HIPCHK(ctx, hipMemcpyAsync(&h_cnt, d_cnt, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (layout == BNS_LAYOUT_BUCKET && h_cnt >= n_slots) return fail(ctx, BNS_ERR_TABLE, "bucket table too small for the key count");
    release.keep = true;
    ctx->slots = slots; ctx->n_slots = n_slots; ctx->n_mb = (u32)n_mb; ctx->n_keys = h_cnt;
    ctx->ovf_slots = ovf; ctx->n_ovf_slots = n_ovf_slots; ctx->n_ovf_keys = n_ovf_keys;
    ctx->layout = layout; ctx->table_k = ctx->k;
    ctx->table_m = table_spec.m; ctx->table_len = table_spec.len; ctx->table_shift = table_spec.shift; ctx->table_canon = table_spec.canon;
    ctx->table_wide = table_spec.wide != 0;
    if (same && ctx->own_khash) {                     // host-upload path: the khash copy is no longer needed
        (void)hipFree((void *)ctx->kflags); (void)hipFree((void *)ctx->kkeys); (void)hipFree((void *)ctx->kvals);
        ctx->own_khash = false;
    }
    ctx->kflags = nullptr; ctx->kkeys = nullptr; ctx->kvals = nullptr; ctx->kh_nb = 0;
    return BNS_OK;
}

int bns_load_table(bns_ctx *ctx, uint64_t n_buckets, const uint32_t *flags, const uint64_t *keys, const uint32_t *vals, int layout)
{
    if (!ctx || !flags || !keys || !vals) return BNS_ERR_ARG;
    if (n_buckets == 0 || (n_buckets & (n_buckets - 1))) return fail(ctx, BNS_ERR_TABLE, "n_buckets must be a power of two");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    free_table(ctx);
    const size_t fs = n_buckets < 16 ? 1 : (size_t)(n_buckets >> 4);
    if (layout != BNS_LAYOUT_KHASH) {
        // a db whose arrays and a useful table (16 bytes per khash bucket) do not fit the HBM together is streamed from the host
        // buffers instead of being uploaded whole (BNS_DBG_STREAM_LOAD forces that path: tests)
        size_t free_b = 0, total_b = 0;
        HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
        const size_t arrays = fs * 4 + (size_t)n_buckets * 12;
        if ((ctx->dbg & BNS_DBG_STREAM_LOAD) || arrays + (size_t)n_buckets * 16 > free_b / 100 * 85)
            return load_table_impl(ctx, n_buckets, nullptr, nullptr, nullptr, KhHost{flags, keys, vals}, layout, ctx->stream);
    }
    u32 *df = nullptr; u64 *dk = nullptr; u32 *dv = nullptr;
    // each pointer goes into the context as soon as it exists, so free_table() releases it whatever fails next
    ctx->own_khash = true; ctx->kh_nb = n_buckets;
    HIPCHK(ctx, hipMalloc((void **)&df, fs * 4));
    ctx->kflags = df;
    HIPCHK(ctx, hipMalloc((void **)&dk, (size_t)n_buckets * 8));
    ctx->kkeys = dk;
    HIPCHK(ctx, hipMalloc((void **)&dv, (size_t)n_buckets * 4));
    ctx->kvals = dv;
    HIPCHK(ctx, hipMemcpyAsync(df, flags, fs * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dk, keys, (size_t)n_buckets * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dv, vals, (size_t)n_buckets * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int rc = bns_load_table_device(ctx, n_buckets, df, dk, dv, layout, ctx->stream);
    if (rc == BNS_OK && layout == BNS_LAYOUT_KHASH) ctx->own_khash = true;
    return rc;
}

// ---- multi-GPU table load: one upload, RCCL broadcast over xGMI, one device-side re-hash per GPU -------------------------
// librccl is opened lazily and privately (RTLD_LOCAL) the first time a broadcast is wanted: a single-GPU process never maps it,
// and a host that already carries its own RCCL (PyTorch ships one) does not get a second set of nccl* symbols in its global
// namespace.  The entry points' types come from <rccl/rccl.h> itself (decltype of the declarations): a signature or enum that
// drifts is a compile error here, not a silent ABI mismatch behind dlsym.
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    int version = 0, last_ranks = 0;            // what bns_rccl_info reports: the library's version code, the ranks of the last broadcast
    std::once_flag once;
    std::string open_err;
    bool ok = false;
    bool open(std::string &err)
    {
        std::call_once(once, [this] {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
            if (!lib) { const char *de = dlerror(); open_err = std::string("cannot open librccl: ") + (de ? de : "?"); return; }
            auto sym = [&](const char *n) { void *q = dlsym(lib, n); if (!q && open_err.empty()) open_err = std::string("librccl lacks ") + n; return q; };
            CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
            CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
            GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
            GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
            Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
            GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
            GetVersion = (decltype(GetVersion))dlsym(lib, "ncclGetVersion");
            if (GetVersion) (void)GetVersion(&version);
            ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && GetErrorString;
        });
        if (!ok) err = open_err;
        return ok;
    }
};
Rccl g_rccl;

// One communicator over the contexts' devices (a single process driving several devices: every rank's call inside one group),
// one grouped ncclBroadcast per array from rank 0.  n_ctx == 1 is legal (a one-rank communicator: how a one-GPU box executes the
// real RCCL calls); duplicate devices are not (RCCL admits a device once per communicator).
int rccl_broadcast(bns_ctx **ctxs, int n_ctx, const std::vector<std::array<void *, 3>> &dev, const size_t bytes[3], std::string &err)
{
    if (!g_rccl.open(err)) return BNS_ERR_HIP;
    std::vector<int> devs((size_t)n_ctx);
    for (int i = 0; i < n_ctx; ++i) devs[(size_t)i] = ctxs[i]->device;
    std::vector<ncclComm_t> comms((size_t)n_ctx, nullptr);
    ncclResult_t rc = g_rccl.CommInitAll(comms.data(), n_ctx, devs.data());
    if (rc != ncclSuccess) { err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc); return BNS_ERR_HIP; }
    bool hip_fail = false;
    for (int a = 0; a < 3 && rc == ncclSuccess && !hip_fail; ++a) {
        rc = g_rccl.GroupStart();
        for (int i = 0; i < n_ctx && rc == ncclSuccess; ++i) {
            if (hipSetDevice(ctxs[i]->device) != hipSuccess) { hip_fail = true; break; }
            // (every rank passes its OWN buffer as send and receive buffer: in place on the root, ignored as a source elsewhere)
            rc = g_rccl.Broadcast(dev[(size_t)i][a], dev[(size_t)i][a], bytes[a], ncclUint8, 0, comms[(size_t)i], ctxs[i]->stream);
        }
        const ncclResult_t rc2 = g_rccl.GroupEnd();
        if (rc == ncclSuccess) rc = rc2;
    }
    for (int i = 0; i < n_ctx; ++i) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamSynchronize(ctxs[i]->stream); }
    for (int i = 0; i < n_ctx; ++i) if (comms[(size_t)i]) (void)g_rccl.CommDestroy(comms[(size_t)i]);
    if (rc == ncclSuccess && !hip_fail) g_rccl.last_ranks = n_ctx;
    if (rc != ncclSuccess || hip_fail) {
        err = std::string("RCCL broadcast of the table: ") + (hip_fail ? "hipSetDevice failed" : g_rccl.GetErrorString(rc));
        return BNS_ERR_HIP;
    }
    return BNS_OK;
}

// HIPCHK for the multi-context loader: remembers WHICH context failed, so the wrapper can hand its message to the root
#define HIPCHK_M(c, expr) do { failed = (c); HIPCHK((c), expr); failed = nullptr; } while (0)

int load_table_multi_impl(bns_ctx **ctxs, int n_ctx, uint64_t n_buckets, const uint32_t *flags, const uint64_t *keys,
                          const uint32_t *vals, int layout, bns_ctx *&failed)
{
    bns_ctx *root = ctxs[0];
    const size_t fs = n_buckets < 16 ? 1 : (size_t)(n_buckets >> 4);
    const size_t bytes[3] = {fs * 4, (size_t)n_buckets * 8, (size_t)n_buckets * 4};
    const void *host[3] = {flags, keys, vals};
    // 0. a db whose arrays do not fit the HBM next to its table cannot be replicated array by array: every context streams the
    //    host buffers into its own table (bns_load_table), the first alone (its table size and minimizer window are everyone's),
    //    the others side by side -- each over its own PCIe link
    if (layout != BNS_LAYOUT_KHASH) {
        size_t free_b = 0, total_b = 0;
        HIPCHK_M(root, hipSetDevice(root->device));
        HIPCHK_M(root, hipMemGetInfo(&free_b, &total_b));
        if ((root->dbg & BNS_DBG_STREAM_LOAD) || bytes[0] + bytes[1] + bytes[2] + (size_t)n_buckets * 16 > free_b / 100 * 85) {
            failed = root;
            int rc = bns_load_table(root, n_buckets, flags, keys, vals, layout);
            if (rc != BNS_OK) return rc;
            failed = nullptr;
            const u64 buckets_used = layout == BNS_LAYOUT_MINBUCKET ? root->n_mb : 0;
            u32 lg_used = 0;
            if (layout == BNS_LAYOUT_BUCKET) while ((1ULL << lg_used) < root->n_slots) ++lg_used;
            const u32 span_used = layout == BNS_LAYOUT_MINBUCKET ? root->table_span : 0u;
            const int wide_used = layout == BNS_LAYOUT_MINBUCKET ? (root->table_wide ? 1 : 0) : -1;      // the root's minimizer identity
            std::vector<int> rcs((size_t)n_ctx, BNS_OK);
            std::vector<std::thread> th;
            for (int i = 1; i < n_ctx; ++i)
                th.emplace_back([&, i] {
                    bns_ctx *c = ctxs[i];
                    const u64 saved = c->n_buckets_req; const u32 saved_span = c->min_span_req, saved_lg = c->slots_log2_req; const int saved_dbg = c->dbg;
                    const int saved_wide = c->wide_req, saved_fill = c->fill_req;
                    if (layout == BNS_LAYOUT_MINBUCKET) c->fill_req = root->group_fill ? 1 : 0;
                    if (buckets_used) { c->n_buckets_req = buckets_used; c->slots_log2_req = 0; }
                    if (lg_used) c->slots_log2_req = lg_used;
                    // the root's window AND identity: one candidate left, so its trial pass (flags + keys over PCIe once more) is not
                    // repeated and every replica is the same table
                    if (span_used) c->min_span_req = span_used;
                    if (span_used && wide_used >= 0) c->wide_req = wide_used;
                    c->dbg |= root->dbg & (BNS_DBG_STREAM_LOAD | BNS_DBG_STREAM_CHUNK_MASK);
                    rcs[(size_t)i] = bns_load_table(c, n_buckets, flags, keys, vals, layout);
                    c->n_buckets_req = saved; c->min_span_req = saved_span; c->slots_log2_req = saved_lg; c->dbg = saved_dbg; c->wide_req = saved_wide;
                    c->fill_req = saved_fill;
                });
            for (auto &t : th) t.join();
            for (int i = 1; i < n_ctx; ++i) if (rcs[(size_t)i] != BNS_OK) { failed = ctxs[i]; return rcs[(size_t)i]; }
            return BNS_OK;
        }
    }
    // 1. device copies of the khash arrays on every context's device (owned by the contexts: free_table releases them)
    std::vector<std::array<void *, 3>> dev((size_t)n_ctx, std::array<void *, 3>{nullptr, nullptr, nullptr});
    for (int i = 0; i < n_ctx; ++i) {
        bns_ctx *c = ctxs[i];
        HIPCHK_M(c, hipSetDevice(c->device));
        free_table(c);
        c->own_khash = true; c->kh_nb = n_buckets;
        HIPCHK_M(c, hipMalloc(&dev[i][0], bytes[0])); c->kflags = (const u32 *)dev[i][0];
        HIPCHK_M(c, hipMalloc(&dev[i][1], bytes[1])); c->kkeys = (const u64 *)dev[i][1];
        HIPCHK_M(c, hipMalloc(&dev[i][2], bytes[2])); c->kvals = (const u32 *)dev[i][2];
    }
    // 2. ONE host-to-device upload (root), then device-to-device replication
    HIPCHK_M(root, hipSetDevice(root->device));
    for (int a = 0; a < 3; ++a) HIPCHK_M(root, hipMemcpyAsync(dev[0][a], host[a], bytes[a], hipMemcpyHostToDevice, root->stream));
    HIPCHK_M(root, hipStreamSynchronize(root->stream));
    bool distinct = true;
    for (int i = 0; i < n_ctx; ++i) for (int j = 0; j < i; ++j) if (ctxs[i]->device == ctxs[j]->device) distinct = false;
    if (distinct) {
        std::string err;
        const int rc = rccl_broadcast(ctxs, n_ctx, dev, bytes, err);       // RCCL over xGMI
        if (rc != BNS_OK) { failed = root; return fail(root, rc, err); }
    } else {
        // several contexts on ONE device (how a one-GPU box exercises the shard / stitch logic; RCCL refuses duplicate devices):
        // plain copies
        for (int i = 1; i < n_ctx; ++i) {
            HIPCHK_M(ctxs[i], hipSetDevice(ctxs[i]->device));
            for (int a = 0; a < 3; ++a) HIPCHK_M(ctxs[i], hipMemcpyAsync(dev[(size_t)i][a], dev[0][a], bytes[a], hipMemcpyDeviceToDevice, ctxs[i]->stream));
            HIPCHK_M(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
        }
    }
    // 3. every device lays out its own table; one size and one minimizer window for all (the first context's choice), whatever
    //    each finds free
    u64 buckets_used = 0;
    u32 span_used = 0, lg_used = 0;
    int wide_used = -1;
    for (int i = 0; i < n_ctx; ++i) {
        bns_ctx *c = ctxs[i];
        const u64 saved = c->n_buckets_req; const u32 saved_span = c->min_span_req, saved_lg = c->slots_log2_req;
        const int saved_wide = c->wide_req, saved_fill = c->fill_req;
        if (i > 0 && layout == BNS_LAYOUT_MINBUCKET) c->fill_req = ctxs[0]->group_fill ? 1 : 0;
        if (i > 0 && buckets_used) { c->n_buckets_req = buckets_used; c->slots_log2_req = 0; }
        if (i > 0 && lg_used) c->slots_log2_req = lg_used;
        if (i > 0 && span_used) { c->min_span_req = span_used; if (wide_used >= 0) c->wide_req = wide_used; }   // (one candidate: no second trial)
        failed = c;
        const int rc = bns_load_table_device(c, n_buckets, c->kflags, c->kkeys, c->kvals, layout, c->stream);
        c->n_buckets_req = saved; c->min_span_req = saved_span; c->slots_log2_req = saved_lg; c->wide_req = saved_wide; c->fill_req = saved_fill;
        if (rc != BNS_OK) return rc;
        failed = nullptr;
        if (layout == BNS_LAYOUT_KHASH) c->own_khash = true;
        if (i == 0 && layout == BNS_LAYOUT_MINBUCKET) { buckets_used = c->n_mb; span_used = c->table_span; wide_used = c->table_wide ? 1 : 0; }
        if (i == 0 && layout == BNS_LAYOUT_BUCKET) while ((1ULL << lg_used) < c->n_slots) ++lg_used;
    }
    return BNS_OK;
}
}  // namespace

int bns_load_table_multi(bns_ctx **ctxs, int n_ctx, uint64_t n_buckets, const uint32_t *flags, const uint64_t *keys,
                         const uint32_t *vals, int layout)
{
    if (!ctxs || n_ctx < 1 || !flags || !keys || !vals) return BNS_ERR_ARG;
    for (int i = 0; i < n_ctx; ++i) if (!ctxs[i]) return BNS_ERR_ARG;
    bns_ctx *root = ctxs[0];
    // (BNS_DBG_FORCE_RCCL, tests: a single context still takes the replicate path -- a one-rank communicator and real ncclBroadcast
    // calls -- so that a one-GPU box executes the RCCL code)
    if (n_ctx == 1 && !(root->dbg & BNS_DBG_FORCE_RCCL)) return bns_load_table(root, n_buckets, flags, keys, vals, layout);
    if (n_buckets == 0 || (n_buckets & (n_buckets - 1))) return fail(root, BNS_ERR_TABLE, "n_buckets must be a power of two");
    bns_ctx *failed = nullptr;
    const int rc = load_table_multi_impl(ctxs, n_ctx, n_buckets, flags, keys, vals, layout, failed);
    if (rc != BNS_OK) {
        // the caller reads bns_last_error(ctxs[0]); and no context keeps half a replica (raw khash copies next to built tables)
        if (failed && failed != root) root->err = failed->err;
        const std::string msg = root->err;
        for (int i = 0; i < n_ctx; ++i) { (void)hipSetDevice(ctxs[i]->device); free_table(ctxs[i]); }
        root->err = msg;
    }
    return rc;
}

int bns_rccl_info(int *version, int *n_ranks)
{
    if (version) *version = g_rccl.version;
    if (n_ranks) *n_ranks = g_rccl.last_ranks;
    return BNS_OK;
}

int bns_table_info(const bns_ctx *ctx, uint64_t *n_keys, uint64_t *device_bytes, int *layout)
{
    if (!ctx) return BNS_ERR_ARG;
    if (n_keys) *n_keys = ctx->n_keys;
    if (layout) *layout = ctx->layout;
    if (device_bytes) {
        if (ctx->layout == BNS_LAYOUT_BUCKET || ctx->layout == BNS_LAYOUT_MINBUCKET) *device_bytes = (ctx->n_slots + ctx->n_ovf_slots) * sizeof(Slot);
        else if (ctx->layout == BNS_LAYOUT_KHASH) *device_bytes = ctx->kh_nb * 12 + (ctx->kh_nb < 16 ? 4 : ctx->kh_nb / 4);
        else *device_bytes = 0;
    }
    return BNS_OK;
}

int bns_table_stats(const bns_ctx *ctx, uint64_t *stats4)
{
    if (!ctx || !stats4) return BNS_ERR_ARG;
    stats4[0] = ctx->n_keys; stats4[1] = ctx->n_ovf_keys;
    stats4[2] = ctx->layout == BNS_LAYOUT_KHASH ? ctx->kh_nb * 12 + (ctx->kh_nb < 16 ? 4 : ctx->kh_nb / 4) : ctx->n_slots * sizeof(Slot);
    stats4[3] = ctx->n_ovf_slots * sizeof(Slot);
    return BNS_OK;
}

int bns_set_minimizer_span(bns_ctx *ctx, uint32_t span)
{
    if (!ctx) return BNS_ERR_ARG;
    bool ok = span == 0;
    for (const MinCand &c : MIN_CANDS) ok = ok || span == c.span;
    if (!ok) return fail(ctx, BNS_ERR_ARG, "minimizer span must be 0 (chosen from the db), 8, 11 or 15");
    ctx->min_span_req = span;
    return BNS_OK;
}

int bns_table_minimizer(const bns_ctx *ctx, uint32_t *m, uint64_t *spilled_keys)
{
    if (!ctx) return BNS_ERR_ARG;
    if (m) *m = ctx->layout == BNS_LAYOUT_MINBUCKET ? ctx->table_m : 0u;
    if (spilled_keys) *spilled_keys = ctx->layout == BNS_LAYOUT_MINBUCKET ? ctx->n_spilled : 0ULL;
    return BNS_OK;
}

int bns_table_geometry(const bns_ctx *ctx, uint64_t *geo8)
{
    if (!ctx || !geo8) return BNS_ERR_ARG;
    const bool mb = ctx->layout == BNS_LAYOUT_MINBUCKET;
    geo8[0] = mb ? ctx->n_mb : (ctx->layout == BNS_LAYOUT_BUCKET ? ctx->n_slots / 4 : ctx->kh_nb);
    geo8[1] = mb ? ctx->table_m : 0;
    geo8[2] = mb ? (ctx->table_wide ? 52 : 32) : 0;
    geo8[3] = mb ? ctx->n_spilled : 0;
    geo8[4] = mb ? ctx->table_span : 0;
    geo8[5] = mb ? ctx->n_ovf_keys : 0;
    geo8[6] = (mb && ctx->group_fill) ? 1 : 0;
    geo8[7] = 0;
    return BNS_OK;
}

const char *bns_table_warning(const bns_ctx *ctx) { return ctx ? ctx->warn.c_str() : ""; }

// Host-side flattening of the parent map into {parent, Euler interval, flags} records.
int bns_load_taxonomy(bns_ctx *ctx, const uint32_t *parent, uint32_t n)
{
    if (!ctx || !parent || n < 2) return BNS_ERR_ARG;
    if (n > (1u << 28)) return fail(ctx, BNS_ERR_TAX_RANGE, "taxid >= 2^28");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<TaxNode> nodes(n);
    std::vector<u8> inforest(n, 0);
    std::vector<u32> child_cnt(n + 1, 0);
    for (u32 x = 0; x < n; ++x) {
        const u32 p = parent[x];
        nodes[x] = TaxNode{p, 0u, 0u, p != TAX_ABSENT ? NODE_PRESENT : 0u};
        if (p == TAX_ABSENT || x == 0) continue;
        if (p >= n) return fail(ctx, BNS_ERR_ARG, "parent id outside [0,n)");
        inforest[x] = 1;
        if (p != 0) { inforest[p] = 1; ++child_cnt[p]; }
    }
    // CSR of children
    std::vector<u32> child_off(n + 1, 0);
    for (u32 x = 0; x < n; ++x) child_off[x + 1] = child_off[x] + child_cnt[x];
    std::vector<u32> children(child_off[n]);
    {
        std::vector<u32> fillp(child_off.begin(), child_off.end() - 1);
        for (u32 x = 1; x < n; ++x) {
            const u32 p = parent[x];
            if (p != TAX_ABSENT && p != 0) children[fillp[p]++] = x;
        }
    }
    // iterative DFS from every root (forest node that is not a key, or whose parent is 0)
    u32 clock = 0;
    u64 visited = 0, n_forest = 0;
    std::vector<std::pair<u32, u32>> stack;   // (node, next child index)
    for (u32 x = 1; x < n; ++x) n_forest += inforest[x];
    for (u32 root = 1; root < n; ++root) {
        if (!inforest[root]) continue;
        const u32 pr = parent[root];
        if (!(pr == TAX_ABSENT || pr == 0)) continue;
        nodes[root].tin = ++clock;
        if (pr == 0) nodes[root].flags |= NODE_CHAIN_OK;
        ++visited;
        stack.emplace_back(root, child_off[root]);
        while (!stack.empty()) {
            auto &top = stack.back();
            const u32 x = top.first;
            if (top.second < child_off[x + 1]) {
                const u32 ch = children[top.second++];
                nodes[ch].tin = ++clock;
                if (nodes[x].flags & NODE_CHAIN_OK) nodes[ch].flags |= NODE_CHAIN_OK;   // ch is a key by construction
                ++visited;
                stack.emplace_back(ch, child_off[ch]);
            } else {
                nodes[x].tout = ++clock;
                stack.pop_back();
            }
        }
    }
    if (visited != n_forest) return fail(ctx, BNS_ERR_TAX_CYCLE, "parent map has a cycle: the reference's walks would not terminate");
    TaxNode *d = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d, (size_t)n * sizeof(TaxNode)));
    HIPCHK(ctx, hipMemcpy(d, nodes.data(), (size_t)n * sizeof(TaxNode), hipMemcpyHostToDevice));
    if (ctx->nodes) (void)hipFree(ctx->nodes);
    ctx->nodes = d; ctx->n_nodes = n;
    return BNS_OK;
}

int bns_set_timing(bns_ctx *ctx, int enabled)
{
    if (!ctx) return BNS_ERR_ARG;
    ctx->timing = enabled != 0;
    ctx->ev_count = 0;
    return BNS_OK;
}

float bns_last_kernel_ms(const bns_ctx *ctx)
{
    if (!ctx || ctx->ev_count == 0) return -1.0f;
    const int i = (ctx->ev_head + bns_ctx::EV_RING - 1) % bns_ctx::EV_RING;
    float ms = -1.0f;
    if (hipEventSynchronize(ctx->ev1[i]) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0[i], ctx->ev1[i]) != hipSuccess) return -1.0f;
    return ms;
}

int bns_timing_summary(bns_ctx *ctx, double *sum_ms, int *count)
{
    if (!ctx || !sum_ms || !count) return BNS_ERR_ARG;
    double sum = 0.0;
    int n = 0;
    for (int j = 0; j < ctx->ev_count; ++j) {
        const int i = (ctx->ev_head + bns_ctx::EV_RING - 1 - j) % bns_ctx::EV_RING;
        float ms = 0.0f;
        HIPCHK(ctx, hipEventSynchronize(ctx->ev1[i]));
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0[i], ctx->ev1[i]));
        sum += ms; ++n;
    }
    *sum_ms = sum; *count = n;
    ctx->ev_count = 0;
    return BNS_OK;
}

// One batch, inputs resident: ASCII (d_bases) or packed (d_words [+ d_nmask]); exactly one of d_bases / d_words is set.
static int classify_device_impl(bns_ctx *ctx, const char *d_bases, const uint64_t *d_words, const uint32_t *d_nmask, const uint64_t *d_offsets,
                                uint64_t n_reads, uint64_t total_bases, uint32_t max_read_len, int paired, uint32_t *d_taxon,
                                uint32_t *d_missing, uint32_t *d_ambig, uint32_t *d_n_hits, uint32_t *d_hits, void *stream)
{
    int rc = ready(ctx, true, true);
    if (rc != BNS_OK) return rc;
    const bool packed = d_words != nullptr;
    if (!d_offsets || !d_taxon || (!d_bases && !d_words && total_bases)) return BNS_ERR_ARG;
    if (paired && (n_reads & 1)) return fail(ctx, BNS_ERR_ARG, "paired input needs an even number of reads");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const int nm = paired ? 2 : 1;
    const u64 n_units = n_reads / (u64)nm;
    if (n_units == 0) return BNS_OK;
    if (n_units >= (1ULL << 32) - (1ULL << 22)) return fail(ctx, BNS_ERR_ARG, "more than 2^32 - 2^22 units in one batch: split it");   // (headroom: every wavefront claims one chunk past the end)

    SmallLayout *sm = (SmallLayout *)ctx->small.p;
    u32 *d_ovf = &sm->ovf_count;
    u32 *d_work = &sm->work_counter;
    HIPCHK(ctx, hipMemsetAsync(sm, 0, SMALL_CLASSIFY_ZERO, st));   // work_counter, ovf_count, max_len in one memset
    if (max_read_len == 0) {
        hipLaunchKernelGGL(max_len_kernel, dim3(grid_for(ctx, n_reads, 256)), dim3(256), 0, st, d_offsets, (u64)n_reads, &sm->max_len);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(&max_read_len, &sm->max_len, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    const u64 max_unit_kmers = max_read_len >= ctx->c ? (u64)nm * (max_read_len - ctx->c + 1) : 0;
    const bool can_overflow = max_unit_kmers > LDS_CAP;
    if (can_overflow && (rc = ensure(ctx, ctx->ovf_list, (size_t)n_units * 8)) != BNS_OK) return rc;

    ClassifyParams p;
    fill_params(ctx, p);
    p.offsets = d_offsets; p.n_units = n_units; p.nmates = nm; p.bases = (const u8 *)d_bases;
    if (packed) { p.words = (const u64 *)d_words; p.nmask = (const u32 *)d_nmask; }
    p.taxon = d_taxon; p.missing = d_missing; p.ambig = d_ambig; p.n_hits = d_n_hits; p.hits = d_hits;
    p.want_hits = d_hits ? 1 : 0;
    if ((rc = ensure(ctx, ctx->records, (size_t)n_units * 16)) != BNS_OK) return rc;
    p.records = (uint4 *)ctx->records.p;
    p.ovf_count = d_ovf; p.ovf_list = can_overflow ? (u64 *)ctx->ovf_list.p : nullptr;
    p.work_counter = d_work;
    // reference behaviour for a spaced seed through the string for_each: nothing is emitted
    p.emit_none = (ctx->spaced && !ctx->spaced_intended) ? 1 : 0;

    // units per claim: the full chunk when that leaves every resident wave (8 blocks of 4 per CU) at least one, else an even share
    // (>= 4: a claim is an atomic plus an offsets load)
    const u64 resident_waves = (u64)ctx->n_cu * 8 * 4;
    const u32 chunk = (u32)std::min<u64>(classify_chunk((u32)nm), std::max<u64>(4, (n_units + resident_waves - 1) / resident_waves));
    p.chunk = chunk;
    unsigned grid = grid_for(ctx, (n_units + chunk - 1) / chunk, 4);
    const int evi = ctx->ev_head;
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev0[evi], st));
    // Contiguous seeds on the clustered table with a common k: k, the mates per unit and the minimizer window are compile-time
    // constants (k = 31 in every form -- either overflow lookup, either minimizer identity; k = 21, 25, 27, 32 in the usual one).
    // (the form of the overflow-table lookup: cooperative when more than 1 key in 1000 lives there, see probe_minbucket)
    const bool ovf_heavy = (ctx->dbg & BNS_DBG_OVC_ON) || (!(ctx->dbg & BNS_DBG_OVC_OFF) && ctx->n_ovf_keys * 1000ULL > ctx->n_keys);
    const bool clustered = !ctx->spaced && ctx->layout == BNS_LAYOUT_MINBUCKET;
    bool launched = false;
    if (clustered) launched = launch_fixed_k(p, grid, st, ovf_heavy, ctx->table_wide, packed);
    if (!launched && clustered && ctx->table_wide) {
        if (packed) hipLaunchKernelGGL((classify_kernel<false, 2, 0, 0, 8, false, true, true>), dim3(grid), dim3(256), 0, st, p);
        else        hipLaunchKernelGGL((classify_kernel<false, 2, 0, 0, 8, false, true>), dim3(grid), dim3(256), 0, st, p);
        launched = true;
    }
    if (!launched)
        dispatch_sp_layout(ctx->spaced, ctx->layout, [&](auto sp, auto ly) {
            auto kern = packed ? classify_kernel<decltype(sp)::value, decltype(ly)::value, 0, 0, 8, false, false, true>
                               : classify_kernel<decltype(sp)::value, decltype(ly)::value, 0, 0>;
            // persistent grid = the blocks that are resident at once (a wide probe stage takes more LDS per block than 8 per CU allow)
            int per_cu = 8;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 8;
            const unsigned g = std::min<unsigned>(grid, (unsigned)ctx->n_cu * (unsigned)std::min(per_cu, 8));
            hipLaunchKernelGGL(kern, dim3(g), dim3(256), 0, st, p);
        });
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) {
        HIPCHK(ctx, hipEventRecord(ctx->ev1[evi], st));
        ctx->ev_head = (evi + 1) % bns_ctx::EV_RING;
        ctx->ev_count = std::min(ctx->ev_count + 1, (int)bns_ctx::EV_RING);
    }

    if (can_overflow) {
        u32 h_ovf = 0;
        HIPCHK(ctx, hipMemcpyAsync(&h_ovf, d_ovf, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (h_ovf) {
            if ((rc = ensure(ctx, ctx->scratch, (size_t)total_bases * 16)) != BNS_OK) return rc;
            const dim3 og(std::min<u32>(h_ovf, (u32)ctx->n_cu * 8));
            if (clustered && ctx->table_wide) {
                if (packed) hipLaunchKernelGGL((classify_overflow_kernel<false, 2, true, true>), og, dim3(64), 0, st, p, (u32 *)ctx->scratch.p, (u64)total_bases);
                else        hipLaunchKernelGGL((classify_overflow_kernel<false, 2, true>), og, dim3(64), 0, st, p, (u32 *)ctx->scratch.p, (u64)total_bases);
            } else
            dispatch_sp_layout(ctx->spaced, ctx->layout, [&](auto sp, auto ly) {
                if (packed) hipLaunchKernelGGL((classify_overflow_kernel<decltype(sp)::value, decltype(ly)::value, false, true>), og, dim3(64), 0, st, p,
                                               (u32 *)ctx->scratch.p, (u64)total_bases);
                else        hipLaunchKernelGGL((classify_overflow_kernel<decltype(sp)::value, decltype(ly)::value>), og, dim3(64), 0, st, p,
                                               (u32 *)ctx->scratch.p, (u64)total_bases);
            });
            HIPCHK(ctx, hipGetLastError());
        }
    }
    hipLaunchKernelGGL(unpack_kernel, dim3(grid_for(ctx, n_units, 256)), dim3(256), 0, st, (const uint4 *)ctx->records.p, (u64)n_units,
                       d_taxon, d_missing, d_ambig, d_n_hits);
    HIPCHK(ctx, hipGetLastError());
    return BNS_OK;
}

int bns_classify_batch_device(bns_ctx *ctx, const char *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                              uint64_t total_bases, uint32_t max_read_len, int paired, uint32_t *d_taxon,
                              uint32_t *d_missing, uint32_t *d_ambig, uint32_t *d_n_hits, uint32_t *d_hits, void *stream)
{
    if (!d_bases && total_bases) return BNS_ERR_ARG;
    return classify_device_impl(ctx, d_bases, nullptr, nullptr, d_offsets, n_reads, total_bases, max_read_len, paired, d_taxon, d_missing, d_ambig,
                                d_n_hits, d_hits, stream);
}

int bns_classify_batch_packed_device(bns_ctx *ctx, const uint64_t *d_words, const uint32_t *d_nmask, const uint64_t *d_offsets,
                                     uint64_t n_reads, uint64_t total_bases, uint32_t max_read_len, int paired, uint32_t *d_taxon,
                                     uint32_t *d_missing, uint32_t *d_ambig, uint32_t *d_n_hits, uint32_t *d_hits, void *stream)
{
    if (!d_words) return BNS_ERR_ARG;
    return classify_device_impl(ctx, nullptr, d_words, d_nmask, d_offsets, n_reads, total_bases, max_read_len, paired, d_taxon, d_missing, d_ambig,
                                d_n_hits, d_hits, stream);
}

// ---- packed reads on the host side ---------------------------------------------------------------------------------------
uint64_t bns_packed_words(uint64_t total_bases, uint64_t n_reads) { return (total_bases >> 5) + n_reads + 1; }

namespace {
// 32 ASCII bases -> one image word (2-bit codes, first base in the top two bits) + 32 invalid-base flags (bit 31 - i = base i).
// alphabet.h:128 semantics: A/C/G/T in either case, everything else invalid (its code bits are 0).
inline void pack32_scalar(const unsigned char *s, unsigned n, u64 &word, u32 &bad)
{
    u64 w = 0; u32 b = 0;
    for (unsigned i = 0; i < n; ++i) {
        const unsigned c = s[i] & 0xDFu, h = (c >> 1) & 3u;                    // A0 C1 T2 G3
        const bool ok = c == (unsigned)"ACTG"[h];
        const u64 code = ok ? (h ^ (h >> 1)) : 0u;                             // A0 C1 G2 T3
        w |= code << (62u - 2u * i);
        b |= (ok ? 0u : 1u) << (31u - i);
    }
    word = w; bad = b;
}
#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) inline void pack32_avx2(const unsigned char *s, u64 &word, u32 &bad)
{
    const __m256i x = _mm256_loadu_si256((const __m256i *)s);
    const __m256i fold = _mm256_and_si256(x, _mm256_set1_epi8((char)0xDF));
    const __m256i h = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));            // A0 C1 T2 G3
    const __m256i tbl = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i ok = _mm256_cmpeq_epi8(fold, _mm256_shuffle_epi8(tbl, h));
    const __m256i code = _mm256_and_si256(_mm256_xor_si256(h, _mm256_and_si256(_mm256_srli_epi16(h, 1), _mm256_set1_epi8(1))), ok);   // A0 C1 G2 T3, 0 where invalid
    const u32 lo = (u32)_mm256_movemask_epi8(_mm256_slli_epi16(code, 7));                        // bit i = low code bit of base i
    const u32 hi = (u32)_mm256_movemask_epi8(_mm256_slli_epi16(code, 6));
    const u32 okm = (u32)_mm256_movemask_epi8(ok);
    // base 0 to the top: reverse the 32-bit planes, then interleave them (hi plane on the odd bits)
    word = _pdep_u64((u64)__builtin_bitreverse32(hi), 0xAAAAAAAAAAAAAAAAULL) | _pdep_u64((u64)__builtin_bitreverse32(lo), 0x5555555555555555ULL);
    bad = __builtin_bitreverse32(~okm);
}
#endif
struct BadList { std::vector<u64> idx; std::vector<u32> mask; };
// src(r) = where read r's bases are (offsets[] say how many)
extern "C++" template <class Src>
void pack_range(Src src, const u64 *offsets, u64 r0, u64 r1, u64 n_total, u64 *words, BadList &bl, bool simd)
{
    for (u64 r = r0; r < r1; ++r) {
        const u64 o = offsets[r], L = offsets[r + 1] - o, wb = (o >> 5) + r;
        const unsigned char *s = (const unsigned char *)src(r);
        const u64 full = L >> 5;
        for (u64 w = 0; w <= full; ++w) {
            const unsigned n = w < full ? 32u : (unsigned)(L & 31u);
            if (!n) break;
            u64 word; u32 bad;
#if defined(__x86_64__)
            if (simd && n == 32u) pack32_avx2(s + 32 * w, word, bad); else
#endif
            pack32_scalar(s + 32 * w, n, word, bad);
            words[wb + w] = word;
            if (bad) { bl.idx.push_back(wb + w); bl.mask.push_back(bad); }
        }
        // the slack word(s) between this read's last word and the next read's first -- and the one behind the last read -- are
        // zeroed: the image is then a pure function of the reads (recycled buffers upload no stale bytes, images compare byte for byte)
        for (u64 w = wb + ((L + 31u) >> 5), e = (offsets[r + 1] >> 5) + r + 1; w <= e && (w < e || r + 1 == n_total); ++w) words[w] = 0;
    }
}
}  // namespace

extern "C++" template <class Src>
static int pack_reads_impl(Src src, const uint64_t *offsets, uint64_t n_reads, uint64_t *words, uint64_t *bad_word,
                           uint32_t *bad_mask, uint64_t bad_cap, uint64_t *n_bad, int threads)
{
    bool simd = false;
#if defined(__x86_64__)
    simd = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
#endif
    const unsigned nt = (unsigned)std::max(1, std::min<int>(threads, (int)(n_reads / 4096 + 1)));
    std::vector<BadList> bl(nt);
    if (nt == 1) pack_range(src, offsets, 0, n_reads, n_reads, words, bl[0], simd);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] { pack_range(src, offsets, n_reads * t / nt, n_reads * (t + 1) / nt, n_reads, words, bl[t], simd); });
        for (auto &x : th) x.join();
    }
    u64 tot = 0;
    for (auto &b : bl) tot += b.idx.size();
    *n_bad = tot;
    if (tot > bad_cap || (tot && (!bad_word || !bad_mask))) return BNS_ERR_ARG;       // (*n_bad says how much room is needed)
    u64 at = 0;
    for (auto &b : bl) {                                           // (thread ranges are contiguous: the list comes out sorted by word)
        if (b.idx.empty()) continue;
        std::memcpy(bad_word + at, b.idx.data(), b.idx.size() * 8);
        std::memcpy(bad_mask + at, b.mask.data(), b.mask.size() * 4);
        at += b.idx.size();
    }
    return BNS_OK;
}

int bns_pack_reads(const char *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t *words, uint64_t *bad_word,
                   uint32_t *bad_mask, uint64_t bad_cap, uint64_t *n_bad, int threads)
{
    if (!offsets || !words || !n_bad || (!bases && n_reads && offsets[n_reads])) return BNS_ERR_ARG;
    return pack_reads_impl([=](u64 r) { return bases + offsets[r]; }, offsets, n_reads, words, bad_word, bad_mask, bad_cap, n_bad, threads);
}

int bns_pack_reads_ptrs(const char *const *seqs, const uint32_t *lens, uint64_t n_reads, uint64_t *offsets, uint64_t *words,
                        uint64_t *bad_word, uint32_t *bad_mask, uint64_t bad_cap, uint64_t *n_bad, int threads)
{
    if (!offsets || !words || !n_bad || (n_reads && (!seqs || !lens))) return BNS_ERR_ARG;
    offsets[0] = 0;
    for (u64 r = 0; r < n_reads; ++r) offsets[r + 1] = offsets[r] + lens[r];
    return pack_reads_impl([=](u64 r) { return seqs[r]; }, offsets, n_reads, words, bad_word, bad_mask, bad_cap, n_bad, threads);
}

namespace {
__global__ void scatter_mask_kernel(const u64 *__restrict__ idx, const u32 *__restrict__ mask, u64 n, u32 *__restrict__ dense)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dense[idx[i]] = mask[i];
}

// What a host batch is made of: ASCII bases, or packed words + the sparse list of words that hold an invalid base.
struct HostIn { const char *bases = nullptr; const u64 *words = nullptr; const u64 *bad_word = nullptr; const u32 *bad_mask = nullptr; u64 n_bad = 0; };

// Upload + classify of one host batch, results left in ctx->st_out[0..3] (+ st_hits): large batches go up in slices on a second
// stream, slice i+1 while slice i is classified (the upload is the longer leg: 150 -- packed 40 -- bytes per read over PCIe against
// ~0.6 ns of kernel).  A slice is a unit-aligned read range; device offsets stay absolute, so a slice is just a shifted offsets
// pointer and a shifted output pointer.
// Where the per-unit results of a host batch go; a sliced batch copies each slice's results back as soon as that slice is classified
// (the copy runs while the next slice is uploaded: the two directions of the link are separate), and says so in `copied`.
struct HostOut { uint32_t *taxon = nullptr, *missing = nullptr, *ambig = nullptr, *n_hits = nullptr; bool copied = false; };

int classify_host_impl(bns_ctx *ctx, const HostIn &in, const uint64_t *offsets, uint64_t n_reads, int paired, bool want_missing, bool want_ambig,
                       bool want_nhits, bool want_hits, HostOut *out = nullptr)
{
    int rc;
    const bool packed = in.words != nullptr;
    const u64 total = offsets[n_reads];
    const u64 n_units = n_reads / (paired ? 2 : 1);
    // the longest read of a range of the batch (what a launch sizes its rounds by): scanned per slice, next to that slice's
    // launch, so that the scan of 10 M offsets (80 MB) runs under the uploads instead of in front of them
    auto longest = [&](u64 r0, u64 r1) { u32 m = 0; for (u64 r = r0; r < r1; ++r) m = std::max<u32>(m, (u32)(offsets[r + 1] - offsets[r])); return m; };
    const u64 n_words = bns_packed_words(total, n_reads);
    if (packed) {
        if ((rc = ensure(ctx, ctx->st_words, (size_t)n_words * 8 + 8)) != BNS_OK) return rc;
        if (in.n_bad && (rc = ensure(ctx, ctx->st_nmask, (size_t)n_words * 4 + 4)) != BNS_OK) return rc;
        if (in.n_bad && (rc = ensure(ctx, ctx->st_bad, (size_t)in.n_bad * 12)) != BNS_OK) return rc;
    } else if ((rc = ensure(ctx, ctx->st_bases, (size_t)total + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_offsets, (size_t)(n_reads + 1) * 8)) != BNS_OK) return rc;
    for (int i = 0; i < 4; ++i) if ((rc = ensure(ctx, ctx->st_out[i], (size_t)n_units * 4)) != BNS_OK) return rc;
    if (want_hits && (rc = ensure(ctx, ctx->st_hits, (size_t)total * 4 + 4)) != BNS_OK) return rc;
    hipStream_t st = ctx->stream;
    const u32 *d_nmask = nullptr;
    if (packed && in.n_bad) {
        // the dense flag words the kernel reads: zero, then the few words that hold an invalid base scattered in
        u64 *d_idx = (u64 *)ctx->st_bad.p;
        u32 *d_msk = (u32 *)((char *)ctx->st_bad.p + (size_t)in.n_bad * 8);
        HIPCHK(ctx, hipMemsetAsync(ctx->st_nmask.p, 0, (size_t)n_words * 4, st));
        HIPCHK(ctx, hipMemcpyAsync(d_idx, in.bad_word, (size_t)in.n_bad * 8, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(d_msk, in.bad_mask, (size_t)in.n_bad * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(scatter_mask_kernel, dim3(grid_for(ctx, in.n_bad, 256)), dim3(256), 0, st, (const u64 *)d_idx, (const u32 *)d_msk, (u64)in.n_bad,
                           (u32 *)ctx->st_nmask.p);
        HIPCHK(ctx, hipGetLastError());
        d_nmask = (const u32 *)ctx->st_nmask.p;
    }
    auto run = [&](u64 r0, u64 nr, u64 u0) -> int {
        u32 *o0 = (u32 *)ctx->st_out[0].p + u0, *o1 = want_missing ? (u32 *)ctx->st_out[1].p + u0 : nullptr, *o2 = want_ambig ? (u32 *)ctx->st_out[2].p + u0 : nullptr;
        u32 *o3 = (want_nhits || want_hits) ? (u32 *)ctx->st_out[3].p + u0 : nullptr, *oh = want_hits ? (u32 *)ctx->st_hits.p : nullptr;
        // (packed: the word base of a read is (offsets[r] >> 5) + r with r counted from the START of the batch, so a slice hands the
        // kernel word / flag pointers advanced by r0 words: read r0 + i of the batch is read i of the launch)
        return classify_device_impl(ctx, packed ? nullptr : (const char *)ctx->st_bases.p, packed ? (const u64 *)ctx->st_words.p + r0 : nullptr,
                                    (packed && d_nmask) ? d_nmask + r0 : nullptr, (const u64 *)ctx->st_offsets.p + r0, nr, total, std::max<u32>(longest(r0, r0 + nr), 1),
                                    paired, o0, o1, o2, o3, oh, st);
    };
    auto upload = [&](u64 r0, u64 r1, hipStream_t cs) -> int {
        const u64 b0 = offsets[r0], b1 = offsets[r1];
        if (packed) {
            const u64 w0 = (b0 >> 5) + r0, w1 = r1 == n_reads ? n_words : (b1 >> 5) + r1;
            if (w1 > w0) HIPCHK(ctx, hipMemcpyAsync((u64 *)ctx->st_words.p + w0, in.words + w0, (size_t)(w1 - w0) * 8, hipMemcpyHostToDevice, cs));
        } else if (b1 > b0) HIPCHK(ctx, hipMemcpyAsync((char *)ctx->st_bases.p + b0, in.bases + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, cs));
        HIPCHK(ctx, hipMemcpyAsync((u64 *)ctx->st_offsets.p + r0, offsets + r0, (size_t)(r1 - r0 + 1) * 8, hipMemcpyHostToDevice, cs));
        return BNS_OK;
    };
    size_t slice_bytes = (size_t)64 << 20;                                // (8 .. 128 MiB measured within 5 % of each other; 64 the best)
    if (ctx->dbg & BNS_DBG_SLICE_8K) slice_bytes = (size_t)8 << 10;       // (tests force slicing on small batches)
    const int nmr = paired ? 2 : 1;
    const u64 in_bytes = packed ? n_words * 8 : total;
    u64 n_slices = std::min<u64>(16, std::max<u64>(1, in_bytes / slice_bytes));
    if (n_slices > n_units) n_slices = n_units;
    if (n_slices > 1) {
        if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        if (out && !ctx->back_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->back_stream, hipStreamNonBlocking));
        for (u64 i = 0; i < n_slices; ++i) {
            if (!ctx->slice_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->slice_ev[i], hipEventDisableTiming));
            if (out && !ctx->done_ev[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->done_ev[i], hipEventDisableTiming));
        }
        // size the per-call workspaces for the largest slice up front: growing one mid-loop would hipFree, i.e. drain the GPU
        const u64 max_slice_units = n_units / n_slices + 2;
        if ((rc = ensure(ctx, ctx->records, (size_t)max_slice_units * 16)) != BNS_OK) return rc;
        if ((rc = ensure(ctx, ctx->ovf_list, (size_t)max_slice_units * 8)) != BNS_OK) return rc;
        HIPCHK(ctx, hipStreamSynchronize(st));                            // the staging buffers may still be read by earlier work
        hipStream_t cs = ctx->copy_stream;              // (two alternating copy streams were tried: the link, not the DMA engine, is the limit)
        auto slice_end = [&](u64 i) { return (i + 1 == n_slices) ? n_units : n_units * (i + 1) / n_slices; };
        auto send = [&](u64 i) -> int {                 // slice i's upload, and the event the classify stream waits for
            const u64 a = i ? slice_end(i - 1) : 0, b = slice_end(i);
            int rc2 = upload(a * nmr, b * nmr, cs);
            if (rc2 != BNS_OK) return rc2;
            HIPCHK(ctx, hipEventRecord(ctx->slice_ev[i], cs));
            return BNS_OK;
        };
        if ((rc = send(0)) != BNS_OK) return rc;
        u64 u0 = 0;
        for (u64 i = 0; i < n_slices; ++i) {
            const u64 u1 = slice_end(i);
            // slice i + 1 is on its way before slice i is classified (a launch whose units can overflow the in-LDS counter ends
            // with a host-side wait: the next upload must not queue up behind that)
            if (i + 1 < n_slices && (rc = send(i + 1)) != BNS_OK) { (void)hipStreamSynchronize(cs); return rc; }
            HIPCHK(ctx, hipStreamWaitEvent(st, ctx->slice_ev[i], 0));
            if (u1 > u0 && (rc = run(u0 * nmr, (u1 - u0) * nmr, u0)) != BNS_OK) { (void)hipStreamSynchronize(cs); return rc; }
            if (out && u1 > u0) {
                // (on a stream of their own: queued on the classify stream these copies held up the next slice's launch, and with it
                // the upload after that -- 711 -> 860 M reads/s with all four arrays coming back)
                const size_t nb = (size_t)(u1 - u0) * 4;
                hipStream_t bs = ctx->back_stream;
                HIPCHK(ctx, hipEventRecord(ctx->done_ev[i], st));
                HIPCHK(ctx, hipStreamWaitEvent(bs, ctx->done_ev[i], 0));
                HIPCHK(ctx, hipMemcpyAsync(out->taxon + u0, (u32 *)ctx->st_out[0].p + u0, nb, hipMemcpyDeviceToHost, bs));
                if (out->missing) HIPCHK(ctx, hipMemcpyAsync(out->missing + u0, (u32 *)ctx->st_out[1].p + u0, nb, hipMemcpyDeviceToHost, bs));
                if (out->ambig) HIPCHK(ctx, hipMemcpyAsync(out->ambig + u0, (u32 *)ctx->st_out[2].p + u0, nb, hipMemcpyDeviceToHost, bs));
                if (out->n_hits) HIPCHK(ctx, hipMemcpyAsync(out->n_hits + u0, (u32 *)ctx->st_out[3].p + u0, nb, hipMemcpyDeviceToHost, bs));
            }
            u0 = u1;
        }
        if (out) {
            out->copied = true;
            HIPCHK(ctx, hipStreamSynchronize(ctx->back_stream));       // (the caller's arrays are complete when this returns)
        }
        return BNS_OK;
    }
    if ((rc = upload(0, n_reads, st)) != BNS_OK) return rc;
    return run(0, n_reads, 0);
}
}  // namespace

static int classify_host_entry(bns_ctx *ctx, const HostIn &in, const uint64_t *offsets, uint64_t n_reads, int paired,
                               uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint32_t *hits)
{
    int rc = ready(ctx, true, true);
    if (rc != BNS_OK) return rc;
    if (!offsets || !taxon) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const u64 total = offsets[n_reads];
    const u64 n_units = n_reads / (paired ? 2 : 1);
    if (n_units == 0) return BNS_OK;
    HostOut out; out.taxon = taxon; out.missing = missing; out.ambig = ambig; out.n_hits = n_hits;
    if ((rc = classify_host_impl(ctx, in, offsets, n_reads, paired, missing != nullptr, ambig != nullptr, n_hits != nullptr, hits != nullptr, &out)) != BNS_OK) {
        (void)hipStreamSynchronize(ctx->stream);                  // (copies into the caller's arrays may be in flight)
        if (ctx->back_stream) (void)hipStreamSynchronize(ctx->back_stream);
        return rc;
    }
    hipStream_t st = ctx->stream;
    if (!out.copied) {
        HIPCHK(ctx, hipMemcpyAsync(taxon, ctx->st_out[0].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
        if (missing) HIPCHK(ctx, hipMemcpyAsync(missing, ctx->st_out[1].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
        if (ambig) HIPCHK(ctx, hipMemcpyAsync(ambig, ctx->st_out[2].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
        if (n_hits) HIPCHK(ctx, hipMemcpyAsync(n_hits, ctx->st_out[3].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    }
    if (hits && total) HIPCHK(ctx, hipMemcpyAsync(hits, ctx->st_hits.p, (size_t)total * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

int bns_classify_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads, int paired,
                       uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint32_t *hits)
{
    HostIn in; in.bases = bases;
    return classify_host_entry(ctx, in, offsets, n_reads, paired, taxon, missing, ambig, n_hits, hits);
}

int bns_classify_batch_packed(bns_ctx *ctx, const uint64_t *words, const uint64_t *bad_word, const uint32_t *bad_mask, uint64_t n_bad,
                              const uint64_t *offsets, uint64_t n_reads, int paired,
                              uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint32_t *hits)
{
    if (!words || (n_bad && (!bad_word || !bad_mask))) return BNS_ERR_ARG;
    HostIn in; in.words = words; in.bad_word = bad_word; in.bad_mask = bad_mask; in.n_bad = n_bad;
    return classify_host_entry(ctx, in, offsets, n_reads, paired, taxon, missing, ambig, n_hits, hits);
}

static int classify_runs_entry(bns_ctx *ctx, const HostIn &in, const uint64_t *offsets, uint64_t n_reads, int paired,
                               uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint64_t *run_start,
                               uint32_t *n_runs, const uint32_t **run_tax, const uint32_t **run_len, uint64_t *n_runs_total)
{
    int rc = ready(ctx, true, true);
    if (rc != BNS_OK) return rc;
    if (!offsets || !taxon || !run_start || !n_runs || !run_tax || !run_len) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const u64 total = offsets[n_reads];
    const int nm = paired ? 2 : 1;
    const u64 n_units = n_reads / (u64)nm;
    *run_tax = *run_len = nullptr;
    if (n_runs_total) *n_runs_total = 0;
    if (n_units == 0) return BNS_OK;
    if ((rc = ensure(ctx, ctx->st_runs[0], (size_t)n_units * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_runs[1], (size_t)n_units * 4)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_runs[2], (size_t)total * 4 + 4)) != BNS_OK) return rc;      // a run per hit at worst
    if ((rc = ensure(ctx, ctx->st_runs[3], (size_t)total * 4 + 4)) != BNS_OK) return rc;
    if ((rc = classify_host_impl(ctx, in, offsets, n_reads, paired, true, true, true, true)) != BNS_OK) return rc;
    hipStream_t st = ctx->stream;
    unsigned long long *d_cur = &((SmallLayout *)ctx->small.p)->runs_cursor;
    HIPCHK(ctx, hipMemsetAsync(d_cur, 0, 8, st));
    hipLaunchKernelGGL(hit_runs_kernel, dim3(grid_for(ctx, (n_units + HIT_RUNS_GROUP - 1) / HIT_RUNS_GROUP, 4)), dim3(256), 0, st, (const u32 *)ctx->st_hits.p,
                       (const u64 *)ctx->st_offsets.p, (u32)nm, (const u32 *)ctx->st_out[3].p, (u64)n_units, (u64 *)ctx->st_runs[0].p,
                       (u32 *)ctx->st_runs[1].p, (u32 *)ctx->st_runs[2].p, (u32 *)ctx->st_runs[3].p, d_cur);
    HIPCHK(ctx, hipGetLastError());
    unsigned long long n_tot = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n_tot, d_cur, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(taxon, ctx->st_out[0].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    if (missing) HIPCHK(ctx, hipMemcpyAsync(missing, ctx->st_out[1].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    if (ambig) HIPCHK(ctx, hipMemcpyAsync(ambig, ctx->st_out[2].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    if (n_hits) HIPCHK(ctx, hipMemcpyAsync(n_hits, ctx->st_out[3].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(run_start, ctx->st_runs[0].p, (size_t)n_units * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(n_runs, ctx->st_runs[1].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));                    // n_tot is known from here on
    if (ctx->h_run_cap < n_tot) {
        if (ctx->h_run_tax) (void)hipHostFree(ctx->h_run_tax);
        if (ctx->h_run_len) (void)hipHostFree(ctx->h_run_len);
        ctx->h_run_tax = ctx->h_run_len = nullptr; ctx->h_run_cap = 0;
        const size_t want = (size_t)n_tot + (size_t)n_tot / 2;
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_run_tax, want * 4, hipHostMallocDefault));
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_run_len, want * 4, hipHostMallocDefault));
        ctx->h_run_cap = want;
    }
    if (n_tot) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_run_tax, ctx->st_runs[2].p, (size_t)n_tot * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_run_len, ctx->st_runs[3].p, (size_t)n_tot * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    *run_tax = ctx->h_run_tax; *run_len = ctx->h_run_len;
    if (n_runs_total) *n_runs_total = n_tot;
    return BNS_OK;
}

int bns_classify_batch_runs(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads, int paired,
                            uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint64_t *run_start,
                            uint32_t *n_runs, const uint32_t **run_tax, const uint32_t **run_len, uint64_t *n_runs_total)
{
    HostIn in; in.bases = bases;
    return classify_runs_entry(ctx, in, offsets, n_reads, paired, taxon, missing, ambig, n_hits, run_start, n_runs, run_tax, run_len, n_runs_total);
}

int bns_classify_batch_packed_runs(bns_ctx *ctx, const uint64_t *words, const uint64_t *bad_word, const uint32_t *bad_mask, uint64_t n_bad,
                                   const uint64_t *offsets, uint64_t n_reads, int paired,
                                   uint32_t *taxon, uint32_t *missing, uint32_t *ambig, uint32_t *n_hits, uint64_t *run_start,
                                   uint32_t *n_runs, const uint32_t **run_tax, const uint32_t **run_len, uint64_t *n_runs_total)
{
    if (!words || (n_bad && (!bad_word || !bad_mask))) return BNS_ERR_ARG;
    HostIn in; in.words = words; in.bad_word = bad_word; in.bad_mask = bad_mask; in.n_bad = n_bad;
    return classify_runs_entry(ctx, in, offsets, n_reads, paired, taxon, missing, ambig, n_hits, run_start, n_runs, run_tax, run_len, n_runs_total);
}

int bns_encode_batch_device(bns_ctx *ctx, const char *d_bases, const uint64_t *d_offsets, uint64_t n_reads,
                            uint64_t total_bases, uint64_t *d_kmers, uint32_t *d_n_kmers, void *stream)
{
    int rc = ready(ctx, false, false);
    if (rc != BNS_OK) return rc;
    if (!d_offsets || !d_n_kmers || (!d_kmers && total_bases)) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (n_reads == 0) return BNS_OK;
    if ((rc = pack_reads(ctx, d_bases, d_offsets, n_reads, total_bases, st)) != BNS_OK) return rc;
    ClassifyParams p;
    fill_params(ctx, p);
    p.offsets = d_offsets; p.n_units = n_reads; p.nmates = 1;
    p.emit_none = (ctx->spaced && !ctx->spaced_intended) ? 1 : 0;
    const unsigned grid = grid_for(ctx, n_reads, 4);
    if ((rc = set_win_scratch(ctx, p, grid)) != BNS_OK) return rc;
    if (ctx->spaced) hipLaunchKernelGGL(encode_kernel<true>, dim3(grid), dim3(256), 0, st, p, d_kmers, d_n_kmers);
    else             hipLaunchKernelGGL(encode_kernel<false>, dim3(grid), dim3(256), 0, st, p, d_kmers, d_n_kmers);
    HIPCHK(ctx, hipGetLastError());
    return BNS_OK;
}

int bns_encode_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t *kmers, uint32_t *n_kmers)
{
    int rc = ready(ctx, false, false);
    if (rc != BNS_OK) return rc;
    if (!offsets || !n_kmers) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_reads == 0) return BNS_OK;
    const u64 total = offsets[n_reads];
    if ((rc = ensure(ctx, ctx->st_bases, (size_t)total + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_offsets, (size_t)(n_reads + 1) * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_kmers, (size_t)total * 8 + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_out[0], (size_t)n_reads * 4)) != BNS_OK) return rc;
    hipStream_t st = ctx->stream;
    if (total) HIPCHK(ctx, hipMemcpyAsync(ctx->st_bases.p, bases, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_offsets.p, offsets, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    rc = bns_encode_batch_device(ctx, (const char *)ctx->st_bases.p, (const u64 *)ctx->st_offsets.p, n_reads, total,
                                 (u64 *)ctx->st_kmers.p, (u32 *)ctx->st_out[0].p, st);
    if (rc != BNS_OK) return rc;
    if (total && kmers) HIPCHK(ctx, hipMemcpyAsync(kmers, ctx->st_kmers.p, (size_t)total * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(n_kmers, ctx->st_out[0].p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

// wy::WyRand stand-in for the default character tables: Lemire's wyhash64 (state += 0x60bee2bee120fc15, two 64x64->128
// multiply folds), seeded as encoder.h:682-683 seeds the two hashers.  PARITY UNPINNED: pass real tables instead.
static u64 wyhash64_next(u64 &state)
{
    state += 0x60bee2bee120fc15ULL;
    unsigned __int128 t = (unsigned __int128)state * 0xa3b195354a39b70dULL;
    const u64 m1 = (u64)(t >> 64) ^ (u64)t;
    t = (unsigned __int128)m1 * 0x1b03738712fad5c9ULL;
    return (u64)(t >> 64) ^ (u64)t;
}

int bns_rolling_tables(uint64_t seed1, uint64_t seed2, uint64_t *fwd, uint64_t *rc)
{
    if (!fwd || !rc) return BNS_ERR_ARG;
    u64 sf = (u32)(seed1 ^ seed2), sr = (u32)((seed2 * seed1) ^ (seed2 ^ seed1));
    for (int i = 0; i < 256; ++i) fwd[i] = wyhash64_next(sf);
    for (int i = 0; i < 256; ++i) rc[i] = wyhash64_next(sr);
    return BNS_OK;
}

int bns_rolling_hash_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                           const uint64_t *fwd_table, const uint64_t *rc_table, uint64_t *hashes, uint32_t *n_hashes)
{
    return bns_rolling_hash_windowed_batch(ctx, bases, offsets, n_seqs, k, canon, 0, fwd_table, rc_table, hashes, n_hashes);
}

int bns_rolling_hash_windowed_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                                    uint32_t w, const uint64_t *fwd_table, const uint64_t *rc_table, uint64_t *hashes, uint32_t *n_hashes)
{
    if (!ctx || !offsets || !n_hashes) return BNS_ERR_ARG;
    const bool windowed = w > k;                                          // RollingHasher::window(): w <= k_ means none (encoder.h:664-665)
    const u32 per = (windowed && canon) ? 2u : 1u;                        // entries per base the buffers are laid out with
    if (k == 0) return fail(ctx, BNS_ERR_ARG, "k must be positive");
    if ((fwd_table == nullptr) != (rc_table == nullptr)) return fail(ctx, BNS_ERR_ARG, "pass both character tables or neither");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_seqs == 0) return BNS_OK;
    const u64 total = offsets[n_seqs];
    u64 tabs[512];
    if (fwd_table) { std::memcpy(tabs, fwd_table, 2048); std::memcpy(tabs + 256, rc_table, 2048); }
    else bns_rolling_tables(1337, 137, tabs, tabs + 256);                 // RollingHasher's default seeds (encoder.h:673)
    int rc;
    if ((rc = ensure(ctx, ctx->st_bases, (size_t)total + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_offsets, (size_t)(n_seqs + 1) * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_kmers, (size_t)total * 8 * per + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_out[0], (size_t)n_seqs * 4)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_aux, sizeof(tabs))) != BNS_OK) return rc;
    if (windowed) {
        if ((rc = ensure(ctx, ctx->st_hits, (size_t)total * 8 * per + 8)) != BNS_OK) return rc;
        if ((rc = ensure(ctx, ctx->st_out[1], (size_t)n_seqs * 4)) != BNS_OK) return rc;
    }
    hipStream_t st = ctx->stream;
    if (total) HIPCHK(ctx, hipMemcpyAsync(ctx->st_bases.p, bases, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_offsets.p, offsets, (size_t)(n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_aux.p, tabs, sizeof(tabs), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rolling_hash_kernel, dim3(grid_for(ctx, n_seqs, 4)), dim3(256), 0, st, (const u8 *)ctx->st_bases.p,
                       (const u64 *)ctx->st_offsets.p, (u64)n_seqs, (u32)k, canon ? 1 : 0, windowed ? 1 : 0, (const u64 *)ctx->st_aux.p,
                       (const u64 *)ctx->st_aux.p + 256, (u64 *)ctx->st_kmers.p, (u32 *)ctx->st_out[0].p);
    HIPCHK(ctx, hipGetLastError());
    const void *d_res = ctx->st_kmers.p, *d_cnt = ctx->st_out[0].p;
    if (windowed) {
        hipLaunchKernelGGL(stream_window_kernel, dim3(grid_for(ctx, n_seqs, 4)), dim3(256), 0, st, (const u64 *)ctx->st_kmers.p,
                           (const u32 *)ctx->st_out[0].p, (const u64 *)ctx->st_offsets.p, (u64)n_seqs, per, (u32)(w - k + 1),
                           (u64 *)ctx->st_hits.p, (u32 *)ctx->st_out[1].p);
        HIPCHK(ctx, hipGetLastError());
        d_res = ctx->st_hits.p; d_cnt = ctx->st_out[1].p;
    }
    if (total && hashes) HIPCHK(ctx, hipMemcpyAsync(hashes, d_res, (size_t)total * 8 * per, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(n_hashes, d_cnt, (size_t)n_seqs * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

// make_nthash_lut's geometry (encoder.h:93-103): the base's seed at the letter (both cases), its complement's seed at
// index (letter & 7), everything else 0.
int bns_nthash_tables(uint64_t seed_a, uint64_t seed_c, uint64_t seed_g, uint64_t seed_t, uint64_t *table256)
{
    if (!table256) return BNS_ERR_ARG;
    std::memset(table256, 0, 256 * sizeof(uint64_t));
    table256[4] = table256['a'] = table256['A'] = seed_a;      // 'T' & 7 == 4: T's complement
    table256[7] = table256['c'] = table256['C'] = seed_c;      // 'G' & 7 == 7
    table256[3] = table256['g'] = table256['G'] = seed_g;      // 'C' & 7 == 3
    table256[1] = table256['t'] = table256['T'] = seed_t;      // 'A' & 7 == 1
    return BNS_OK;
}

int bns_for_each_hash_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                            const uint64_t *table256, uint64_t *hashes, uint32_t *n_hashes)
{
    if (!ctx || !offsets || !n_hashes) return BNS_ERR_ARG;
    if (k == 0) {                                             // encoder.h:357,362: k = 0 means the Spacer's k
        if (!ctx->enc_set) return fail(ctx, BNS_ERR_STATE, "k = 0 takes the encoder's k: configure the encoder first (bns_set_encoder)");
        k = ctx->k;
    }
    if (ctx->enc_set && ctx->spaced) return fail(ctx, BNS_ERR_ARG, "Can't for_each_hash for a spaced spacer");       // encoder.h:364
    if (ctx->enc_set && ctx->win > ctx->c) return fail(ctx, BNS_ERR_ARG, "Can't for_each_hash for a windowed spacer"); // encoder.h:363
    if (canon < 0) canon = (ctx->enc_set && ctx->canon) ? 1 : 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_seqs == 0) return BNS_OK;
    const u64 total = offsets[n_seqs];
    u64 tab[256];
    if (table256) std::memcpy(tab, table256, sizeof(tab));
    else bns_nthash_tables(0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL, 0x295549f54be24456ULL, tab);   // ntHash's published seeds
    int rc;
    if ((rc = ensure(ctx, ctx->st_bases, (size_t)total + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_offsets, (size_t)(n_seqs + 1) * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_kmers, (size_t)total * 8 + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_out[0], (size_t)n_seqs * 4)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_aux, sizeof(tab))) != BNS_OK) return rc;
    hipStream_t st = ctx->stream;
    if (total) HIPCHK(ctx, hipMemcpyAsync(ctx->st_bases.p, bases, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_offsets.p, offsets, (size_t)(n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_aux.p, tab, sizeof(tab), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(nthash_kernel, dim3(grid_for(ctx, n_seqs, 4)), dim3(256), 0, st, (const u8 *)ctx->st_bases.p,
                       (const u64 *)ctx->st_offsets.p, (u64)n_seqs, (u32)k, canon ? 1 : 0, (const u64 *)ctx->st_aux.p,
                       (u64 *)ctx->st_kmers.p, (u32 *)ctx->st_out[0].p);
    HIPCHK(ctx, hipGetLastError());
    if (total && hashes) HIPCHK(ctx, hipMemcpyAsync(hashes, ctx->st_kmers.p, (size_t)total * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(n_hashes, ctx->st_out[0].p, (size_t)n_seqs * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

// CharacterHash<__uint128_t>'s default tables: two draws per entry, of which only the second survives (its 128-bit mask is
// truncated to 64 bits on the way into the uint64_t member maxval_, characterhash.h:82-97) -- hi = 0 in every entry.
int bns_rolling_tables128(uint64_t seed1, uint64_t seed2, uint64_t *fwd_lohi, uint64_t *rc_lohi)
{
    if (!fwd_lohi || !rc_lohi) return BNS_ERR_ARG;
    u64 sf = (u32)(seed1 ^ seed2), sr = (u32)((seed2 * seed1) ^ (seed2 ^ seed1));
    for (int i = 0; i < 256; ++i) { (void)wyhash64_next(sf); fwd_lohi[2 * i] = wyhash64_next(sf); fwd_lohi[2 * i + 1] = 0; }
    for (int i = 0; i < 256; ++i) { (void)wyhash64_next(sr); rc_lohi[2 * i] = wyhash64_next(sr); rc_lohi[2 * i + 1] = 0; }
    return BNS_OK;
}

int bns_rolling_hash128_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                              const uint64_t *fwd_lohi, const uint64_t *rc_lohi, uint64_t *hashes_lohi, uint32_t *n_hashes)
{
    return bns_rolling_hash128_windowed_batch(ctx, bases, offsets, n_seqs, k, canon, 0, fwd_lohi, rc_lohi, hashes_lohi, n_hashes);
}

int bns_rolling_hash128_windowed_batch(bns_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_seqs, uint32_t k, int canon,
                                       uint32_t w, const uint64_t *fwd_lohi, const uint64_t *rc_lohi, uint64_t *hashes_lohi, uint32_t *n_hashes)
{
    if (!ctx || !offsets || !n_hashes) return BNS_ERR_ARG;
    const bool windowed = w > k;                                          // RollingHasher::window(): w <= k_ means none (encoder.h:664-665)
    const u32 per = (windowed && canon) ? 2u : 1u;
    if (k == 0) return fail(ctx, BNS_ERR_ARG, "k must be positive");
    if ((fwd_lohi == nullptr) != (rc_lohi == nullptr)) return fail(ctx, BNS_ERR_ARG, "pass both character tables or neither");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_seqs == 0) return BNS_OK;
    const u64 total = offsets[n_seqs];
    std::vector<u64> tabs(1024);
    if (fwd_lohi) { std::memcpy(tabs.data(), fwd_lohi, 4096); std::memcpy(tabs.data() + 512, rc_lohi, 4096); }
    else bns_rolling_tables128(1337, 137, tabs.data(), tabs.data() + 512);
    int rc;
    if ((rc = ensure(ctx, ctx->st_bases, (size_t)total + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_offsets, (size_t)(n_seqs + 1) * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_kmers, (size_t)total * 16 * per + 16)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_out[0], (size_t)n_seqs * 4)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_aux, 8192)) != BNS_OK) return rc;
    if (windowed) {
        if ((rc = ensure(ctx, ctx->st_hits, (size_t)total * 16 * per + 16)) != BNS_OK) return rc;
        if ((rc = ensure(ctx, ctx->st_out[1], (size_t)n_seqs * 4)) != BNS_OK) return rc;
    }
    hipStream_t st = ctx->stream;
    if (total) HIPCHK(ctx, hipMemcpyAsync(ctx->st_bases.p, bases, (size_t)total, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_offsets.p, offsets, (size_t)(n_seqs + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_aux.p, tabs.data(), 8192, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rolling_hash128_kernel, dim3(grid_for(ctx, n_seqs, 4)), dim3(256), 0, st, (const u8 *)ctx->st_bases.p,
                       (const u64 *)ctx->st_offsets.p, (u64)n_seqs, (u32)k, canon ? 1 : 0, windowed ? 1 : 0, (const u64 *)ctx->st_aux.p,
                       (const u64 *)ctx->st_aux.p + 512, (u64 *)ctx->st_kmers.p, (u32 *)ctx->st_out[0].p);
    HIPCHK(ctx, hipGetLastError());
    const void *d_res = ctx->st_kmers.p, *d_cnt = ctx->st_out[0].p;
    if (windowed) {
        hipLaunchKernelGGL(stream_window128_kernel, dim3(grid_for(ctx, n_seqs, 4)), dim3(256), 0, st, (const u64 *)ctx->st_kmers.p,
                           (const u32 *)ctx->st_out[0].p, (const u64 *)ctx->st_offsets.p, (u64)n_seqs, per, (u32)(w - k + 1),
                           (u64 *)ctx->st_hits.p, (u32 *)ctx->st_out[1].p);
        HIPCHK(ctx, hipGetLastError());
        d_res = ctx->st_hits.p; d_cnt = ctx->st_out[1].p;
    }
    if (total && hashes_lohi) HIPCHK(ctx, hipMemcpyAsync(hashes_lohi, d_res, (size_t)total * 16 * per, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(n_hashes, d_cnt, (size_t)n_seqs * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

int bns_probe_device(bns_ctx *ctx, const uint64_t *d_kmers, uint64_t n, uint32_t *d_vals, uint8_t *d_found, void *stream)
{
    if (!ctx) return BNS_ERR_ARG;
    if (ctx->layout < 0) return fail(ctx, BNS_ERR_STATE, "no table loaded (bns_load_table)");
    if (n == 0) return BNS_OK;
    if (!d_kmers || !d_vals) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    ClassifyParams p;
    fill_params(ctx, p);
    const unsigned grid = grid_for(ctx, n, 256);
    const int evi = ctx->ev_head;
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev0[evi], st));
    if (ctx->layout == BNS_LAYOUT_MINBUCKET && ctx->table_k != ctx->k)
        return fail(ctx, BNS_ERR_STATE, "encoder k changed after a BNS_LAYOUT_MINBUCKET table was built; reload the table");
    const bool whole_key = ctx->table_canon && ctx->table_shift == 0 && ctx->table_len == ctx->table_k;
    if (ctx->layout == 2 && !whole_key) hipLaunchKernelGGL((probe_kernel<2, 1>), dim3(grid), dim3(256), 0, st, p, d_kmers, (u64)n, d_vals, d_found);
    else if (ctx->layout == 2 && ctx->table_wide) hipLaunchKernelGGL((probe_kernel<2, 2>), dim3(grid), dim3(256), 0, st, p, d_kmers, (u64)n, d_vals, d_found);
    else if (ctx->layout == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(grid), dim3(256), 0, st, p, d_kmers, (u64)n, d_vals, d_found);
    else if (ctx->layout == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(grid), dim3(256), 0, st, p, d_kmers, (u64)n, d_vals, d_found);
    else                       hipLaunchKernelGGL(probe_kernel<0>, dim3(grid), dim3(256), 0, st, p, d_kmers, (u64)n, d_vals, d_found);
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) {
        HIPCHK(ctx, hipEventRecord(ctx->ev1[evi], st));
        ctx->ev_head = (evi + 1) % bns_ctx::EV_RING;
        ctx->ev_count = std::min(ctx->ev_count + 1, (int)bns_ctx::EV_RING);
    }
    return BNS_OK;
}

int bns_probe(bns_ctx *ctx, const uint64_t *kmers, uint64_t n, uint32_t *vals, uint8_t *found)
{
    if (!ctx) return BNS_ERR_ARG;
    if (n == 0) return BNS_OK;
    if (!kmers || !vals) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->st_kmers, (size_t)n * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_out[0], (size_t)n * 4)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_aux, (size_t)n)) != BNS_OK) return rc;
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_kmers.p, kmers, (size_t)n * 8, hipMemcpyHostToDevice, st));
    rc = bns_probe_device(ctx, (const u64 *)ctx->st_kmers.p, n, (u32 *)ctx->st_out[0].p, (u8 *)ctx->st_aux.p, st);
    if (rc != BNS_OK) return rc;
    HIPCHK(ctx, hipMemcpyAsync(vals, ctx->st_out[0].p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (found) HIPCHK(ctx, hipMemcpyAsync(found, ctx->st_aux.p, (size_t)n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

int bns_resolve_batch(bns_ctx *ctx, const uint32_t *keys, const uint16_t *counts, const uint64_t *starts,
                      uint64_t n_units, uint32_t *taxon)
{
    if (!ctx || !starts || !taxon) return BNS_ERR_ARG;
    if (!ctx->nodes) return fail(ctx, BNS_ERR_STATE, "no taxonomy loaded (bns_load_taxonomy)");
    if (n_units == 0) return BNS_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const u64 total = starts[n_units];
    std::vector<u32> c32((size_t)total);
    for (u64 i = 0; i < total; ++i) c32[(size_t)i] = counts[i];
    int rc;
    if ((rc = ensure(ctx, ctx->st_kmers, (size_t)total * 4 + 4)) != BNS_OK) return rc;       // keys
    if ((rc = ensure(ctx, ctx->st_hits, (size_t)total * 4 + 4)) != BNS_OK) return rc;        // counts
    if ((rc = ensure(ctx, ctx->st_offsets, (size_t)(n_units + 1) * 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->scratch, (size_t)total * 8 + 8)) != BNS_OK) return rc;
    if ((rc = ensure(ctx, ctx->st_out[0], (size_t)n_units * 4)) != BNS_OK) return rc;
    hipStream_t st = ctx->stream;
    if (total) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->st_kmers.p, keys, (size_t)total * 4, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->st_hits.p, c32.data(), (size_t)total * 4, hipMemcpyHostToDevice, st));
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_offsets.p, starts, (size_t)(n_units + 1) * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)std::min<u64>(n_units, (u64)ctx->n_cu * 16)), dim3(64), 0, st,
                       (const u32 *)ctx->st_kmers.p, (const u32 *)ctx->st_hits.p, (const u64 *)ctx->st_offsets.p, (u64)n_units,
                       (u32 *)ctx->scratch.p, (u64)total, (const TaxNode *)ctx->nodes, ctx->n_nodes, (u32 *)ctx->st_out[0].p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(taxon, ctx->st_out[0].p, (size_t)n_units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return BNS_OK;
}

int bns_build_table_device(bns_ctx *ctx, const char *d_bases, const uint64_t *d_offsets, uint64_t n_genomes,
                           uint64_t total_bases, const uint32_t *d_taxid, uint64_t n_buckets, uint32_t *d_flags,
                           uint64_t *d_keys, uint32_t *d_vals, uint64_t *header4, void *stream)
{
    int rc = ready(ctx, false, true);
    if (rc != BNS_OK) return rc;
    if (!d_offsets || !d_taxid || !d_flags || !d_keys || !d_vals) return BNS_ERR_ARG;
    if (n_buckets < 4 || (n_buckets & (n_buckets - 1))) return fail(ctx, BNS_ERR_TABLE, "n_buckets must be a power of two >= 4");
    // ~0 marks an empty slot while building: a contiguous non-canonical 32-mer can BE ~0 (a spaced one equal to ~0 is never
    // emitted, encoder.h:236-238; a canonical one is never ~0)
    if (ctx->k == 32 && !ctx->canon && !ctx->spaced && !(ctx->win > ctx->c)) return fail(ctx, BNS_ERR_ARG, "device build needs canonical or spaced k-mers when k == 32");
    if (ctx->spaced && !ctx->spaced_intended) return fail(ctx, BNS_ERR_ARG, "spaced build needs spaced_intended=1");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if ((rc = pack_reads(ctx, d_bases, d_offsets, n_genomes, total_bases, st)) != BNS_OK) return rc;
    const unsigned fgrid = grid_for(ctx, n_buckets, 256);
    hipLaunchKernelGGL(fill_u64_kernel, dim3(fgrid), dim3(256), 0, st, d_keys, (u64)n_buckets, BUILD_EMPTY);
    hipLaunchKernelGGL(fill_u32_kernel, dim3(fgrid), dim3(256), 0, st, d_vals, (u64)n_buckets, 0u);
    unsigned long long *d_cnt = ((SmallLayout *)ctx->small.p)->load_cnt;
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, 16, st));           // [0] keys inserted, [1] "table too small" flag of pass 1
    ClassifyParams p;
    fill_params(ctx, p);
    p.offsets = d_offsets; p.n_units = n_genomes; p.nmates = 1;
    const unsigned grid = (unsigned)ctx->n_cu * 8;
    if ((rc = set_win_scratch(ctx, p, grid)) != BNS_OK) return rc;
    if (ctx->spaced) {
        hipLaunchKernelGGL((build_kernel<true, 1>), dim3(grid), dim3(256), 0, st, p, d_taxid, (u64)n_buckets, d_keys, d_vals, d_cnt);
    } else {
        hipLaunchKernelGGL((build_kernel<false, 1>), dim3(grid), dim3(256), 0, st, p, d_taxid, (u64)n_buckets, d_keys, d_vals, d_cnt);
    }
    HIPCHK(ctx, hipGetLastError());
    // the load-factor contract (khash64.h:198) must hold before pass 2, and a full table would never terminate
    unsigned long long h_cnt2[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(h_cnt2, d_cnt, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    const unsigned long long h_cnt = h_cnt2[0];
    const u64 upper = (u64)(n_buckets * 0.77 + 0.5);
    if (h_cnt2[1]) return fail(ctx, BNS_ERR_TABLE, "n_buckets too small: the table filled up while inserting");
    if (h_cnt > upper) return fail(ctx, BNS_ERR_TABLE, "n_buckets too small: load factor would exceed 0.77");
    if (ctx->spaced) {
        hipLaunchKernelGGL((build_kernel<true, 2>), dim3(grid), dim3(256), 0, st, p, d_taxid, (u64)n_buckets, d_keys, d_vals, d_cnt);
    } else {
        hipLaunchKernelGGL((build_kernel<false, 2>), dim3(grid), dim3(256), 0, st, p, d_taxid, (u64)n_buckets, d_keys, d_vals, d_cnt);
    }
    hipLaunchKernelGGL(build_finish_kernel, dim3(grid_for(ctx, std::max<u64>(1, n_buckets >> 4), 256)), dim3(256), 0, st,
                       (u64)n_buckets, d_flags, d_keys, d_vals);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (header4) { header4[0] = n_buckets; header4[1] = h_cnt; header4[2] = h_cnt; header4[3] = upper; }
    return BNS_OK;
}

int bns_dev_alloc(bns_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(out, bytes ? bytes : 4));
    return BNS_OK;
}
int bns_host_alloc(bns_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return BNS_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // (portable: a block read for one device may be classified by another -- `bonsai classify -g 0-7` hands blocks to whichever is free)
    HIPCHK(ctx, hipHostMalloc(out, bytes ? bytes : 4, hipHostMallocPortable));
    return BNS_OK;
}
int bns_host_free(bns_ctx *ctx, void *p)
{
    if (!ctx) return BNS_ERR_ARG;
    if (p) HIPCHK(ctx, hipHostFree(p));
    return BNS_OK;
}
int bns_dev_free(bns_ctx *ctx, void *p)
{
    if (!ctx) return BNS_ERR_ARG;
    if (p) HIPCHK(ctx, hipFree(p));
    return BNS_OK;
}
int bns_dev_upload(bns_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return BNS_ERR_ARG;
    if (bytes) HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return BNS_OK;
}
int bns_dev_download(bns_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx) return BNS_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bytes) HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return BNS_OK;
}
int bns_dev_sync(bns_ctx *ctx)
{
    if (!ctx) return BNS_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipDeviceSynchronize());
    return BNS_OK;
}

}  // extern "C"

#include "bns_ingest.hip"
