// bns_inflate.hip -- BGZF members inflated on the GPU: the kernels around bns_inflate_wave.hpp's member-per-wavefront decoder (the
// default) and bns_inflate.hpp's member-per-lane decoder, and their C ABI
// (include/bonsai_amd.h, "BGZF members inflated on the device").  A translation unit of its own: it shares nothing with the classify
// path but the device, and has its own handle (stream, staging buffers), so a reader thread can inflate while classify calls run.
#include "../../include/bonsai_amd.h"
#include <hip/hip_runtime.h>
#include "bns_inflate.hpp"
#include "bns_inflate_wave.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>

namespace {
using bns_inf::u8;
using bns_inf::u16;
using bns_inf::u32;
using bns_inf::u64;

// One wavefront per block, one member per lane, MPW lanes busy.  The decoder is a serial chain per member (~110 instructions and two
// LDS round trips per symbol, ~35 ms for a 64 KiB member whatever the neighbours do), so what a batch needs is not busy lanes but
// MANY WAVEFRONTS whose latencies overlap on a SIMD: the code tables in LDS are sized by the busy lanes (MPW x 3872 B with the
// two codes' direct tables, x 800 B without, interleaved by lane, + the CRC table), and a batch of a few thousand members runs
// as 8-lane wavefronts, five (twenty) per CU -- beside the next batch of another handle and beside classify blocks.
template <int MPW, bool LUT>
__global__ __launch_bounds__(64) void inflate_members_kernel(const u8 *__restrict__ comp, const u64 *__restrict__ in_off, const u32 *__restrict__ in_len,
                                                             const u64 *__restrict__ out_off, const u32 *__restrict__ out_len, u64 n,
                                                             u8 *__restrict__ text, u8 *__restrict__ scratch, u32 *__restrict__ crc, u32 *__restrict__ status)
{
    __shared__ u32 s_tables[((LUT ? bns_inf::T_U16 : bns_inf::T_U16_NOLUT) / 2) * MPW];
    __shared__ u32 s_crc[256];
    const u32 lane = threadIdx.x;
    for (u32 i = lane; i < 256u; i += 64u) s_crc[i] = bns_inf::crc32_entry(i);
    __syncthreads();
    const u64 m = (u64)blockIdx.x * MPW + lane;
    if (lane >= (u32)MPW || m >= n) return;
    const bns_inf::Tables<MPW> t{reinterpret_cast<u16 *>(s_tables + lane)};
    u8 *out = text + out_off[m];
    u32 got = 0;
    const u32 st = bns_inf::inflate_member<MPW, LUT>(comp + in_off[m], in_len[m], out, out_len[m], t, scratch + m * (u64)bns_inf::SCRATCH_BYTES, &got);
    // CRC-32 of what was written: the lane's own bytes again (L2-resident), four at a time
    u32 c = 0xFFFFFFFFu, i = 0;
#ifdef BNS_INF_ABLATE_CRC                                    // measurement builds only (wrong checksums): what does the CRC pass cost?
    i = got;
#endif
    for (; i + 4u <= got; i += 4u) {
        c ^= bns_inf::load32u(out + i);
        c = s_crc[c & 0xFFu] ^ (c >> 8);
        c = s_crc[c & 0xFFu] ^ (c >> 8);
        c = s_crc[c & 0xFFu] ^ (c >> 8);
        c = s_crc[c & 0xFFu] ^ (c >> 8);
    }
    for (; i < got; ++i) c = s_crc[(c ^ out[i]) & 0xFFu] ^ (c >> 8);
    crc[m] = ~c;
    status[m] = st;
}

// The other form: ONE MEMBER PER WAVEFRONT (bns_inflate_wave.hpp): the chain of a member's symbols on the scalar unit, tables, copies,
// stores and the CRC by all 64 lanes.  9 KB of LDS per wavefront: sixteen per CU, 4096 members resident.
__global__ __launch_bounds__(64) void inflate_wave_kernel(const u8 *__restrict__ comp, const u8 *__restrict__ comp_end, const u64 *__restrict__ in_off,
                                                          const u32 *__restrict__ in_len, const u64 *__restrict__ out_off, const u32 *__restrict__ out_len, u64 n,
                                                          u8 *text, u32 *__restrict__ crc, u32 *__restrict__ status)
{
    __shared__ bns_infw::WaveLds S;
    const u64 m = blockIdx.x;
    if (m >= n) return;
    u8 *out = text + out_off[m];
    u32 got = 0;
    const u32 st = bns_infw::inflate_member_wave<bns_inf::u8, false>(&S, comp + in_off[m], in_len[m], comp_end, out, out_len[m], &got);
    __syncthreads();
    const u32 c = bns_infw::crc32_wave(S.lut, out, got);
    if (threadIdx.x == 0) { crc[m] = c; status[m] = st; }
}

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};
}  // namespace

struct bns_inflater {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    Buf d_comp, d_text, d_tab, d_res, d_scratch;
    int n_cu = 256;
    float last_kernel_ms = -1.f;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t done = nullptr;      // blocking-sync: the calling thread sleeps through the batch instead of spinning on the stream (the host is short of CPUs, not the GPU)
    // bns_inflate_stream_prefetch: two ranges of host bytes brought up ahead on a stream of their own
    Buf d_pre[2];
    const uint8_t *pre_host[2] = {nullptr, nullptr};
    size_t pre_bytes[2] = {0, 0};
    hipEvent_t pre_done[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    int pre_turn = 0;
    bool called = false;            // `done` has been recorded
    unsigned stream_room = 0;       // bns_inflate_stream_room: symbols of room per byte of a chunk (0: BNS_GZ_RATIO_CAP, or 16)
};

namespace {
#define INFCHK(h, expr)                                                                            \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
            return _e == hipErrorOutOfMemory ? BNS_ERR_NOMEM : BNS_ERR_HIP;                        \
        }                                                                                         \
    } while (0)

int ensure(bns_inflater *h, Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return BNS_OK;
    if (b.p) { INFCHK(h, hipStreamSynchronize(h->stream)); INFCHK(h, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    INFCHK(h, hipMalloc(&b.p, want));
    b.cap = want;
    return BNS_OK;
}
}  // namespace

extern "C" {

int bns_inflater_create(int device, bns_inflater **out)
{
    if (!out) return BNS_ERR_ARG;
    *out = nullptr;
    // One handle at a time: reader threads open theirs side by side, and in a process that has not touched the device yet that is
    // the HIP runtime's own start-up (and this module's load) raced by several threads -- seen to hang on the GPU box.
    static std::mutex create_mu;
    std::lock_guard<std::mutex> lk(create_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return BNS_ERR_HIP;
    bns_inflater *h = new bns_inflater();
    h->device = device;
    hipDeviceProp_t prop;
    hipFuncAttributes fa;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&h->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess ||
        hipGetDeviceProperties(&prop, device) != hipSuccess ||
        hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&inflate_members_kernel<8, true>)) != hipSuccess) {      // (loads the module here, not in the first batch)
        delete h;
        return BNS_ERR_HIP;
    }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = h;
    return BNS_OK;
}

void bns_inflater_destroy(bns_inflater *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    for (hipEvent_t e : h->pre_done) if (e) (void)hipEventDestroy(e);
    for (Buf *b : {&h->d_comp, &h->d_text, &h->d_tab, &h->d_res, &h->d_scratch, &h->d_pre[0], &h->d_pre[1]})
        if (b->p) (void)hipFree(b->p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->done) (void)hipEventDestroy(h->done);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *bns_inflater_error(const bns_inflater *h) { return h ? h->err.c_str() : "null inflater"; }
float bns_inflater_last_kernel_ms(const bns_inflater *h) { return h ? h->last_kernel_ms : -1.f; }

int bns_inflater_host_alloc(bns_inflater *h, size_t bytes, void **out)
{
    if (!h || !out) return BNS_ERR_ARG;
    INFCHK(h, hipSetDevice(h->device));
    INFCHK(h, hipHostMalloc(out, bytes ? bytes : 4, hipHostMallocDefault));
    return BNS_OK;
}
int bns_inflater_host_free(bns_inflater *h, void *p)
{
    if (!h) return BNS_ERR_ARG;
    if (p) INFCHK(h, hipHostFree(p));
    return BNS_OK;
}

}  // extern "C"

// text: where the members' text goes on the HOST (copied back), or d_text_out: where it stays on the DEVICE
static int inflate_members_impl(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                                const uint64_t *out_off, const uint32_t *out_len, uint64_t n_members, uint8_t *text, uint8_t *d_text_out, uint64_t text_bytes,
                                uint32_t *crc32, uint32_t *status)
{
    if (!h) return BNS_ERR_ARG;
    if (n_members == 0) return BNS_OK;
    if (!comp || !in_off || !in_len || !out_off || !out_len || (!text && !d_text_out) || !crc32 || !status) return BNS_ERR_ARG;
    // every member inside its buffers (the kernel trusts the table); lanes read up to 40 bytes past a payload: the staging buffer is padded
    for (u64 i = 0; i < n_members; ++i)
        if (in_off[i] > comp_bytes || in_len[i] > comp_bytes - in_off[i] || out_off[i] > text_bytes || out_len[i] > text_bytes - out_off[i]) {
            h->err = "bns_inflate_members: member " + std::to_string(i) + " lies outside the buffers it was given";
            return BNS_ERR_ARG;
        }
    INFCHK(h, hipSetDevice(h->device));
    int rc;
    const size_t tab_bytes = (size_t)n_members * 24;           // in_off, out_off (u64), in_len, out_len (u32)
    if ((rc = ensure(h, h->d_comp, (size_t)comp_bytes + 64)) != BNS_OK) return rc;
    if (!d_text_out && (rc = ensure(h, h->d_text, (size_t)text_bytes + 16)) != BNS_OK) return rc;
    u8 *const d_out = d_text_out ? d_text_out : (u8 *)h->d_text.p;
    if ((rc = ensure(h, h->d_tab, tab_bytes)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_res, (size_t)n_members * 8)) != BNS_OK) return rc;
    if ((rc = ensure(h, h->d_scratch, (size_t)n_members * bns_inf::SCRATCH_BYTES)) != BNS_OK) return rc;
    hipStream_t st = h->stream;
    u64 *d_in_off = (u64 *)h->d_tab.p, *d_out_off = d_in_off + n_members;
    u32 *d_in_len = (u32 *)(d_out_off + n_members), *d_out_len = d_in_len + n_members;
    u32 *d_crc = (u32 *)h->d_res.p, *d_status = d_crc + n_members;
    INFCHK(h, hipMemcpyAsync(h->d_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
    INFCHK(h, hipMemsetAsync((u8 *)h->d_comp.p + comp_bytes, 0, 64, st));
    INFCHK(h, hipMemcpyAsync(d_in_off, in_off, (size_t)n_members * 8, hipMemcpyHostToDevice, st));
    INFCHK(h, hipMemcpyAsync(d_out_off, out_off, (size_t)n_members * 8, hipMemcpyHostToDevice, st));
    INFCHK(h, hipMemcpyAsync(d_in_len, in_len, (size_t)n_members * 4, hipMemcpyHostToDevice, st));
    INFCHK(h, hipMemcpyAsync(d_out_len, out_len, (size_t)n_members * 4, hipMemcpyHostToDevice, st));
    // 8 busy lanes per wavefront.  With the direct tables of the literal/length and the distance code (32 KB of LDS per wavefront: five
    // per CU, 10 k members in flight) a member takes ~20 % less time; without them (7.4 KB: twenty per CU) a batch beyond those 10 k
    // members still runs in one round -- 32 k members 58 against 127 ms.  BNS_INFLATE_LUT=0/1 forces one (measurement switch).
    bool lut = n_members <= (u64)h->n_cu * 5u * 8u;
    if (const char *e = getenv("BNS_INFLATE_LUT")) lut = atoi(e) != 0;
    // Busy lanes per wavefront.  The members of a wavefront are in different states at every step (one in a literal, one in a match, one
    // building a block's tables), and the wavefront executes the union: one member per wavefront runs 20 ms where eight take 30-34
    // (profiles/r05_inflate_mpw.txt).  So: as few members per wavefront as still leaves the batch resident at about four wavefronts per
    // SIMD -- 1 up to 4 k members, 2 up to 8 k, 4 up to 16 k, 8 beyond (16 / 32 / 64 lanes were measured too: no better, the chain of
    // ONE member bounds a batch, not instruction issue).  BNS_INFLATE_MPW overrides.
    u32 mpw = 8u;
    {
        const u64 per = (u64)h->n_cu * 16u;
        mpw = n_members <= per ? 1u : n_members <= 2 * per ? 2u : n_members <= 4 * per ? 4u : 8u;
    }
    if (const char *e = getenv("BNS_INFLATE_MPW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) mpw = (u32)v; }
    if (mpw > 8u) lut = false;
    const u64 blocks = (n_members + mpw - 1) / mpw;
    // The form.  One member per wavefront (bns_inflate_wave.hpp) is the faster one at every batch size measured (1 k members 4.4 ms
    // against 20, 4 k 42 GB/s against 13, 16 k 47 against 24: profiles/r05_inflate_wave.txt); the member-per-lane form stays for
    // comparison and as the second opinion of the tests (BNS_INFLATE_FORM=lane).
    bool wave_form = true;
    if (const char *e = getenv("BNS_INFLATE_FORM")) wave_form = e[0] != 'l';
    // (its stream position is a bit count in 32 bits: a member of 256 MiB of DEFLATE and more -- nothing BGZF can hold -- goes the other way)
    for (u64 i = 0; wave_form && i < n_members; ++i) if (in_len[i] >= (1u << 28)) wave_form = false;
    INFCHK(h, hipEventRecord(h->ev0, st));
    if (wave_form) {
        const u8 *ce = (const u8 *)h->d_comp.p + (((size_t)comp_bytes + 64) & ~(size_t)3);
        hipLaunchKernelGGL(inflate_wave_kernel, dim3((unsigned)n_members), dim3(64), 0, st, (const u8 *)h->d_comp.p, ce, (const u64 *)d_in_off, (const u32 *)d_in_len,
                           (const u64 *)d_out_off, (const u32 *)d_out_len, (u64)n_members, d_out, d_crc, d_status);
    } else
#define BNS_INF_LAUNCH(M, L)                                                                                                                           \
    hipLaunchKernelGGL((inflate_members_kernel<M, L>), dim3((unsigned)blocks), dim3(64), 0, st, (const u8 *)h->d_comp.p, (const u64 *)d_in_off,         \
                       (const u32 *)d_in_len, (const u64 *)d_out_off, (const u32 *)d_out_len, (u64)n_members, d_out, (u8 *)h->d_scratch.p,              \
                       d_crc, d_status)
    if (mpw == 1u) { if (lut) BNS_INF_LAUNCH(1, true); else BNS_INF_LAUNCH(1, false); }
    else if (mpw == 2u) { if (lut) BNS_INF_LAUNCH(2, true); else BNS_INF_LAUNCH(2, false); }
    else if (mpw == 4u) { if (lut) BNS_INF_LAUNCH(4, true); else BNS_INF_LAUNCH(4, false); }
    else if (mpw == 64u) BNS_INF_LAUNCH(64, false);
    else if (mpw == 32u) BNS_INF_LAUNCH(32, false);
    else if (mpw == 16u) BNS_INF_LAUNCH(16, false);
    else if (lut) BNS_INF_LAUNCH(8, true);
    else BNS_INF_LAUNCH(8, false);
#undef BNS_INF_LAUNCH
    INFCHK(h, hipGetLastError());
    INFCHK(h, hipEventRecord(h->ev1, st));
    if (!d_text_out) INFCHK(h, hipMemcpyAsync(text, h->d_text.p, (size_t)text_bytes, hipMemcpyDeviceToHost, st));
    INFCHK(h, hipMemcpyAsync(crc32, d_crc, (size_t)n_members * 4, hipMemcpyDeviceToHost, st));
    INFCHK(h, hipMemcpyAsync(status, d_status, (size_t)n_members * 4, hipMemcpyDeviceToHost, st));
    INFCHK(h, hipEventRecord(h->done, st));
    INFCHK(h, hipEventSynchronize(h->done));
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_kernel_ms = ms;
    return BNS_OK;
}

extern "C" {

int bns_inflate_members(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                        const uint64_t *out_off, const uint32_t *out_len, uint64_t n_members, uint8_t *text, uint64_t text_bytes,
                        uint32_t *crc32, uint32_t *status)
{
    if (n_members == 0) return h ? BNS_OK : BNS_ERR_ARG;       // (an empty batch is nothing to do, whatever the pointers: as before the wave form)
    if (!text) return BNS_ERR_ARG;
    return inflate_members_impl(h, comp, comp_bytes, in_off, in_len, out_off, out_len, n_members, text, nullptr, text_bytes, crc32, status);
}

int bns_inflate_members_device(bns_inflater *h, const uint8_t *comp, uint64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                               const uint64_t *out_off, const uint32_t *out_len, uint64_t n_members, void *d_text, uint64_t text_bytes,
                               uint32_t *crc32, uint32_t *status)
{
    if (n_members == 0) return h ? BNS_OK : BNS_ERR_ARG;
    if (!d_text) return BNS_ERR_ARG;
    return inflate_members_impl(h, comp, comp_bytes, in_off, in_len, out_off, out_len, n_members, nullptr, (uint8_t *)d_text, text_bytes, crc32, status);
}

}  // extern "C"

#include "bns_gzstream.hip"
