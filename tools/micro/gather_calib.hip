// Calibration of the L2 memory-side read counters (TCC_EA0_RDREQ*, FETCH_SIZE) on gfx950 for the access shapes of this repo:
// three kernels with a KNOWN number of bytes that must come from HBM, each launched a few times, to be run under
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace
// bytes per request = known bytes / RDREQ.  (MI355X_MICROARCH.md: FETCH_SIZE = RDREQ x 64 B under-reports a wide coalesced
// read by 2x; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".)
//   calib_gather128  the minimizer-bucket probe's pattern exactly: per wave and pass 16 random 128-byte buckets as two
//                    coalesced 1 KiB global_load_lds_dwordx4 (lane l: 16-byte chunk l&7 of bucket l>>3), nt
//   calib_gather64   16 random 64-byte buckets per pass as one 1 KiB load (lane l: chunk l&3 of bucket l>>2)
//   calib_stream     float4 streaming read of the first `stream_bytes` of the table
// The table is 2^29 x 128 B = 68.7 GB (far beyond L2 + Infinity Cache), bucket ids come from a 32-bit mixer over a running
// counter, so repeats are negligible (2.1e8 draws from 5.4e8 buckets per launch: ~17 % of draws hit a bucket drawn earlier
// in the same launch, but 32 MiB of L2 + 256 MiB of Infinity Cache hold 2.4e6 lines -- 0.4 % of the table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned u32;
__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

__global__ __launch_bounds__(256) void calib_gather128(const uint4 *__restrict__ tab, u32 bucket_mask, int passes, u32 salt, u32 *out)
{
    __shared__ uint4 stage[4][128];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    u32 acc = 0, ctr = wave * 7919u + salt;
    for (int p = 0; p < passes; ++p, ++ctr) {
        const u32 b0 = mix(ctr * 16u + (lane >> 3)) & bucket_mask, b1 = mix(ctr * 16u + 8u + (lane >> 3)) & bucket_mask;
        __builtin_amdgcn_global_load_lds((gptr_t)(tab + ((u64)b0 * 8 + (lane & 7u))), (lptr_t)&stage[wv][0], 16, 0, 2);
        __builtin_amdgcn_global_load_lds((gptr_t)(tab + ((u64)b1 * 8 + (lane & 7u))), (lptr_t)&stage[wv][64], 16, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= stage[wv][lane].x ^ stage[wv][64 + lane].y;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather64(const uint4 *__restrict__ tab, u32 bucket_mask, int passes, u32 salt, u32 *out)
{
    __shared__ uint4 stage[4][64];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    u32 acc = 0, ctr = wave * 7919u + salt;
    for (int p = 0; p < passes; ++p, ++ctr) {
        const u32 b0 = mix(ctr * 16u + (lane >> 2)) & bucket_mask;          // 64-byte buckets: mask covers twice as many
        __builtin_amdgcn_global_load_lds((gptr_t)(tab + ((u64)b0 * 4 + (lane & 3u))), (lptr_t)&stage[wv][0], 16, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= stage[wv][lane].x;
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_stream(const uint4 *__restrict__ tab, u64 n_vec, u32 *out)
{
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (u64)gridDim.x * 256) acc ^= tab[i].x;
    if (acc == 0x12345u) out[0] = acc;
}
int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 29;
    const int passes = argc > 2 ? atoi(argv[2]) : 200;
    uint4 *tab; u32 *out;
    const size_t bytes = (size_t)128 << lg;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 1, bytes); hipMalloc(&out, 64);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * 8;
    const u64 stream_bytes = 16ull << 30;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(calib_gather128, dim3(blocks), dim3(256), 0, 0, tab, (1u << lg) - 1u, passes, 1000003u * rep, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double f128 = (double)blocks * 4 * passes * 16;
        printf("{\"kernel\": \"calib_gather128\", \"fetches\": %.0f, \"bytes\": %.0f, \"ms\": %.3f, \"Gfetch_s\": %.2f, \"GB_s\": %.1f}\n", f128, f128 * 128, ms, f128 / ms / 1e6, f128 * 128 / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(calib_gather64, dim3(blocks), dim3(256), 0, 0, tab, (2u << lg) - 1u, passes, 1000003u * rep, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double f64 = (double)blocks * 4 * passes * 16;
        printf("{\"kernel\": \"calib_gather64\", \"fetches\": %.0f, \"bytes\": %.0f, \"ms\": %.3f, \"Gfetch_s\": %.2f, \"GB_s\": %.1f}\n", f64, f64 * 64, ms, f64 / ms / 1e6, f64 * 64 / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(calib_stream, dim3(blocks), dim3(256), 0, 0, tab + (size_t)rep * (stream_bytes / 16), stream_bytes / 16, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("{\"kernel\": \"calib_stream\", \"fetches\": 0, \"bytes\": %.0f, \"ms\": %.3f, \"GB_s\": %.1f}\n", (double)stream_bytes, ms, stream_bytes / ms / 1e6);
    }
    return 0;
}
