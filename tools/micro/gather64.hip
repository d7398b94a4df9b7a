// Random 64-byte bucket fetch rate on gfx950 (what a half-size bucket for plain-hashed keys would be bounded by).
// Each wave fetches 32 buckets per "pass" as two 1 KiB loads (lane l reads 16-byte chunk l&3 of bucket l>>2 / +16); `depth` = passes in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned u32;
__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
template <int DEPTH, int PER_PASS>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ tab, u32 bucket_mask, int passes, u32 *out)
{
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    u32 acc = 0, ctr = wave * 7919u;
    for (int p = 0; p < passes; p += DEPTH) {
        uint4 v[DEPTH][2];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 slot = (lane >> 2) + 16u * h;
                const u32 b = mix((ctr + d) * 32u + slot) & bucket_mask;
                if (slot < (u32)PER_PASS) v[d][h] = tab[(u64)b * 4 + (lane & 3u)];
                else v[d][h] = make_uint4(0, 0, 0, 0);
            }
        }
        ctr += DEPTH;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d][0].x ^ v[d][1].y;
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int DEPTH, int PER_PASS>
static void run(const uint4 *tab, u32 mask, u32 *out, int blocks, const char *tag, double gb)
{
    const int passes = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<DEPTH, PER_PASS>), dim3(blocks), dim3(256), 0, 0, tab, mask, 16, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((gather<DEPTH, PER_PASS>), dim3(blocks), dim3(256), 0, 0, tab, mask, passes, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fetches = (double)blocks * 4 * passes * PER_PASS;
    printf("%-10s table %6.1f GB depth %d buckets/pass %2d: %7.2f G fetch/s  %7.1f GB/s  (%.2f ms)\n", tag, gb, DEPTH, PER_PASS, fetches / ms / 1e6, fetches * 64 / ms / 1e6, ms);
}
int main(int argc, char **argv)
{
    const int max_lg = argc > 1 ? atoi(argv[1]) : 30;                  // log2 buckets of the largest table (30 -> 68.7 GB)
    uint4 *tab; u32 *out;
    const size_t bytes = (size_t)64 << max_lg;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 1, bytes); hipMalloc(&out, 64);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int blocks = pr.multiProcessorCount * 8;
    for (int lg = 24; lg <= max_lg; lg += 3) {
        const u32 mask = (1u << lg) - 1u; const double gb = 64.0 * (1ull << lg) / 1e9;
        run<1, 32>(tab, mask, out, blocks, "d1", gb);
        run<2, 32>(tab, mask, out, blocks, "d2", gb);
        run<4, 32>(tab, mask, out, blocks, "d4", gb);
    }
    return 0;
}
