// Does structured buffer addressing (idxen, stride 128) reach a table larger than 4 GiB on gfx950, and with which NUM_RECORDS?
// Fills bucket b's 8 uint4 chunks with (b, chunk, ~b, 7) and reads random buckets back through
//   (1) buffer_load_dwordx4 ... idxen offen        (to VGPRs)
//   (2) buffer_load_dwordx4 ... idxen offen lds    (LDS DMA)
// for NUM_RECORDS = number of buckets and = 0xFFFFFFFF.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned u32;
typedef void __attribute__((address_space(3))) *lptr_t;
__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
__global__ void fill(uint4 *tab, u64 n_vec)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (u64)gridDim.x * 256) tab[i] = make_uint4((u32)(i >> 3), (u32)(i & 7), ~(u32)(i >> 3), 7u);
}
template <int MODE>
__global__ __launch_bounds__(64) void rd(const uint4 *tab, u32 mask, u32 num_records, u32 *bad)
{
    __shared__ uint4 stage[64];
    const u32 lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tab, (short)128, (int)num_records, 0x00020000);
    u32 errs = 0;
    for (int it = 0; it < 64; ++it) {
        const u32 b = mix(blockIdx.x * 64u + it * 977u + (lane >> 3)) & mask;
        uint4 v;
        if (MODE == 0) {
            v = tab[(u64)b * 8 + (lane & 7)];
        } else if (MODE == 1) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 idxen offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"((u64)b | ((u64)((lane & 7) * 16) << 32)), "s"(rsrc) : "memory");
        } else {
            __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (lptr_t)stage, 16, (int)b, (int)((lane & 7) * 16), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            v = stage[lane];
            __builtin_amdgcn_wave_barrier();
        }
        if (v.x != b || v.y != (lane & 7) || v.z != ~b || v.w != 7u) ++errs;
    }
    if (errs) atomicAdd(bad, errs);
}
int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 27;                       // 2^27 buckets = 16 GiB
    uint4 *tab; u32 *bad;
    const size_t bytes = (size_t)128 << lg;
    if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&bad, 4);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, tab, (u64)8 << lg);
    hipDeviceSynchronize();
    const u32 mask = (1u << lg) - 1u;
    for (int mode = 0; mode < 3; ++mode)
        for (u32 nr : {1u << lg, 0xFFFFFFFFu, (u32)(((u64)128 << lg) > 0xFFFFFFFFull ? 0xFFFFFFFFull : ((u64)128 << lg))}) {
            hipMemset(bad, 0, 4);
            if (mode == 0) hipLaunchKernelGGL(rd<0>, dim3(4096), dim3(64), 0, 0, tab, mask, nr, bad);
            if (mode == 1) hipLaunchKernelGGL(rd<1>, dim3(4096), dim3(64), 0, 0, tab, mask, nr, bad);
            if (mode == 2) hipLaunchKernelGGL(rd<2>, dim3(4096), dim3(64), 0, 0, tab, mask, nr, bad);
            u32 h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            printf("mode %d (0 flat, 1 buffer->vgpr, 2 buffer->lds) num_records %10u: %u wrong of %u\n", mode, nr, h, 4096u * 64u * 64u);
        }
    return 0;
}
