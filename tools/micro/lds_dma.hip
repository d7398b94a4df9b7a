// Semantics check of global_load_lds_dwordx4 on gfx950: where does lane i's 16 bytes land in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const uint4 *g, uint4 *out)
{
    __shared__ __attribute__((aligned(16))) uint4 stage[256];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256; i += 64) stage[i] = make_uint4(0xdead, 0, 0, 0);
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(g + lane * 3), (void __attribute__((address_space(3))) *)(stage + 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = stage[i];
}
int main()
{
    uint4 *g, *o; hipMalloc(&g, 64 * 3 * 16); hipMalloc(&o, 256 * 16);
    uint4 h[192]; for (int i = 0; i < 192; ++i) h[i] = make_uint4(i, i * 10, 7, 9);
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    uint4 r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; ++i) if (r[64 + i].x != (unsigned)(i * 3) || r[64 + i].y != (unsigned)(i * 30)) ok = 0;
    for (int i = 0; i < 64; ++i) if (r[i].x != 0xdead || r[128 + i].x != 0xdead) ok = 0;
    printf("lane-linear 16-byte placement at the given LDS address: %s  (stage[64].x=%u stage[65].x=%u stage[127].x=%u)\n", ok ? "YES" : "NO", r[64].x, r[65].x, r[127].x);
    return !ok;
}
