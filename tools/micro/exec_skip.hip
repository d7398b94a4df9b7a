// Does a VALU instruction cost less when part of the wavefront is masked off?  gfx950, 8 waves/SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -o exec_skip tools/micro/exec_skip.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define KERNEL(name, body)                                                                              \
    __global__ __launch_bounds__(256) void k_##name(unsigned *out, int iters, unsigned long long mask)  \
    {                                                                                                   \
        unsigned a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = 0x9E3779B1u, d = 7u;                  \
        unsigned long long q = ((unsigned long long)a << 32) | b;                                       \
        unsigned long long saved;                                                                       \
        asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));              \
        for (int i = 0; i < iters; ++i) { R64(body) }                                                   \
        asm volatile("s_mov_b64 exec, %0" : : "s"(saved));                                              \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (unsigned)q;                              \
    }
KERNEL(add,     asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(mul_lo,  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));)
KERNEL(lshl_b64, asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q) : "v"(d));)
KERNEL(cndmask, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b));)
KERNEL(add3,    asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
struct T { const char *name; void (*fn)(unsigned *, int, unsigned long long); };
#define E(n) { #n, k_##n }
int main()
{
    std::vector<T> tests = { E(add), E(mul_lo), E(lshl_b64), E(cndmask), E(add3) };
    struct M { const char *name; unsigned long long m; } masks[] = {
        {"all64", ~0ULL}, {"low32", 0xFFFFFFFFULL}, {"low16", 0xFFFFULL}, {"low8", 0xFFULL}, {"lane0", 1ULL},
        {"8+8(0,32)", 0xFF000000FFULL}, {"1per16", 0x0001000100010001ULL}, {"high16", 0xFFFF000000000000ULL}, {"mid32", 0x0000FFFFFFFF0000ULL}};
    unsigned *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned) * 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount, iters = 8000;
    const double clk = pr.clockRate * 1e3;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_add, dim3(n_cu * 8), dim3(256), 0, 0, out, iters, ~0ULL);
    hipDeviceSynchronize();
    printf("cycles per wave-instruction per SIMD at 8 waves/SIMD, by exec mask\n%-10s", "");
    for (auto &m : masks) printf(" %10s", m.name);
    printf("\n");
    for (auto &t : tests) {
        printf("%-10s", t.name);
        for (auto &m : masks) {
            hipLaunchKernelGGL(t.fn, dim3(n_cu * 8), dim3(256), 0, 0, out, 10, m.m);
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.fn, dim3(n_cu * 8), dim3(256), 0, 0, out, iters, m.m);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %10.2f", ms * 1e-3 * clk / ((double)iters * 64 * 8));
        }
        printf("\n");
    }
    return 0;
}
