// VALU issue-rate microbenchmark for gfx950: time per instruction relative to v_add_u32, at 8 waves/SIMD.
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate tools/micro/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define KERNEL(name, body)                                                                              \
    __global__ __launch_bounds__(256) void k_##name(unsigned *out, int iters)                           \
    {                                                                                                   \
        unsigned a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = 0x9E3779B1u, d = 7u;                  \
        unsigned long long q = ((unsigned long long)a << 32) | b, r = q * 3;                            \
        __shared__ unsigned long long lds[512];                                                         \
        lds[threadIdx.x] = q; lds[256 + threadIdx.x] = r;                                               \
        unsigned la = (threadIdx.x & 63u) * 8u;                                                         \
        __syncthreads();                                                                                \
        for (int i = 0; i < iters; ++i) { R64(body) }                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (unsigned)q + (unsigned)r + la;           \
    }
KERNEL(add,        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(xor,        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(mul_lo,     asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(c));)
KERNEL(mul_hi,     asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(c));)
KERNEL(mul_u24,    asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(c));)
KERNEL(mad_u24,    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));)
KERNEL(mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q) : "v"(a), "v"(c) : "vcc");)
KERNEL(lshl_b64,   asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q) : "v"(d));)
KERNEL(lshr_b64,   asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q));)
KERNEL(lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q) : "v"(r));)
KERNEL(cmp_u64,    asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q), "v"(r) : "vcc");)
KERNEL(cmp_u32,    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");)
KERNEL(cndmask,    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");)
KERNEL(bfrev,      asm volatile("v_bfrev_b32 %0, %0" : "+v"(a));)
KERNEL(alignbit,   asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(b));)
KERNEL(perm,       asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(bitop3,     asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(bfi,        asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(min3,       asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(mbcnt,      asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));)
KERNEL(mov_dpp,    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a) : "v"(b));)
KERNEL(mov_dpp_q,  asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(a) : "v"(b));)
KERNEL(readlane,   asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a) : "s20");)
KERNEL(writelane,  asm volatile("v_writelane_b32 %0, s4, 3" : "+v"(a));)
KERNEL(readfirst,  asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a) : "s20");)
KERNEL(sdwa,       asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a));)
KERNEL(lshl_add,   asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(b));)
KERNEL(add3,       asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(bcnt,       asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a) : "v"(b));)
KERNEL(mov_b64,    asm volatile("v_mov_b64 %0, %1" : "+v"(q) : "v"(r));)
KERNEL(pk_add,     asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(ds_read64,  asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(la));)
KERNEL(ds_read64_nowait, asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(la)); )
KERNEL(ds_read32_nowait, asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(la)); )
KERNEL(ds_read128_nowait, { unsigned __attribute__((ext_vector_type(4))) t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(la)); a += 0; } )
KERNEL(salu,       asm volatile("s_add_u32 s20, s20, 3" : : : "s20", "scc");)
KERNEL(salu64,     asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc");)
KERNEL(add_salu,   asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 3" : "+v"(a) : "v"(b) : "s20", "scc");)
KERNEL(add_ds,     asm volatile("v_add_u32 %0, %0, %1\n ds_read_b32 %2, %3" : "+v"(a), "=v"(d) : "v"(b), "v"(la));)

KERNEL(and_,       asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(or_,        asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(sub,        asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(not_,       asm volatile("v_not_b32 %0, %0" : "+v"(a));)
KERNEL(mov,        asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));)
KERNEL(lshl32,     asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a));)
KERNEL(lshr32,     asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a));)
KERNEL(lshl32v,    asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(d));)
KERNEL(and_or,     asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(xad,        asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(add_lit,    asm volatile("v_add_u32 %0, 0x12345678, %0" : "+v"(a));)
KERNEL(and_lit,    asm volatile("v_and_b32 %0, 0x55555555, %0" : "+v"(a));)
KERNEL(add_sgpr,   asm volatile("v_add_u32 %0, s4, %0" : "+v"(a));)
KERNEL(add_e64,    asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a) : "v"(b));)
KERNEL(cnd_vcc,    asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));)
KERNEL(cnd_sgpr,   asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b));)
KERNEL(cmp_cnd,    asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");)
KERNEL(cmp_cnd_s,  asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");)
KERNEL(cmp_e64,    asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(a), "v"(b) : "s20", "s21");)
KERNEL(add2indep,  asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(d) : "v"(b));)
KERNEL(add_xor,    asm volatile("v_add_u32 %0, %0, %2\n v_xor_b32 %1, %1, %2" : "+v"(a), "+v"(d) : "v"(b));)
KERNEL(add_mul,    asm volatile("v_add_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(a), "+v"(d) : "v"(b));)
KERNEL(add_cmp64,  asm volatile("v_add_u32 %0, %0, %1\n v_cmp_lt_u64 vcc, %2, %3" : "+v"(a) : "v"(b), "v"(q), "v"(r) : "vcc");)
KERNEL(add_min3,   asm volatile("v_add_u32 %0, %0, %2\n v_min3_u32 %1, %1, %2, %0" : "+v"(a), "+v"(d) : "v"(b));)
KERNEL(bitop3_sgpr, asm volatile("v_bitop3_b32 %0, %0, %1, s4 bitop3:0x6c" : "+v"(a) : "v"(b));)
KERNEL(add_nc_co,  asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc");)
KERNEL(addc,       asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");)
KERNEL(ds_write32, asm volatile("ds_write_b32 %0, %1" : : "v"(la), "v"(a));)
KERNEL(ds_write8,  asm volatile("ds_write_b8 %0, %1" : : "v"(la), "v"(a));)
KERNEL(ds_read2_32, asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(q) : "v"(la));)
KERNEL(ballot_like, asm volatile("v_cmp_ne_u32_e64 s[20:21], 0, %0\n s_bcnt1_i32_b64 s22, s[20:21]" : : "v"(a) : "s20", "s21", "s22", "scc");)

struct T { const char *name; void (*fn)(unsigned *, int); };
#define E(n) { #n, k_##n }
int main()
{
    std::vector<T> tests = { E(add), E(xor), E(mul_lo), E(mul_hi), E(mul_u24), E(mad_u24), E(mad_u64_u32), E(lshl_b64), E(lshr_b64),
        E(lshl_add_u64), E(cmp_u64), E(cmp_u32), E(cndmask), E(bfrev), E(alignbit), E(perm), E(bitop3), E(bfi), E(min3), E(mbcnt),
        E(mov_dpp), E(mov_dpp_q), E(readlane), E(writelane), E(readfirst), E(sdwa), E(lshl_add), E(add3), E(bcnt), E(mov_b64), E(pk_add),
        E(ds_read64), E(ds_read64_nowait), E(ds_read32_nowait), E(ds_read128_nowait), E(salu), E(salu64), E(add_salu), E(add_ds),
        E(and_), E(or_), E(sub), E(not_), E(mov), E(lshl32), E(lshr32), E(lshl32v), E(and_or), E(xad), E(add_lit), E(and_lit), E(add_sgpr), E(add_e64), E(cnd_vcc), E(cnd_sgpr), E(cmp_cnd), E(cmp_cnd_s), E(cmp_e64), E(add2indep), E(add_xor), E(add_mul), E(add_cmp64), E(add_min3), E(bitop3_sgpr), E(add_nc_co), E(addc), E(ds_write32), E(ds_write8), E(ds_read2_32), E(ballot_like) };
    unsigned *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned) * 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount, iters = 8000;
    const double clk = pr.clockRate * 1e3;                     // Hz
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_add, dim3(n_cu * 8), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    printf("CUs %d clock %.0f MHz; cycles per wave-instruction per SIMD (8 waves/SIMD resident, 1 wave/SIMD)\n", n_cu, clk / 1e6);
    for (auto &t : tests) {
        double res[2];
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = mode == 0 ? n_cu * 8 : n_cu;    // 256 threads = 4 waves = 1 wave per SIMD
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 10);
            hipEventRecord(e0);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double waves_per_simd = mode == 0 ? 8 : 1;
            res[mode] = ms * 1e-3 * clk / ((double)iters * 64 * waves_per_simd);
        }
        printf("%-18s %7.2f %7.2f\n", t.name, res[0], res[1]);
    }
    return 0;
}
