// round 6: what page-locking costs, and whether pre-faulted (transparent huge page) memory registered with hipHostRegister is cheaper than
// hipHostMalloc -- the first 0.15 s of a BGZF file are five 96 MiB slots being page-locked one behind the other (profiles/r06_bgzf_trace.txt).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/pin_bench tools/micro/pin_bench.hip && tools/micro/bin/pin_bench [MiB=96]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)std::atol(argv[1]) : 96, n = mib << 20;
    CK(hipSetDevice(0));
    void *d = nullptr;
    CK(hipMalloc(&d, n));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; ++rep) {
        // (a) hipHostMalloc
        double t0 = now();
        void *p = nullptr;
        CK(hipHostMalloc(&p, n, hipHostMallocPortable));
        double t1 = now();
        std::memset(p, 1, n);
        double t2 = now();
        CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostFree(p));
        double t4 = now();
        std::printf("hipHostMalloc %zu MiB: alloc %.2f ms (%.3f ms/MiB), first touch %.2f ms, H2D %.2f ms (%.1f GB/s), free %.2f ms\n", mib, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / mib,
                    (t2 - t1) * 1e3, (t3 - t2) * 1e3, n / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
        // (b) mmap + huge-page advice + touch (4 threads), then hipHostRegister
        for (int huge = 0; huge < 2; ++huge) {
            t0 = now();
            void *q = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) { std::printf("mmap failed\n"); return 1; }
            if (huge) madvise(q, n, MADV_HUGEPAGE);
            std::vector<std::thread> th;
            for (int t = 0; t < 4; ++t) th.emplace_back([=] { std::memset((char *)q + n / 4 * t, 1, n / 4); });
            for (auto &x : th) x.join();
            t1 = now();
            CK(hipHostRegister(q, n, hipHostRegisterPortable));
            t2 = now();
            CK(hipMemcpyAsync(d, q, n, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            t3 = now();
            CK(hipHostUnregister(q));
            t4 = now();
            munmap(q, n);
            std::printf("mmap%s + touch on 4 threads %.2f ms, hipHostRegister %.2f ms (%.3f ms/MiB), H2D %.2f ms (%.1f GB/s), unregister %.2f ms\n", huge ? " (MADV_HUGEPAGE)" : "", (t1 - t0) * 1e3,
                        (t2 - t1) * 1e3, (t2 - t1) * 1e3 / mib, (t3 - t2) * 1e3, n / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
        }
        // (c) pageable memory, no registration
        {
            void *q = std::malloc(n); std::memset(q, 1, n);
            t0 = now();
            CK(hipMemcpy(d, q, n, hipMemcpyHostToDevice));
            t1 = now();
            std::printf("pageable hipMemcpy H2D %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
            std::free(q);
        }
    }
    // sustained copy rates (10 x the buffer, one stream): hipHostMalloc / registered 4 KiB pages / registered huge pages / both directions
    {
        void *p = nullptr; CK(hipHostMalloc(&p, n, hipHostMallocPortable)); std::memset(p, 1, n);
        void *q4 = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(q4, n, MADV_NOHUGEPAGE); std::memset(q4, 1, n);
        void *qh = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(qh, n, MADV_HUGEPAGE); std::memset(qh, 1, n);
        CK(hipHostRegister(q4, n, hipHostRegisterPortable)); CK(hipHostRegister(qh, n, hipHostRegisterPortable));
        const char *names[3] = {"hipHostMalloc", "registered, 4 KiB pages", "registered, huge pages"};
        void *bufs[3] = {p, q4, qh};
        for (int rep = 0; rep < 2; ++rep)
            for (int b = 0; b < 3; ++b)
                for (int dir = 0; dir < 2; ++dir) {
                    CK(hipStreamSynchronize(st));
                    const double t0 = now();
                    for (int i = 0; i < 10; ++i) CK(dir ? hipMemcpyAsync(bufs[b], d, n, hipMemcpyDeviceToHost, st) : hipMemcpyAsync(d, bufs[b], n, hipMemcpyHostToDevice, st));
                    CK(hipStreamSynchronize(st));
                    std::printf("%-26s %s 10 x %zu MiB: %.1f GB/s\n", names[b], dir ? "D2H" : "H2D", mib, 10.0 * n / (now() - t0) / 1e9);
                }
    }
    return 0;
}
