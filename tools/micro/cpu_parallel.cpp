// How many host cores does this box really give us?  memchr scan of a 64 MiB buffer split over T threads (T = 1, 2, 4, ...).
#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
int main(int argc, char **argv) {
    int T = atoi(argv[1]);
    size_t N = 64u << 20;
    std::vector<char> buf(N, 'A'); for (size_t i = 0; i < N; i += 80) buf[i] = '\n';
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th; std::vector<size_t> cnt(T * 16);
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (int rep = 0; rep < 8; ++rep) { const char *p = buf.data() + N * t / T, *e = buf.data() + N * (t + 1) / T; size_t c = 0; while ((p = (const char *)memchr(p, '\n', e - p))) { ++c; ++p; } cnt[t * 16] += c; } });
    for (auto &x : th) x.join();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("T=%d %.3f s %.2f GB/s\n", T, dt, 8.0 * N / dt / 1e9);
}
