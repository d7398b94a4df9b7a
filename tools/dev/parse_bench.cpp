#include "../../bonsai_amd/csrc/host/bns_host.hpp"
#include <chrono>
#include <cstdio>
using namespace bns;
int main(int argc, char **argv) {
    for (int rep = 0; rep < 12; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        SeqReader r(argv[1]);
        ReadChunk c; size_t n = 0, bases = 0;
        while (bseq_read(argc > 2 ? atoi(argv[2]) : (1 << 24), r, nullptr, c) > 0) { n += c.recs.size(); for (auto &b : c.recs) bases += b.seq.size(); }
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%zu reads %zu bases %.3f s  %.1f M reads/s\n", n, bases, s, n / s / 1e6);
    }
}
