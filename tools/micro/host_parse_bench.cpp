// Host FASTA/FASTQ parse throughput (no GPU work): records/s of bns::bseq_read over a file.
// build: g++ -O2 -std=c++17 -Ibonsai_amd/csrc/host -Iinclude tools/micro/host_parse_bench.cpp bonsai_amd/csrc/host/bns_host.o \
//        -o /tmp/host_parse_bench -Lbonsai_amd/lib -lbonsai_amd -lz -lpthread -Wl,-rpath,$PWD/bonsai_amd/lib -Wl,-rpath,/opt/rocm/lib
#include "bns_host.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
    using namespace bns;
    if (argc < 2) { std::fprintf(stderr, "usage: %s <file> [block_bytes]\n", argv[0]); return 2; }
    const auto t0 = std::chrono::steady_clock::now();
    SeqReader r(argv[1], argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 0);
    ReadChunk seqs;
    size_t n = 0, bases = 0;
    while (bseq_read(1 << 24, r, nullptr, seqs) > 0) { n += seqs.recs.size(); for (auto &b : seqs.recs) bases += b.seq.size(); }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%zu records %zu bases %.3f s  %.2f M reads/s\n", n, bases, dt, n / dt / 1e6);
    return 0;
}
