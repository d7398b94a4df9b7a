// Test harness for the C++ Template-API shims of bns_host.hpp: compiled against libbns_host.so the way a caller
// of the reference's Encoder / RollingHasher would be -- bin/kmercnt.cpp:19-30, python/bns.cpp:42-86 -- and driven by
// tests/test_gpu_cpp_api.py.  Writes every value the functor receives, in order, as little-endian u64 (u128: lo, hi).
//   bns_api_check <mode> <path> <out.bin> k canon w score gaps|- [seed1 seed2]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
#include "../../bonsai_amd/csrc/host/bns_host.hpp"

struct kseq_t;                                   // callers of the reference pass one; the shims accept and ignore it

int main(int argc, char **argv)
{
    if (argc < 9) { std::fprintf(stderr, "usage: %s mode path out k canon w score gaps|- [seed1 seed2]\n", argv[0]); return 2; }
    const std::string mode = argv[1];
    const char *path = argv[2];
    const unsigned k = (unsigned)std::atoi(argv[4]);
    const bool canon = std::atoi(argv[5]) != 0;
    const long long w = std::atoll(argv[6]);
    const int score = std::atoi(argv[7]);
    bns::spvec_t gaps;
    if (std::strcmp(argv[8], "-") != 0) gaps = bns::parse_spacing(argv[8], k);
    const bns::u64 seed1 = argc > 9 ? std::strtoull(argv[9], nullptr, 10) : 1337, seed2 = argc > 10 ? std::strtoull(argv[10], nullptr, 10) : 137;
    std::vector<bns::u64> out;
    auto take = [&](bns::u64 v) { out.push_back(v); };
    auto take128 = [&](unsigned __int128 v) { out.push_back((bns::u64)v); out.push_back((bns::u64)(v >> 64)); };
    try {
        if (mode.rfind("enc_", 0) == 0) {
            bns::Encoder enc(k, gaps, canon, 0, w > 0 ? (unsigned)w : 0, score);
            kseq_t *ks = nullptr;
            if (mode == "enc_path") enc.for_each(take, path);
            else if (mode == "enc_path_ks") enc.for_each(take, path, ks);
            else if (mode == "enc_path_string") enc.for_each(take, std::string(path));
            else if (mode == "enc_paths") enc.for_each(take, std::vector<std::string>{path, path});
            else if (mode == "enc_canon_path") enc.for_each_canon(take, path);
            else if (mode == "enc_uncanon_path") enc.for_each_uncanon(take, path, ks);
            else if (mode == "enc_hash_path") enc.for_each_hash(take, path);
            else if (mode == "enc_str") {                          // the file's bytes as one string (the string overload's rules)
                std::ifstream f(path, std::ios::binary);
                const std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
                enc.for_each(take, s.data(), (bns::u64)s.size());
            } else { std::fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2; }
        } else if (mode.rfind("roll", 0) == 0) {
            const bool wide = mode.rfind("roll128", 0) == 0;
            const bool str = mode.find("_str") != std::string::npos;
            std::string s;
            if (str) { std::ifstream f(path, std::ios::binary); s.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
            if (!wide) {
                bns::RollingHasher<bns::u64> rh(k, canon, bns::DNA, w, seed1, seed2);
                if (str) rh.for_each_hash(take, s.data(), s.size());
                else if (mode == "roll64_path_each") rh.for_each(take, path);
                else if (mode == "roll64_path_flip") { rh.canonicalize(!canon); rh.for_each_hash(take, path); }
                else rh.for_each_hash(take, path);
            } else {
                bns::RollingHasher<unsigned __int128> rh(k, canon, bns::DNA, w, seed1, seed2);
                if (str) rh.for_each_hash(take128, s.data(), s.size());
                else rh.for_each_hash(take128, path);
            }
        } else { std::fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2; }
    } catch (const bns::Error &e) {
        std::fprintf(stderr, "bns::Error: %s\n", e.what());
        return 3;
    }
    std::FILE *f = std::fopen(argv[3], "wb");
    if (!f) return 4;
    if (!out.empty() && std::fwrite(out.data(), 8, out.size(), f) != out.size()) return 4;
    std::fclose(f);
    return 0;
}
