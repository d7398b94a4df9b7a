// bonsai_main.cpp -- `bonsai classify` drop-in (bin/bonsai.cpp:107-163, :521-540) over the MI355X hot path.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "bns_host.hpp"

namespace {

void usage(const char *exe)
{
    std::fprintf(stderr,
                 "Usage:\n%s classify [flags] <dbpath> <tax_path> <inr1.fq> [<inr2.fq>]\n"
                 "Flags (as bonsai classify):\n"
                 "-o:\tRedirect output to path instead of stdout.\n"
                 "-c:\tSet chunk size in bases per GPU batch. Default: %i\n"
                 "-a:\tEmit all records, not just classified.\n"
                 "-p:\tNumber of host threads (accepted; the hot path runs on the GPU).\n"
                 "-S:\tper_set (accepted; meaningless without the pthread pool).\n"
                 "-C:\tDo not canonicalize.\n"
                 "-k/-K:\tEmit / do not emit kraken-style output.\n"
                 "-f/-F:\tEmit / do not emit fastq-style output.\n"
                 "Added:\n"
                 "-g:\tGPU index [0].\n"
                 "-L:\tTable layout in HBM: minbucket (default, minimizer-clustered 128 B buckets), bucket (hashed 64 B buckets)\n"
                 "\tor khash (probe the bns.db arrays as they are).\n",
                 exe, 1 << 24);
    std::exit(EXIT_FAILURE);
}

int classify_main(int argc, char *argv[])
{
    int co, num_threads = 1, emit_kraken = 1, emit_fastq = 0, emit_all = 0, chunk_size = 1 << 24, device = 0;
    int layout = BNS_LAYOUT_MINBUCKET;
    bool canonicalize = true;
    std::FILE *ofp = stdout;
    if (argc < 4) usage(argv[0]);
    while ((co = getopt(argc, argv, "Cc:p:o:S:afFkKg:L:h?")) >= 0) {
        switch (co) {
            case 'h': case '?': usage(argv[0]); break;
            case 'C': canonicalize = false; break;
            case 'a': emit_all = 1; break;
            case 'c': chunk_size = std::atoi(optarg); break;
            case 'F': emit_fastq = 0; break;
            case 'f': emit_fastq = 1; break;
            case 'K': emit_kraken = 0; break;
            case 'k': emit_kraken = 1; break;
            case 'p': num_threads = std::atoi(optarg); break;
            case 'o': ofp = std::fopen(optarg, "w"); break;
            case 'S': break;
            case 'g': device = std::atoi(optarg); break;
            case 'L':
                if (std::strcmp(optarg, "khash") == 0) layout = BNS_LAYOUT_KHASH;
                else if (std::strcmp(optarg, "bucket") == 0) layout = BNS_LAYOUT_BUCKET;
                else if (std::strcmp(optarg, "minbucket") == 0) layout = BNS_LAYOUT_MINBUCKET;
                else usage(argv[0]);
                break;
        }
    }
    if (!ofp) { std::fprintf(stderr, "Could not open output file\n"); return EXIT_FAILURE; }
    const int npos = argc - optind;
    if (npos != 3 && npos != 4) usage(argv[0]);
    try {
        bns::Database db(argv[optind]);
        const std::vector<bns::u32> taxmap = bns::build_parent_map(argv[optind + 1]);
        bns::ClassifierGeneric c(db, taxmap, device, num_threads, emit_all, emit_fastq, emit_kraken, canonicalize, layout);
        bns::process_dataset(c, argv[optind + 2], npos == 4 ? argv[optind + 3] : nullptr, ofp, (unsigned)chunk_size);
        std::fprintf(stderr, "Classified %llu, unclassified %llu\n", (unsigned long long)c.n_classified(),
                     (unsigned long long)c.n_unclassified());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[E] %s\n", e.what());
        return EXIT_FAILURE;
    }
    if (ofp != stdout) std::fclose(ofp);
    std::fprintf(stderr, "Successfully completed classify!\n");
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char *argv[])
{
    if (argc > 1 && std::strcmp(argv[1], "classify") == 0) return classify_main(argc - 1, argv + 1);
    std::fprintf(stderr, "Usage: %s classify <opts> <dbpath> <tax_path> <inr1.fq> [<inr2.fq>]\n"
                         "Only the classify subcommand is provided by this build (see DESIGN.md, scope).\n", argv[0]);
    return EXIT_FAILURE;
}
