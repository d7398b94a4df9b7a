// bonsai_main.cpp -- `bonsai classify` drop-in (bin/bonsai.cpp:107-163, :521-540) over the MI355X hot path.
#include <dlfcn.h>
#include <getopt.h>
#include <unistd.h>
#include <algorithm>

#include <chrono>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <string>
#include <vector>

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
                 "-p:\tHost threads for packing reads and formatting output [4, or as many CPUs as there are] (-1: all); the hot path runs on the GPU.\n"
                 "-S:\tper_set (accepted; meaningless without the pthread pool).\n"
                 "-C:\tDo not canonicalize.\n"
                 "-k/-K:\tEmit / do not emit kraken-style output.\n"
                 "-f/-F:\tEmit / do not emit fastq-style output.\n"
                 "Added:\n"
                 "-g:\tGPUs: an index, a range or a list (0, 0-7, 0,2,5, all) [0].  Several GPUs: the db is uploaded once and\n"
                 "\tbroadcast over xGMI (RCCL); every chunk of reads is sharded across them.\n"
                 "-L:\tTable layout in HBM: minbucket (default, minimizer-clustered 128 B buckets), bucket (hashed 64 B buckets)\n"
                 "\tor khash (probe the bns.db arrays as they are).\n"
                 "-P:\tParser threads for one plain (not gzip, not piped) input file [2]: stretches of the file are parsed side by side;\n"
                 "\tn:bytes sets the stretch length.  Output does not depend on it.\n"
                 "-N:\tDo not bind the host threads to the CPUs next to the GPU(s) (the default narrows the affinity mask to them).\n"
                 "-b:\tAlso write every read's (pair's) taxon, in input order, as raw little-endian u32 to this path.\n"
                 "<inr1.fq> may be a read container written by `bonsai pack` (2-bit reads + names): no parsing, no packing.\n",
                 exe, 1 << 24);
    std::exit(EXIT_FAILURE);
}

int classify_main(int argc, char *argv[])
{
    int co, num_threads = std::min(4, bns::usable_cpus()), emit_kraken = 1, emit_fastq = 0, emit_all = 0, chunk_size = 1 << 24;
    bool chunk_given = false, bind_cpus = true;
    unsigned parser_threads = 2;
    unsigned long long segment_bytes = 0;
    std::string devices = "0";
    int layout = BNS_LAYOUT_MINBUCKET;
    bool canonicalize = true;
    std::FILE *ofp = stdout, *taxon_fp = nullptr;
    if (argc < 4) usage(argv[0]);
    while ((co = getopt(argc, argv, "Cc:p:o:S:afFkKg:L:NP:b:h?")) >= 0) {
        switch (co) {
            case 'h': case '?': usage(argv[0]); break;
            case 'C': canonicalize = false; break;
            case 'a': emit_all = 1; break;
            case 'c': chunk_size = std::atoi(optarg); chunk_given = true; break;
            case 'F': emit_fastq = 0; break;
            case 'f': emit_fastq = 1; break;
            case 'K': emit_kraken = 0; break;
            case 'k': emit_kraken = 1; break;
            case 'p': num_threads = std::atoi(optarg); if (num_threads < 0) num_threads = bns::usable_cpus(); break;
            case 'o': ofp = std::fopen(optarg, "w"); break;
            case 'b': taxon_fp = std::fopen(optarg, "wb"); if (!taxon_fp) { std::fprintf(stderr, "Could not open taxon file\n"); return EXIT_FAILURE; } break;
            case 'S': break;
            case 'g': devices = optarg; break;
            case 'N': bind_cpus = false; break;
            case 'P': {
                char *end = nullptr;
                parser_threads = (unsigned)std::max(1L, std::strtol(optarg, &end, 10));
                if (end && *end == ':') segment_bytes = std::strtoull(end + 1, nullptr, 10);
                break;
            }
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
    const auto t_start = std::chrono::steady_clock::now();
    try {
        // (never destroyed, like the classifier below: giving 6.6 GB of arrays back page by page was 0.3 s between the last read and the exit)
        bns::Database &db = *new bns::Database(argv[optind]);
        const auto t_db = std::chrono::steady_clock::now();
        const std::vector<bns::u32> taxmap = bns::build_parent_map(argv[optind + 1]);
        const auto t_tax = std::chrono::steady_clock::now();
        // -g 0-7 / 0,2 / all: one context per GPU, db broadcast over xGMI, whole chunks (2^24 bases unless -c says otherwise) dealt
        // to the devices as they become free
        const std::vector<int> devs = bns::parse_devices(devices.c_str());
        (void)chunk_given;
        if (bind_cpus) {
            const int n = bns::bind_near_devices(devs);
            if (n && std::getenv("BNS_CLI_TIMING")) std::fprintf(stderr, "[timing] threads bound to the %d CPUs next to the GPU(s)\n", n);
        }
        // will the device inflate the input?  (decided here: the waits' mode below must be set before the first context exists)
        bool dev_inflate = false;
        {
            const char *g = std::getenv("BNS_BGZF_GPU");
            unsigned long long bgzf_bytes = 0;
            bool any_bgzf = false;
            for (int a = 2; a < npos; ++a)
                if (bns::is_bgzf_file(argv[optind + a])) {
                    any_bgzf = true;
                    if (std::FILE *f = std::fopen(argv[optind + a], "rb")) { if (std::fseek(f, 0, SEEK_END) == 0) bgzf_bytes += (unsigned long long)std::ftell(f); std::fclose(f); }
                }
            dev_inflate = any_bgzf && (g ? std::atoi(g) != 0 : (bns::usable_cpus() < 12 || bgzf_bytes >= (8ull << 30)));
        }
        // A host thread that waits for the device spins by default (lowest latency) -- one CPU per caller thread for as long as a
        // call lasts, and a call lasts longer when inflate kernels share the device.  When the device inflates the input (below), the
        // waits block instead (the runtime's own switch, set before the first context exists): BGZF +3-6 % on 4-12 CPUs; plain input
        // keeps the spinning waits (blocking: 0 to -7 %).  BNS_BLOCKING_SYNC=0 / 1 by hand.
        {
            const char *e = std::getenv("BNS_BLOCKING_SYNC");
            const bool blocking = e ? std::atoi(e) != 0 : dev_inflate;
            if (blocking) {
                using set_dev_t = int (*)(int);
                using set_flags_t = int (*)(unsigned);
                const auto set_dev = reinterpret_cast<set_dev_t>(::dlsym(RTLD_DEFAULT, "hipSetDevice"));
                const auto set_flags = reinterpret_cast<set_flags_t>(::dlsym(RTLD_DEFAULT, "hipSetDeviceFlags"));
                if (set_dev && set_flags)
                    for (int dv : devs) { (void)set_dev(dv); (void)set_flags(0x4u /* hipDeviceScheduleBlockingSync */); }
            }
        }
        // (never destroyed: the process leaves through _exit when the subcommand returns, and freeing the table, the page-locked
        // buffers and the contexts one by one first was 0.2 s of a 1.5 s run)
        bns::ClassifierGeneric &c = *new bns::ClassifierGeneric(db, taxmap, devs, num_threads, emit_all, emit_fastq, emit_kraken,
                                                                canonicalize, layout);
        c.taxon_out_ = taxon_fp;
        if (devs.size() > 1) {                                   // which collective library replicated the db over how many devices
            // (one line per device: a multi-GPU record says what it ran on)
            for (size_t i = 0; i < devs.size(); ++i) {
                char pci[64] = "?";
                (void)bns_device_pci_bus_id(devs[i], pci, (int)sizeof(pci));
                std::string low(pci), numa = "?";
                for (char &ch : low) ch = (char)std::tolower((unsigned char)ch);
                if (std::FILE *nf = std::fopen(("/sys/bus/pci/devices/" + low + "/numa_node").c_str(), "r")) {
                    char nb[32] = {0};
                    if (std::fgets(nb, sizeof(nb), nf)) { numa = nb; while (!numa.empty() && (numa.back() == '\n' || numa.back() == ' ')) numa.pop_back(); }
                    std::fclose(nf);
                }
                std::fprintf(stderr, "context %zu of %zu: device %d (PCI %s, NUMA node %s)\n", i, devs.size(), devs[i], pci, numa.c_str());
            }
            int ver = 0, ranks = 0;
            (void)bns_rccl_info(&ver, &ranks);
            if (ranks) std::fprintf(stderr, "%zu devices: db broadcast by RCCL %d.%d.%d over %d ranks (one upload, xGMI); reads sharded, no collective inside classification\n",
                                    devs.size(), ver / 10000, (ver / 100) % 100, ver % 100, ranks);
            else std::fprintf(stderr, "%zu devices: every device built its table from the host arrays (db too large to replicate array by array, or contexts on one device)\n", devs.size());
        }
        if (std::getenv("BNS_CLI_TIMING"))
            std::fprintf(stderr, "[timing] start-up (db + taxonomy read, context, table load) %.3f s = db file %.3f + nodes.dmp %.3f + contexts, table and taxonomy on the device %.3f\n",
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), std::chrono::duration<double>(t_db - t_start).count(),
                         std::chrono::duration<double>(t_tax - t_db).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tax).count());
        // Blocked-gzip input is inflated on the first device as well (one member per lane; batches taken from the back of the reader's
        // task queue beside the CPU inflaters, or from the front without them) when that pays: on a host short of CPUs (4 CPUs: 3.9 ->
        // 13-22 M reads/s, the device alone; 12: 13 -> 16 M), and on any host when the input is large -- a dozen CPU inflaters are
        // 22 M reads/s, with the device beside them 30-34 M on a 96 M-read file, but its start-up (page-locked staging, the first
        // batches) makes a 32 M-read file a draw (profiles/r04_bgzf_gpu.txt, r04_bgzf_cpus.txt).  BNS_BGZF_GPU=0 / 1 decides it by hand.
        if (dev_inflate && !devs.empty()) bns::set_bgzf_device(devs[0]);
        const auto t_pd = std::chrono::steady_clock::now();
        bns::process_dataset(c, argv[optind + 2], npos == 4 ? argv[optind + 3] : nullptr, ofp, (unsigned)chunk_size, parser_threads, segment_bytes);
        if (std::getenv("BNS_CLI_TIMING"))
            std::fprintf(stderr, "[timing] process_dataset %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pd).count());
        std::fprintf(stderr, "Classified %llu, unclassified %llu\n", (unsigned long long)c.n_classified(),
                     (unsigned long long)c.n_unclassified());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[E] %s\n", e.what());
        return EXIT_FAILURE;
    }
    if (ofp != stdout) std::fclose(ofp);
    if (taxon_fp) std::fclose(taxon_fp);
    if (std::getenv("BNS_CLI_TIMING"))
        std::fprintf(stderr, "[timing] since start %.3f s\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    std::fprintf(stderr, "Successfully completed classify!\n");
    return EXIT_SUCCESS;
}

// `bonsai pack`: FASTA / FASTQ (plain or gzip, one file or a pair) -> the pre-packed read container `bonsai classify` takes as it is
// (format: bns_host.hpp).  Host only: no GPU, no db.
int pack_main(int argc, char *argv[])
{
    int co, threads = std::min(8, bns::usable_cpus());
    unsigned parser_threads = 2, chunk = 0;
    bool names = true;
    std::string out;
    while ((co = getopt(argc, argv, "o:c:p:P:nh?")) >= 0) {
        switch (co) {
            case 'o': out = optarg; break;
            case 'c': chunk = (unsigned)std::strtoul(optarg, nullptr, 10); break;
            case 'p': threads = std::atoi(optarg); if (threads < 1) threads = bns::usable_cpus(); break;
            case 'P': parser_threads = (unsigned)std::max(1, std::atoi(optarg)); break;
            case 'n': names = false; break;
            default:
                std::fprintf(stderr, "Usage: %s pack [-o out.bnsp] [-c bases per chunk (2^27)] [-p pack threads] [-P parser threads] [-n: no read names] <in1.fq> [<in2.fq>]\n", argv[0]);
                return EXIT_FAILURE;
        }
    }
    const int npos = argc - optind;
    if ((npos != 1 && npos != 2) || out.empty()) {
        std::fprintf(stderr, "Usage: %s pack -o out.bnsp [-c bases per chunk] [-p threads] [-P parser threads] [-n] <in1.fq> [<in2.fq>]\n", argv[0]);
        return EXIT_FAILURE;
    }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        const auto r = bns::pack_dataset(argv[optind], npos == 2 ? argv[optind + 1] : nullptr, out.c_str(), chunk, parser_threads, threads, names);
        std::fprintf(stderr, "Packed %llu reads, %llu bases in %.2f s\n", (unsigned long long)r.first, (unsigned long long)r.second,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[E] %s\n", e.what());
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

// `bonsai build` (bin/bonsai.cpp:169-281, lex / entropy modes): db construction on the GPU.
void build_usage(const char *exe)
{
    std::fprintf(stderr,
                 "Usage: %s build <flags> <out.path> <ignored> <paths>\n"
                 "(positional arguments as bin/bonsai.cpp:214,227 reads them: the db is written to the first, the second is\n"
                 " not used in lex/entropy mode, genome paths start at the third unless -F is given)\nFlags:\n"
                 "-k: Set k. [31]\n-w: Set window size.\n-e: Use entropy maximization (score::Entropy) instead of score::Lex.\n"
                 "-S: Set spacing.\n-T: Set tax_path (nodes.dmp).\n-M: Set seq2taxpath (name<TAB>taxid).\n"
                 "-F: Load genome paths from file.\n-C: Do not canonicalize.\n-z: Write gzip-compressed.\n"
                 "-p: Threads (accepted; the build runs on the GPU).\n-g: GPU index [0].\n"
                 "-t / -f (tax-depth / feature-count minimisation) are not provided by this build.\n", exe);
    std::exit(EXIT_FAILURE);
}

int build_main(int argc, char *argv[])
{
    bns::BuildOptions opt;
    std::string spacing, tax_path, seq2taxpath, paths_file;
    bool gz = false;
    int c;
    if (argc < 4) build_usage(argv[0]);
    while ((c = getopt(argc, argv, "Cw:M:S:p:k:T:F:g:tefzHh?")) >= 0) {
        switch (c) {
            case 'C': opt.canon = false; break;
            case 'k': opt.k = (unsigned)std::atoi(optarg); break;
            case 'w': opt.wsz = std::atoi(optarg); break;
            case 'e': opt.entropy = true; break;
            case 'S': spacing = optarg; break;
            case 'T': tax_path = optarg; break;
            case 'M': seq2taxpath = optarg; break;
            case 'F': paths_file = optarg; break;
            case 'z': gz = true; break;
            case 'p': break;
            case 'g': opt.device = std::atoi(optarg); break;
            case 't': case 'f':
                // bin/bonsai.cpp:228 tests `LEX == mode || score_scheme::ENTROPY` -- always true -- and :260-261 then picks
                // lca_map<score::Entropy> for every mode but LEX: -t and -f DO an entropy lca build in the reference.
                opt.entropy = true;
                std::fprintf(stderr, "[W] -%c: the reference's tax-depth / feature-count branch is unreachable (bin/bonsai.cpp:228); "
                                     "building the entropy-minimized lca map it actually builds for this flag\n", c);
                break;
            default: build_usage(argv[0]);
        }
    }
    if (argc - optind < 1) build_usage(argv[0]);
    std::string dbpath = argv[optind];
    const bool has_gz = dbpath.size() > 3 && dbpath.compare(dbpath.size() - 3, 3, ".gz") == 0;
    if (has_gz) gz = true;
    if (gz && !has_gz) { dbpath += ".gz"; std::fprintf(stderr, "Writing gzipped, but without a .gz suffix. Adding it.\n"); }
    std::vector<std::string> inpaths;
    if (!paths_file.empty()) {
        std::FILE *fp = std::fopen(paths_file.c_str(), "r");
        if (!fp) { std::fprintf(stderr, "[E] Could not open %s\n", paths_file.c_str()); return EXIT_FAILURE; }
        char buf[4096];
        while (std::fgets(buf, sizeof(buf), fp)) {
            std::string l(buf);
            while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
            if (!l.empty()) inpaths.push_back(l);
        }
        std::fclose(fp);
    } else {
        for (int i = optind + 2; i < argc; ++i) inpaths.emplace_back(argv[i]);
    }
    try {
        if (inpaths.empty()) throw bns::Error("Need input files from command line or file. See usage.");
        if (seq2taxpath.empty()) throw bns::Error("seq2taxpath required for final database generation.");
        if (tax_path.empty()) throw bns::Error("Tax path required. [See -T option.]");
        if (opt.k < 1 || opt.k > 32) throw bns::Error("k must be in [1,32]");
        opt.spacing = bns::parse_spacing(spacing.c_str(), opt.k);
        const std::vector<bns::u32> taxmap = bns::build_parent_map(tax_path.c_str());
        std::fprintf(stderr, "Final map will be written to %s\n", dbpath.c_str());
        const bns::Database db = bns::lca_map(inpaths, taxmap, seq2taxpath.c_str(), opt);
        // spacing entries as 1 byte each for the plain file (what Database(const char*) reads, database.h:46-48) and
        // 2 bytes each for .gz (what the reference's gz writer emits, database.h:89); this reader takes both
        db.write(dbpath.c_str(), gz ? 2 : 1);
        std::fprintf(stderr, "Wrote %llu keys in %llu buckets\n", (unsigned long long)db.db_.size, (unsigned long long)db.db_.n_buckets);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "[E] %s\n", e.what());
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char *argv[])
{
    // Everything is written and closed when a subcommand returns; leaving through _exit skips the HIP runtime's exit handlers
    // (0.2-0.3 s of a 1.6 s run on 64 M reads).
    // (BNS_NORMAL_EXIT=1: through exit() -- a profiler that writes its trace from an exit handler needs it)
    auto leave = [](int rc) { std::fflush(stdout); std::fflush(stderr); if (std::getenv("BNS_NORMAL_EXIT")) std::exit(rc); _exit(rc); return rc; };
    // The HIP runtime spreads a process's streams over FOUR hardware queues unless told otherwise, and two streams on one queue take
    // turns: a context has three (kernels, uploads, results), the BGZF path two inflaters more, `-g 0,0` twice that -- an upload
    // stream that shares a queue with the kernels it feeds costs a third of the link rate (bench.py's text leg: 116 -> 171 M reads/s
    // with eight queues).  Before the first HIP call; the caller's own setting wins.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    if (argc > 1 && std::strcmp(argv[1], "classify") == 0) return leave(classify_main(argc - 1, argv + 1));
    if (argc > 1 && (std::strcmp(argv[1], "build") == 0 || std::strcmp(argv[1], "phase2") == 0 || std::strcmp(argv[1], "p2") == 0))
        return leave(build_main(argc - 1, argv + 1));                // bin/bonsai.cpp:527-529 aliases
    if (argc > 1 && std::strcmp(argv[1], "pack") == 0) return leave(pack_main(argc - 1, argv + 1));
    std::fprintf(stderr, "Usage: %s <classify|build|pack> ...\n"
                         "  classify <opts> <dbpath> <tax_path> <inr1.fq> [<inr2.fq>]\n"
                         "  build    <opts> <out.path> <ignored> <genome paths>\n"
                         "  pack     -o <out.bnsp> <in1.fq> [<in2.fq>]      (reads -> 2-bit container for classify)\n"
                         "Other reference subcommands (prebuild, hist, metatree) are out of scope (DESIGN.md).\n", argv[0]);
    return EXIT_FAILURE;
}
