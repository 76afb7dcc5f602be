// cobs_amd/csrc/cobs_gpu_query.cpp -- command-line caller of the GPU query path with
// the flags and output format of the reference's `cobs query` sub-tool
// (reference src/cobs.cpp:471-527 flags, :410-469 query-file parsing and output):
//
//   cobs_gpu_query -i IDX [-i IDX2 ...] [-t 0.8] [-l N] (QUERY | -f QUERYFILE)
//
// Output: "doc_name<TAB>score" per hit; in file mode each query is preceded by
// "*comment<TAB>number-of-hits".  Lines starting with '>' or ';' delimit queries,
// sequence lines are concatenated.  Default threshold 0.8 (src/cobs.cpp:481-484).
// Unlike the reference, which runs the queries of a file one after the other, the
// whole file is one device batch.
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/cobs_gpu_batch.h"          // cobs_gpu_write_synthetic (the generator sub-tool)
#include "../../include/cobs_gpu_construct.h"
#include "../../include/cobs_gpu_search.hpp"

static void usage() {
    std::fprintf(stderr,
                 "usage: cobs_gpu_query -i INDEX [-i INDEX ...] [-t THRESHOLD] [-l LIMIT] "
                 "[-d DEVICE[,DEVICE...]] [--hbm-budget GIB] (QUERY | -f QUERY_FILE)\n"
                 "       -d 0,1,2,3: the index is sharded by sub-index block over the listed GPUs, every search is\n"
                 "        one scan per GPU + one RCCL exchange (same results as on one GPU)\n"
                 "       (--load-complete and -T/--threads of `cobs query` are accepted and ignored: the index\n"
                 "        always lives in HBM, or is streamed through it under --hbm-budget)\n"
                 "       cobs_gpu_query classic-construct | compact-construct | classic-combine | compact-construct-combine ...\n"
                 "        (the construction sub-tools of `cobs`, same arguments; see cobs_gpu_tools.cpp)\n"
                 "       cobs_gpu_query --benchmark -i INDEX [-k KMERS] [-q QUERIES] [-w WARMUP] [--seed S] [--dist]\n"
                 "       cobs_gpu_query benchmark-fpr INDEX [-k KMERS] [-q QUERIES] [-w WARMUP] [-d|--dist] [--seed S] [--device N[,M..]]\n"
                 "        (`cobs benchmark-fpr`, same flags: -d / --dist adds the distribution of all scores,\n"
                 "         RESULT name=benchmark_fpr fpr=<score> dist=<count> lines)\n"
                 "       cobs_gpu_query --write-synthetic OUT (--classic -n DOCS -s ROWS | --compact -n DOCS -p PAGE_SIZE\n"
                 "                      -s ROWS_0,ROWS_1,...) [--num-hashes H] [--seed S] [-d DEVICE]\n"
                 "        (a random-bit index file, density 0.3, for benchmarks of any size)\n"
                 "       cobs_gpu_query --construct-random OUT [-s SIGNATURE_SIZE] [-n DOCS] [-m DOCUMENT_SIZE]\n"
                 "                      [--num-hashes H] [--seed S]    (`cobs classic-construct-random`, same flags and defaults)\n");
}

// `cobs benchmark-fpr` (reference src/cobs.cpp:605-730): random ACGT queries of
// num_kmers + 30 characters from one std::mt19937(seed), default threshold 0 and
// num_results 0, one RESULT line.  Here the queries run as one device batch; the
// reference's t_io / t_and / t_add phases are one scan kernel (t_scan).
// --dist (src/cobs.cpp:627-632, 664-670): the reference tallies counts[r.score]++ over every result of every query on
// the host; here every shard's score rows are tallied on the device (cobs_gpu_batch_score_histogram), pass by pass.
static bool score_distribution(const std::vector<cobs_gpu_index*>& shards, const std::vector<std::string>& queries,
                               unsigned num_kmers, std::vector<uint64_t>& hist) {
    hist.assign((size_t)num_kmers + 31, 0);
    for (cobs_gpu_index* ix : shards) {
        const uint64_t slots = std::max<uint64_t>(cobs_gpu_local_counts(ix), 1);
        const size_t per_pass = (size_t)std::max<uint64_t>(1, (4ull << 30) / (slots * 4));      // <= 4 GiB of score rows per pass
        cobs_gpu_batch* b = nullptr;
        if (cobs_gpu_batch_create(ix, 0, 0, &b) != COBS_GPU_OK) return false;
        bool ok = true;
        for (size_t q0 = 0; q0 < queries.size() && ok; q0 += per_pass) {
            const size_t n = std::min(per_pass, queries.size() - q0);
            std::vector<const char*> qp(n);
            std::vector<size_t> ql(n);
            for (size_t i = 0; i < n; ++i) { qp[i] = queries[q0 + i].data(); ql[i] = queries[q0 + i].size(); }
            size_t bad = 0;
            ok = cobs_gpu_batch_set_queries(b, qp.data(), ql.data(), n) == COBS_GPU_OK &&
                 cobs_gpu_batch_run(b, 0.0, nullptr) == COBS_GPU_OK && cobs_gpu_batch_sync(b, nullptr, &bad) == COBS_GPU_OK &&
                 cobs_gpu_batch_score_histogram(b, hist.data(), hist.size()) == COBS_GPU_OK;
        }
        cobs_gpu_batch_destroy(b);
        if (!ok) return false;
    }
    return true;
}

static int benchmark(cobs_gpu::BatchSearch& s, const std::string& index, unsigned num_kmers,
                     unsigned num_queries, unsigned num_warmup, size_t seed, bool dist = false) {
    static const char basepairs[4] = {'A', 'C', 'G', 'T'};
    std::mt19937 rng(seed);
    auto make = [&](unsigned n) {
        std::vector<std::string> v(n);
        for (auto& q : v) {
            q.resize(num_kmers + 30);
            for (auto& c : q) c = basepairs[rng() % 4];
        }
        return v;
    };
    std::vector<std::string> warm = make(num_warmup), queries = make(num_queries);
    std::vector<std::vector<cobs_gpu::SearchResult>> results;
    if (!warm.empty()) s.search_batch(warm, results);
    s.timer().reset();                       // (as benchmark_fpr_run does, src/cobs.cpp:623)
    const auto t0 = std::chrono::steady_clock::now();
    s.search_batch(queries, results);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const cobs_gpu::Timer t = s.timer();     // a copy is a snapshot (:644)
    // the reference's line (src/cobs.cpp:645-661) with its own keys -- t_io is the scan kernel (gather + AND + count are
    // one kernel: t_and / t_add read 0), sse2 / aio do not apply -- and, behind them, the GPU path's own phases
    std::cout << "RESULT name=benchmark  index=" << index << " kmer_queries=" << num_kmers
              << " queries=" << num_queries << " warmup=" << num_warmup
              << " results=" << (results.empty() ? 0 : results.back().size()) << " sse2=off aio=off"
              << " t_hashes=" << t.get("hashes") << " t_io=" << t.get("io") << " t_and=" << t.get("and rows")
              << " t_add=" << t.get("add rows") << " t_sort=" << t.get("sort results") << " backend=gpu"
              << " t_scan=" << t.get("scan") << " t_h2d=" << t.get("h2d") << " t_d2h=" << t.get("d2h")
              << " t_rank=" << t.get("rank") << " t_total=" << wall << " queries_per_s=" << num_queries / wall
              << std::endl;
    if (dist) {
        results.clear();
        results.shrink_to_fit();
        std::vector<cobs_gpu_index*> shards;
        if (auto* sh = dynamic_cast<cobs_gpu::ShardedClassicSearch*>(&s)) shards = sh->shard_handles();
        else shards.push_back(s.handle());
        std::vector<uint64_t> hist;
        if (!score_distribution(shards, queries, num_kmers, hist)) {
            std::fprintf(stderr, "EXCEPTION: %s\n", cobs_gpu_last_error());
            return 1;
        }
        for (size_t sc = 0; sc < hist.size(); ++sc)      // std::map order: ascending scores, those that occur
            if (hist[sc]) std::cout << "RESULT name=benchmark_fpr fpr=" << sc << " dist=" << hist[sc] << std::endl;
    }
    return 0;
}

// `s.timer().print("search")` of the reference's process_query (src/cobs.cpp:468, cobs/util/timer.cpp:77-85):
// one "TIMER info=search name=seconds ... total=seconds" line on stderr.  The reference's phases are hashes / io /
// and rows (/ add rows); here: hashes (K1), h2d (query text), scan (K2: gather + AND + count), d2h, rank.
static void print_timer(const cobs_gpu::BatchSearch& s) { s.timer().print("search"); }

int cobs_gpu_tools_main(int argc, char** argv);      // cobs_gpu_tools.cpp: *-construct, classic-combine, compact-construct-combine

int main(int argc, char** argv) {
    {
        const int rc = cobs_gpu_tools_main(argc, argv);
        if (rc >= 0) return rc;
        if (argc > 1 && std::string(argv[1]) == "query") { ++argv; --argc; }       // `cobs query ...`
    }
    // `cobs benchmark-fpr IN_FILE [-k N] [-q N] [-w N] [-d|--dist] [--seed S]` (src/cobs.cpp:672-730): the index is the
    // positional argument and -d means --dist, as there; the device list of this tool is spelled --device in this mode
    const bool fpr_mode = argc > 1 && std::string(argv[1]) == "benchmark-fpr";
    if (fpr_mode) { ++argv; --argc; }
    std::vector<std::string> index_paths;
    std::string query_line, query_file;
    double threshold = 0.8;
    size_t num_results = 0;
    std::vector<int> devices;
    uint64_t hbm_budget = 0;
    std::string synth_out, synth_rows, random_out;
    uint64_t document_size = 1000000;
    bool synth_compact = false, force_sharded = false;
    uint64_t synth_docs = 10000, synth_page = 0, synth_hashes = 1;
    bool bench = fpr_mode, dist = false;
    unsigned num_kmers = 1000, num_queries = 10000, num_warmup = 100;
    size_t seed = std::random_device{}();
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", what); usage(); std::exit(1); }
            return argv[++i];
        };
        if (a == "-i" || a == "--index") index_paths.push_back(need("-i"));
        else if (a == "-f" || a == "--file") query_file = need("-f");
        else if (a == "-t" || a == "--threshold") threshold = std::atof(need("-t"));
        else if (a == "-l" || a == "--limit") num_results = (size_t)std::strtoull(need("-l"), nullptr, 10);
        else if (a == "--dist" || (fpr_mode && a == "-d")) dist = true;
        else if (a == "-d" || a == "--device") {
            const std::string v = need("-d");
            for (size_t p = 0; p < v.size();) {
                size_t e = v.find(',', p);
                if (e == std::string::npos) e = v.size();
                devices.push_back(std::atoi(v.substr(p, e - p).c_str()));
                p = e + 1;
            }
        }
        else if (a == "--sharded") force_sharded = true;        // the multi-GPU code path even for one device
        else if (a == "--write-synthetic") synth_out = need("--write-synthetic");
        else if (a == "--construct-random") random_out = need("--construct-random");
        else if (a == "-m" || a == "--document-size") document_size = std::strtoull(need("-m"), nullptr, 10);
        else if (a == "--classic") synth_compact = false;
        else if (a == "--compact") synth_compact = true;
        else if (a == "-n" || a == "--num-documents") synth_docs = std::strtoull(need("-n"), nullptr, 10);
        else if (a == "-s" || a == "--signature-size") synth_rows = need("-s");
        else if (a == "-p" || a == "--page-size") synth_page = std::strtoull(need("-p"), nullptr, 10);
        else if (a == "--num-hashes") synth_hashes = std::strtoull(need("--num-hashes"), nullptr, 10);
        else if (a == "--hbm-budget") hbm_budget = (uint64_t)(std::atof(need("--hbm-budget")) * 1073741824.0);
        else if (a == "--load-complete") {}                  // reference flags without a meaning here
        else if (a == "-T" || a == "--threads") (void)need("-T");
        else if (a == "--benchmark") bench = true;
        else if (a == "-k" || a == "--num-kmers") num_kmers = (unsigned)std::atoi(need("-k"));
        else if (a == "-q" || a == "--queries") num_queries = (unsigned)std::atoi(need("-q"));
        else if (a == "-w" || a == "--warmup") num_warmup = (unsigned)std::atoi(need("-w"));
        else if (a == "--seed") seed = (size_t)std::strtoull(need("--seed"), nullptr, 10);
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); usage(); return 1; }
        else if (fpr_mode && index_paths.empty()) index_paths.push_back(a);
        else query_line = a;
    }
    const int device = devices.empty() ? -1 : devices[0];
    auto open_index = [&]() -> std::unique_ptr<cobs_gpu::BatchSearch> {
        // stdout carries the results in `cobs query` format and nothing else: RCCL prints a
        // version banner there when a communicator is created, so stdout points at stderr
        // while the index is opened
        struct StdoutToStderr {
            int saved;
            StdoutToStderr() { std::fflush(stdout); saved = dup(1); dup2(2, 1); }
            ~StdoutToStderr() { std::fflush(stdout); dup2(saved, 1); close(saved); }
        } quiet;
        if (devices.size() > 1 || force_sharded)
            return std::unique_ptr<cobs_gpu::BatchSearch>(new cobs_gpu::ShardedClassicSearch(
                index_paths, devices.empty() ? std::vector<int>{0} : devices, hbm_budget));
        return std::unique_ptr<cobs_gpu::BatchSearch>(new cobs_gpu::ClassicSearch(index_paths, device, hbm_budget));
    };
    if (!random_out.empty()) {
        // `cobs classic-construct-random` (reference src/cobs.cpp:243-291): same flags and defaults
        const uint64_t sig = synth_rows.empty() ? 2ull * 1024 * 1024 : std::strtoull(synth_rows.c_str(), nullptr, 10);
        if (cobs_gpu_construct_random(random_out.c_str(), sig, synth_docs, document_size, synth_hashes, seed, device) != COBS_GPU_OK) {
            std::fprintf(stderr, "EXCEPTION: %s\n", cobs_gpu_last_error());
            return 1;
        }
        return 0;
    }
    if (!synth_out.empty()) {
        std::vector<uint64_t> sigs;
        for (size_t p = 0; p < synth_rows.size();) {
            size_t e = synth_rows.find(',', p);
            if (e == std::string::npos) e = synth_rows.size();
            sigs.push_back(std::strtoull(synth_rows.substr(p, e - p).c_str(), nullptr, 10));
            p = e + 1;
        }
        if (sigs.empty()) sigs.push_back(2 * 1024 * 1024);      // default of `cobs classic-construct-random -s`
        cobs_gpu_synth d{};
        d.kind = synth_compact ? 1 : 0;
        d.term_size = 31;
        d.canonicalize = 1;
        d.num_pages = (uint32_t)sigs.size();
        d.num_hashes = synth_hashes;
        d.page_size = synth_page;
        d.num_docs = synth_docs;
        d.seed = seed;
        d.signature_sizes = sigs.data();
        if (cobs_gpu_write_synthetic(&d, synth_out.c_str(), device) != COBS_GPU_OK) {
            std::fprintf(stderr, "EXCEPTION: %s\n", cobs_gpu_last_error());
            return 1;
        }
        return 0;
    }
    if (bench && !index_paths.empty()) {
        try {
            std::unique_ptr<cobs_gpu::BatchSearch> sp = open_index();
            cobs_gpu::BatchSearch& s = *sp;
            return benchmark(s, index_paths[0], num_kmers, num_queries, num_warmup, seed, dist);
        } catch (const cobs_gpu::Error& e) {
            std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
            return 1;
        }
    }
    if (index_paths.empty() || (query_line.empty() && query_file.empty())) {
        if (!index_paths.empty()) std::fprintf(stderr, "Pass a verbatim query or a query file.\n");
        usage();
        return 1;
    }
    try {
        std::unique_ptr<cobs_gpu::BatchSearch> sp = open_index();
        cobs_gpu::BatchSearch& s = *sp;
        if (!query_line.empty()) {
            std::vector<cobs_gpu::SearchResult> result;
            s.search(query_line, result, threshold, num_results);
            for (const auto& r : result) std::cout << r.doc_name << '\t' << r.score << '\n';
            std::cout.flush();
            print_timer(s);
            return 0;
        }
        std::ifstream qf(query_file);
        if (!qf.good()) { std::fprintf(stderr, "could not open query file %s\n", query_file.c_str()); return 1; }
        std::vector<std::string> queries, comments;
        std::string line, query, comment;
        while (std::getline(qf, line)) {
            if (line.empty()) continue;
            if (line[0] == '>' || line[0] == ';') {
                if (!query.empty()) { queries.push_back(query); comments.push_back(comment); }
                line[0] = '*';
                query.clear();
                comment = line;
            } else {
                query += line;
            }
        }
        if (!query.empty()) { queries.push_back(query); comments.push_back(comment); }
        std::vector<std::vector<cobs_gpu::SearchResult>> results;
        s.search_batch(queries, results, threshold, num_results);
        for (size_t q = 0; q < queries.size(); ++q) {
            std::cout << comments[q] << '\t' << results[q].size() << '\n';
            for (const auto& r : results[q]) std::cout << r.doc_name << '\t' << r.score << '\n';
        }
        std::cout.flush();
        print_timer(s);
    } catch (const cobs_gpu::Error& e) {
        // the reference prints "EXCEPTION: ..." and returns -1 (src/cobs.cpp:1070-1076) or exits
        std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
        return 1;
    }
    return 0;
}
