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
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/cobs_gpu_search.hpp"

static void usage() {
    std::fprintf(stderr,
                 "usage: cobs_gpu_query -i INDEX [-i INDEX ...] [-t THRESHOLD] [-l LIMIT] "
                 "[-d DEVICE] (QUERY | -f QUERY_FILE)\n");
}

int main(int argc, char** argv) {
    std::vector<std::string> index_paths;
    std::string query_line, query_file;
    double threshold = 0.8;
    size_t num_results = 0;
    int device = -1;
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
        else if (a == "-d" || a == "--device") device = std::atoi(need("-d"));
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); usage(); return 1; }
        else query_line = a;
    }
    if (index_paths.empty() || (query_line.empty() && query_file.empty())) {
        if (!index_paths.empty()) std::fprintf(stderr, "Pass a verbatim query or a query file.\n");
        usage();
        return 1;
    }
    try {
        cobs_gpu::ClassicSearch s(index_paths, device);
        if (!query_line.empty()) {
            std::vector<cobs_gpu::SearchResult> result;
            s.search(query_line, result, threshold, num_results);
            for (const auto& r : result) std::cout << r.doc_name << '\t' << r.score << '\n';
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
    } catch (const cobs_gpu::Error& e) {
        // the reference prints "EXCEPTION: ..." and returns -1 (src/cobs.cpp:1070-1076) or exits
        std::fprintf(stderr, "EXCEPTION: %s\n", e.what());
        return 1;
    }
    return 0;
}
