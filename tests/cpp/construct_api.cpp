// tests/cpp/construct_api.cpp -- the C++17 mirror of the reference's construction API
// (include/cobs_gpu_construct.hpp) used the way the reference's tests use theirs
// (tests/fasta_multifile.cpp:54-60, tests/cortex_file.cpp:22-52).  Host-only calls; prints what a
// Python test compares.  argv[1] = tests/golden/documents
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "cobs_gpu_construct.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string root = argv[1];
    try {
        cobs_gpu::DocumentList doc_list(root + "/fasta_multi");
        std::printf("fasta_multi %zu\n", doc_list.size());
        for (size_t i = 0; i < doc_list.size(); ++i) {
            const cobs_gpu::DocumentEntry de = doc_list[i];
            size_t count = 0;
            de.process_terms(31, [&](const char*) { ++count; });
            std::printf("%s %zu %zu %zu %zu\n", de.name_.c_str(), de.size_, de.subdoc_index_, de.num_terms(31), count);
        }
        cobs_gpu::DocumentList ctx;
        ctx.add(root + "/cortex/document.ctx");
        const cobs_gpu::DocumentEntry e = ctx[0];
        std::vector<std::string> kmers;
        e.process_terms(31, [&](const char* t) { kmers.emplace_back(t, 31); });
        std::sort(kmers.begin(), kmers.end());
        std::printf("cortex %s %zu %zu %zu %s %s\n", e.name_.c_str(), e.term_size_, e.term_count_, kmers.size(),
                    kmers.front().c_str(), kmers.back().c_str());
        cobs_gpu::DocumentList fq(root, cobs_gpu::StringToFileType("fastq"));
        fq.sort_by_size();
        for (size_t i = 0; i < fq.size(); ++i) std::printf("fastq %s %zu\n", fq[i].name_.c_str(), fq[i].size_);
        try {
            cobs_gpu::StringToFileType("nonsense");
            return 3;
        } catch (const cobs_gpu::Error& err) {
            std::printf("error %d %s\n", (int)err.status, err.what());
        }
        try {      // the reference refuses an output name without the index extension (classic_index.cpp:596-599)
            cobs_gpu::classic_construct(fq, "/tmp/not_an_index.txt", "", cobs_gpu::ClassicIndexParameters());
            return 4;
        } catch (const cobs_gpu::Error& err) {
            std::printf("refused %s\n", err.what());
        }
    } catch (const cobs_gpu::Error& err) {
        std::printf("EXCEPTION %s\n", err.what());
        return 1;
    }
    return 0;
}
