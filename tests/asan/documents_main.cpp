// tests/asan/documents_main.cpp -- harness of scripts/asan_documents.sh: every reader entry point of
// documents.cpp over the files named on the command line (built with -fsanitize=address,undefined).
#include "documents.hpp"
#include <cstdio>
#include <string>
static thread_local std::string g_err;
cobs_gpu_status cobs_gpu_set_error(cobs_gpu_status st, const char* msg) { g_err = msg; return st; }
extern "C" const char* cobs_gpu_last_error(void) { return g_err.c_str(); }
using namespace cobs_amd;
int main(int argc, char** argv) {
    size_t ok = 0, bad = 0, terms = 0;
    for (int i = 1; i < argc; ++i) {
        std::vector<DocEntry> l;
        if (load_entries(argv[i], l) != COBS_GPU_OK) { ++bad; continue; }
        for (const DocEntry& e : l)
            for (uint32_t k : {31u, 5u, 1u, 64u}) {
                std::string text((size_t)term_text_bound(e, k), '\0'), scratch;
                std::vector<TermSeg> segs;
                TermSink sink; sink.data = &text[0]; sink.cap = text.size();
                if (load_terms(e, k, sink, segs, scratch) == COBS_GPU_OK) {
                    for (auto& g : segs) { if (g.begin + g.len > sink.size) { std::printf("SEG OUT OF RANGE %s\n", argv[i]); return 2; } terms += g.len; }
                }
                (void)num_terms(e, k);
            }
        ++ok;
    }
    std::printf("ok %zu bad %zu terms %zu\n", ok, bad, terms);
    return 0;
}
