// tests/asan/index_main.cpp -- harness of scripts/asan_index.sh: parse_index_header (index_file.cpp)
// over damaged copies of the golden index files, built with -fsanitize=address,undefined.  The
// header bytes are copied to an exactly-sized heap block so that a read past the end is caught.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "index_file.hpp"

using namespace cobs_amd;

int main(int argc, char** argv) {
    size_t ok = 0, bad = 0;
    const unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    std::mt19937 rng(seed);
    for (int i = 2; i < argc; ++i) {
        FILE* f = std::fopen(argv[i], "rb");
        if (!f) continue;
        std::vector<uint8_t> raw;
        uint8_t buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n);
        std::fclose(f);
        for (int trial = 0; trial < 4000; ++trial) {
            std::vector<uint8_t> b = raw;
            switch (trial % 4) {
            case 0: b.resize(rng() % (b.size() + 1)); break;
            case 1: for (int k = 0, m = 1 + rng() % 8; k < m && !b.empty(); ++k) b[rng() % std::min<size_t>(b.size(), 256)] = (uint8_t)rng(); break;
            case 2: if (b.size() > 70) { const size_t at = 18 + rng() % 48; const uint32_t v = trial % 8 == 2 ? 0xFFFFFFFFu : (uint32_t)rng(); std::memcpy(&b[at], &v, 4); } break;
            default: if (b.size() > 70) { const size_t at = 18 + rng() % 40; const uint64_t v = (uint64_t)rng() << (rng() % 33); std::memcpy(&b[at], &v, 8); } break;
            }
            uint8_t* exact = (uint8_t*)std::malloc(b.size() ? b.size() : 1);
            std::memcpy(exact, b.data(), b.size());
            IndexMeta meta;
            std::string err;
            if (parse_index_header(exact, b.size(), meta, err)) {
                // what the engine derives from an accepted header must stay inside the file
                if (meta.data_offset > b.size()) { std::printf("data_offset outside the file\n"); return 2; }
                (void)meta.row_size(); (void)meta.counts_size();
                ++ok;
            } else {
                ++bad;
            }
            std::free(exact);
        }
    }
    std::printf("accepted %zu rejected %zu\n", ok, bad);
    return 0;
}
