// scripts/probes/nt_store_probe.cpp -- what the host gives the expansion leg of the default call: T threads write a
// 307 MB array (256 queries x 100 000 results x 12 bytes) with 16-byte non-temporal stores, nothing else.
//   g++ -O3 -pthread -o /tmp/nt_store_probe scripts/probes/nt_store_probe.cpp && /tmp/nt_store_probe 32 64 16
#include <emmintrin.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

int main(int argc, char** argv) {
    const size_t bytes = 256ull * 100000 * 12;
    void* m = mmap(nullptr, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(m, bytes, MADV_HUGEPAGE);
    char* p = static_cast<char*>(m);
    for (size_t i = 0; i < bytes; i += 4096) p[i] = 1;
    for (int a = 1; a < argc; ++a) {
        const int T = std::atoi(argv[a]);
        double best = 1e9;
        for (int rep = 0; rep < 7; ++rep) {
            std::atomic<size_t> next{0};
            const size_t job = 600000;                     // bytes per job (50 000 results)
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            auto work = [&]() {
                for (;;) {
                    const size_t o = next.fetch_add(job);
                    if (o >= bytes) break;
                    const size_t e = o + job < bytes ? o + job : bytes;
                    const __m128i v = _mm_set1_epi32((int)o);
                    for (size_t b = o; b + 16 <= e; b += 16) _mm_stream_si128(reinterpret_cast<__m128i*>(p + b), v);
                    _mm_sfence();
                }
            };
            for (int t = 1; t < T; ++t) th.emplace_back(work);
            work();
            for (auto& x : th) x.join();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s < best) best = s;
        }
        std::printf("%3d threads: %.3f ms  %.1f GB/s (incl. starting the threads)\n", T, best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
