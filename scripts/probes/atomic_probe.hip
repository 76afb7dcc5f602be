// scripts/probes/atomic_probe.hip -- how fast are fire-and-forget 32-bit atomicOr's on gfx950 as a
// function of where they land?  Decides the layout of the construction kernel (build_kernel):
//   global    random words of one 512 MiB table                  (what a row-major matrix gets)
//   shared2m  random words of ONE 2 MiB table, all XCDs           (a document's filter, unmapped)
//   xcd2m     random words of a 2 MiB table PER XCD (blockIdx % 8) (a document's filter, XCD-local)
//   xcdNm     the same with N MiB per XCD
// hipcc --offload-arch=gfx950 -O3 scripts/probes/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// mode 0: one table of `words`; mode 1: table per XCD (xcd = blockIdx % 8), `words` each
__global__ __launch_bounds__(256) void probe(uint32_t* table, uint64_t words, int mode, uint64_t salt) {
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t h = mix64(gid ^ salt);
    uint32_t* base = mode == 1 ? table + (uint64_t)(blockIdx.x & 7u) * words : table;
    if (mode == 2) reinterpret_cast<uint8_t*>(table)[h % (words * 4)] = 1;            // scattered byte store
    else if (mode == 3) table[h % words] = 1u;                                             // scattered dword store
    else if (mode == 4) { if (table[h % words] == 0xdeadbeefu) table[0] = 1; }             // scattered dword load
    else if (mode == 5 || mode == 6) {                                                     // L2-local (workgroup scope) atomic
        uint32_t* b = mode == 6 ? table + (uint64_t)(blockIdx.x & 7u) * words : table;
        __hip_atomic_fetch_or(b + (h % words), 1u << (h >> 59), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else atomicOr(base + (h % words), 1u << (h >> 59));
}

int main() {
    const uint64_t total_words = (512ull << 20) / 4;
    uint32_t* t;
    hipMalloc(&t, total_words * 4);
    hipMemset(t, 0, total_words * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint64_t n = 256ull << 20;                       // atomics per launch
    struct Case { const char* name; uint64_t words; int mode; } cases[] = {
        {"global 512 MiB", total_words, 0},
        {"shared 2 MiB", (2ull << 20) / 4, 0},
        {"shared 32 MiB", (32ull << 20) / 4, 0},
        {"byte store 512 MiB", total_words, 2},
        {"byte store 2 MiB", (2ull << 20) / 4, 2},
        {"dword store 512 MiB", total_words, 3},
        {"dword load 512 MiB", total_words, 4},
        {"dword load 2 MiB", (2ull << 20) / 4, 4},
        {"wg-scope 512 MiB", total_words, 5},
        {"wg-scope 2 MiB", (2ull << 20) / 4, 5},
        {"wg-scope/XCD 2 MiB", (2ull << 20) / 4, 6},
        {"wg-scope/XCD 32 MiB", (32ull << 20) / 4, 6},
        {"per-XCD 1 MiB", (1ull << 20) / 4, 1},
        {"per-XCD 2 MiB", (2ull << 20) / 4, 1},
        {"per-XCD 3 MiB", (3ull << 20) / 4, 1},
        {"per-XCD 4 MiB", (4ull << 20) / 4, 1},
        {"per-XCD 8 MiB", (8ull << 20) / 4, 1},
        {"per-XCD 32 MiB", (32ull << 20) / 4, 1},
    };
    for (const Case& c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(probe, dim3((uint32_t)(n / 256)), dim3(256), 0, nullptr, t, c.words, c.mode, (uint64_t)rep * 977);
            hipEventRecord(e1, nullptr);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        std::printf("%-20s %8.3f ms  %7.1f G atomics/s\n", c.name, best, n / best / 1e6);
    }
    return 0;
}
