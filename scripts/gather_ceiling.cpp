// scripts/gather_ceiling.cpp -- what the MI355X memory system delivers for the access pattern of
// the scan kernel with all arithmetic removed: every lane group of 8 lanes reads one random
// 128-byte line (16 bytes per lane) of a large buffer, 8 or 16 loads in flight per wave; and a
// plain streaming read of the same buffer for comparison.  Calibration only, not part of the
// library.   hipcc --offload-arch=gfx950 -O3 scripts/gather_ceiling.cpp -o /tmp/gather_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// lines: number of 128-byte lines of the buffer; trips: 8-line batches per lane group
template <int DEPTH, bool NT = false>
__global__ __launch_bounds__(256) void gather(const uint8_t* buf, uint64_t lines, uint32_t trips, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63u, grp = lane >> 3, col = lane & 7u;
    const uint64_t wid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t state = mix(wid * 8 + grp);
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t t = 0; t < trips; ++t) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t line = (state >> 20) % lines;
            const u32x4* p = reinterpret_cast<const u32x4*>(buf + line * 128 + col * 16);
            v[j] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void stream(const uint8_t* buf, uint64_t bytes, uint32_t* sink) {
    const uint64_t n = bytes / 16, stride = (uint64_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        acc ^= reinterpret_cast<const u32x4*>(buf)[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? std::atof(argv[1]) : 18.0;
    const uint64_t bytes = (uint64_t)(gb * 1e9) / 128 * 128, lines = bytes / 128;
    uint8_t* buf;
    uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t groups = 256 * 64;            // work-groups of 4 waves
    const uint32_t trips = 256;
    for (int depth : {8, 16}) {
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (depth == 8) hipLaunchKernelGGL(gather<8>, dim3(groups), dim3(256), 0, 0, buf, lines, trips, sink);
            else hipLaunchKernelGGL(gather<16>, dim3(groups), dim3(256), 0, 0, buf, lines, trips / 2, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double moved = (double)groups * 4 * 8 * trips * 8 * 128;   // waves x groups x trips x lines x bytes
            if (rep == 3) std::printf("random 128-byte lines over %.1f GB, %2d loads in flight per wave: %7.1f GB/s\n", gb, depth, moved / ms / 1e6);
        }
    }
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather<8, true>), dim3(groups), dim3(256), 0, 0, buf, lines, trips, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double moved = (double)groups * 4 * 8 * trips * 8 * 128;
        if (rep == 3) std::printf("random 128-byte lines over %.1f GB, non-temporal loads, 8 in flight:  %7.1f GB/s\n", gb, moved / ms / 1e6);
    }
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream, dim3(256 * 32), dim3(256), 0, 0, buf, bytes, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 3) std::printf("streaming read of %.1f GB: %7.1f GB/s\n", gb, bytes / ms / 1e6);
    }
    return 0;
}
