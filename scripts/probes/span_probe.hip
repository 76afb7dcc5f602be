// scripts/probes/span_probe.hip -- does the ADDRESS SPAN of a tile's column matter once the index is large?
// (round 6.)  Round 4 priced a column-major table on the 18.4 GB index: nothing (EXPERIMENTS.md).  There a tile column of the
// largest sub-index spans 4 M rows x 1664 B = 6.7 GB.  At 8 x the rows (147 GB resident, the `hbm_side` probe of the
// bench line: 0.73-0.76 of 8 TB/s where the 18.4 GB index reaches 0.83-0.84) it spans 53 GB -- 26 k pages of 2 MiB --
// while the lines it actually holds are 32 M x 128 B = 4 GB.  If address translation is what the large index pays for,
// the same gather from a column-major table (a tile's lines contiguous) is faster; if it is the HBM pins, it is not.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/span_probe.hip -o scripts/probes/span_probe
//   scripts/probes/span_probe [scale = 8] [queries = 10000]
// One wave = 8 queries x one tile (8 sixteen-byte chunks = one 128-byte line per gathered row), 1000 lines per query,
// 16 lines of scores written per query and tile -- the headline batch's pattern (scripts/probes/rw_mix_probe.hip).
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// layout 0: rows of `pitch` bytes (13 lines), a tile's lines `pitch` apart; 1: column-major (13 arrays of rows x 128 B);
// 2: row-major in BANDS of `band` rows, column-major inside a band (a tile's lines of one band contiguous: span / 13 with
//    a layout that still streams in from a row-major file band by band)
__global__ __launch_bounds__(64) void probe(const uint8_t* table, uint64_t base, uint64_t rows, uint32_t pitch, uint8_t* scores,
                                            uint32_t nqg, uint32_t trips, uint32_t wlines, int layout, uint64_t band, uint64_t salt) {
    const uint32_t lane = threadIdx.x, grp = lane >> 3, col = lane & 7u;
    const uint32_t tile = blockIdx.x / nqg, qg = blockIdx.x - tile * nqg;       // 13 tiles of ONE sub-index, tile-major
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 x[8];
    for (uint32_t t = 0; t < trips; t += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint64_t h = mix64(((uint64_t)blockIdx.x * 1024u + t + r) * 8u + grp + salt) % rows;
            uint64_t off;
            if (layout == 0) off = base + h * pitch + (uint64_t)tile * 128u;
            else if (layout == 1) off = base + ((uint64_t)tile * rows + h) * 128u;
            else {
                const uint64_t b = h / band, in = h - b * band, n = (b + 1) * band <= rows ? band : rows - b * band;
                off = base + b * band * pitch + ((uint64_t)tile * n + in) * 128u;
            }
            x[r] = t + r < trips ? *reinterpret_cast<const u32x4*>(table + off + col * 16u) : acc;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) acc ^= x[r];
    }
    for (uint32_t i = lane; i < 8u * wlines * 8u; i += 64) {
        const uint32_t q = i / (wlines * 8u), piece = i - q * wlines * 8u;
        uint8_t* dst = scores + ((uint64_t)qg * 8u + q) * 200704u + (uint64_t)tile * 2048u + piece * 16u;
        *reinterpret_cast<u32x4*>(dst) = acc;
    }
    if (wlines == 0 && acc.x == 0xdeadbeefu && acc.y == 0x12345u) scores[0] = 1;
}

int main(int argc, char** argv) {
    const double scale = argc > 1 ? atof(argv[1]) : 8.0;
    const uint32_t nq = argc > 2 ? (uint32_t)atoi(argv[2]) : 10000, nqg = nq / 8;
    const uint32_t pitch = 1664;
    uint64_t rows[8], base[8], table_bytes = 0;
    for (int p = 0; p < 8; ++p) {
        double r = 250000.0 * scale;
        for (int i = 0; i < p; ++i) r *= 1.4859942891369484;
        rows[p] = (uint64_t)(r + 0.5);
        base[p] = table_bytes;
        table_bytes += rows[p] * pitch;
    }
    uint8_t *table, *scores;
    if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&scores, (uint64_t)nq * 200704u) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    hipMemset(table, 0x5a, table_bytes);
    hipMemset(scores, 0, (uint64_t)nq * 200704u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("# table %.1f GB (C3 geometry x %g rows, rows of %u bytes), %u queries x 1000 lines + 16 written per tile, 13 tiles per sub-index; best of 3, ms\n",
           table_bytes / 1e9, scale, pitch, nq);
    printf("# sub-index        rows   column span row-major | row-major  column-major  bands of 64 Ki rows  bands of 8 Ki rows | gather only: row-major  column-major\n");
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (int p = 0; p < 8; ++p) {
        struct V { int layout; uint64_t band; uint32_t wl; } vs[6] = {{0, 1, 16}, {1, 1, 16}, {2, 65536, 16}, {2, 8192, 16}, {0, 1, 0}, {1, 1, 0}};
        float best[6];
        for (int v = 0; v < 6; ++v) {
            best[v] = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(13u * nqg), dim3(64), 0, 0, table, base[p], rows[p], pitch, scores, nqg, 1000u, vs[v].wl,
                                   vs[v].layout, vs[v].band, (uint64_t)(rep * 6 + v) * 7919u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best[v]) best[v] = ms;
            }
            sum[v] += best[v];
        }
        printf("  %d        %10llu   %6.1f GB              | %8.3f   %8.3f      %8.3f            %8.3f          |      %8.3f      %8.3f\n", p,
               (unsigned long long)rows[p], rows[p] * (double)pitch / 1e9, best[0], best[1], best[2], best[3], best[4], best[5]);
        fflush(stdout);
    }
    printf("  all                                          | %8.3f   %8.3f      %8.3f            %8.3f          |      %8.3f      %8.3f\n", sum[0], sum[1],
           sum[2], sum[3], sum[4], sum[5]);
    const double bytes = 8.0 * 13 * nqg * 8.0 * (1000 + 16) * 128.0;
    printf("# algorithmic rate of the row-major / column-major pass: %.0f / %.0f GB/s\n", bytes / sum[0] / 1e6, bytes / sum[1] / 1e6);
    return 0;
}
