// scripts/probes/rw_mix_probe.hip -- what does this memory system deliver for the scan kernel's READ/WRITE MIX
// with all arithmetic removed?  The short-read shapes write a large share of their traffic (50-bp reads: one
// score byte per document and query against 20 gathered 128-byte lines per 1024 documents: 29 % writes); the
// gather-only ceiling (scripts/gather_ceiling.cpp) says nothing about them.
//
// One wave = one work-group of the multi-query scan: 8 lane groups (queries) x one tile of 8 sixteen-byte chunks
// (one 128-byte line per gathered row).  Per trip a wave-load fetches 8 random lines (16 B per lane); after
// `trips` trips the wave stores `wlines` 128-byte lines per query into that query's score row
// (rows 100 352 bytes apart, the tile's 1 KiB at the same offset of each: the kernel's store pattern).
//   rw_mix_probe [table_GB]     prints GB/s (reads + writes) for several (trips, wlines) mixes
// hipcc --offload-arch=gfx950 -O3 scripts/probes/rw_mix_probe.hip -o /tmp/rw_mix_probe && /tmp/rw_mix_probe
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct Pages { uint64_t base[8], rows[8]; };     // the C3 geometry: 8 sub-indexes of 250 k ... 4 M rows

// grid: tiles x query-groups (tile-major, as the scan kernel).  `local`: the lines of tile t come from the t % 13-th
// 128-byte column of sub-index t / 13 (pitch 1664 = 13 lines) -- the kernel's locality: all queries hit one column
// of one sub-index before the grid moves on; else from anywhere in the table.
__global__ __launch_bounds__(64) void probe(const uint8_t* table, uint64_t table_bytes, Pages pg, uint32_t pitch,
                                            uint8_t* scores, uint32_t nqg, uint32_t trips, uint32_t wlines, int local,
                                            uint64_t salt) {
    const uint32_t lane = threadIdx.x, grp = lane >> 3, col = lane & 7u;
    const uint32_t tile = blockIdx.x / nqg, qg = blockIdx.x - tile * nqg;
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 x[8];
    for (uint32_t t = 0; t < trips; t += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint64_t h = mix64(((uint64_t)blockIdx.x * 1024u + t + r) * 8u + grp + salt);
            const uint32_t p = (tile / 13u) & 7u;
            // local 2: the same sub-index stored COLUMN-major in 128-byte columns (a tile's lines are one contiguous
            // array of rows[p] lines instead of lines 1664 bytes apart)
            const uint64_t off = local == 2 ? pg.base[p] + ((uint64_t)(tile % 13u) * pg.rows[p] + h % pg.rows[p]) * 128u
                                 : local ? pg.base[p] + (h % pg.rows[p]) * pitch + (uint64_t)(tile % 13u) * 128u
                                         : (h % (table_bytes / 128u)) * 128u;
            // (the kernel pads a query's last block with the all-zero row, which stays cached: no traffic)
            x[r] = t + r < trips ? *reinterpret_cast<const u32x4*>(table + off + col * 16u) : acc;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) acc ^= x[r];
    }
    // the query group's score bytes of this tile: 8 queries x wlines lines, rows 100 352 bytes apart
    for (uint32_t i = lane; i < 8u * wlines * 8u; i += 64) {       // 16-byte pieces
        const uint32_t q = i / (wlines * 8u), piece = i - q * wlines * 8u;
        uint8_t* dst = scores + ((uint64_t)qg * 8u + q) * 100352u + (uint64_t)(tile % 98u) * 1024u + piece * 16u;
        *reinterpret_cast<u32x4*>(dst) = acc;
    }
    if (wlines == 0 && acc.x == 0xdeadbeefu && acc.y == 0x12345u) scores[0] = 1;    // keep the loads
}

// ROW RANGES SIZED FOR THE L2 (round 5): the tile's column of ONE small sub-index (S_p x 128 bytes:
// 32 ... 105 MB for the four smallest of C3 -- every row of it is looked up 10-40 times per 10k-query batch, from the
// Infinity Cache today, L2 hit rate 5 %) cut into R row ranges of <= 3 MB; the grid runs (tile, range)-major, a wave
// gathers only the lines of ITS range (1000 / R per query) and writes the tile's partial scores (16 lines per query,
// as the full scan does ONCE per tile).  Against the same sub-index scanned as today.  The arithmetic-free ceiling of
// the idea: what the memory system gives for the access pattern, before any kernel is built.
__global__ __launch_bounds__(64) void probe_ranges(const uint8_t* table, uint64_t base, uint64_t rows, uint32_t pitch,
                                                   uint8_t* scores, uint32_t nqg, uint32_t nranges, uint32_t trips,
                                                   uint32_t wlines, uint64_t salt) {
    const uint32_t lane = threadIdx.x, grp = lane >> 3, col = lane & 7u;
    const uint32_t qg = blockIdx.x % nqg, tr = blockIdx.x / nqg;       // queries fastest, then ranges, then tiles
    const uint32_t range = tr % nranges, tile = tr / nranges;
    const uint64_t per = (rows + nranges - 1) / nranges, r0 = (uint64_t)range * per;
    const uint64_t span = r0 + per <= rows ? per : rows - r0;
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 x[8];
    for (uint32_t t = 0; t < trips; t += 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint64_t h = mix64(((uint64_t)blockIdx.x * 1024u + t + r) * 8u + grp + salt);
            const uint64_t off = base + (r0 + h % span) * pitch + (uint64_t)tile * 128u;
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
    // optional: dynamic LDS bytes per work-group, to hold the occupancy at floor(160 KiB / lds) waves per CU
    // (the scan kernel runs 16 waves per CU at 125 VGPRs)
    const size_t lds = argc > 1 ? (size_t)atoi(argv[1]) : 0;
    const uint32_t pitch = 1664;
    Pages pg;
    uint64_t table_bytes = 0;
    for (int p = 0; p < 8; ++p) {
        double r = 250000.0;
        for (int i = 0; i < p; ++i) r *= 1.4859942891369484;       // 16^(1/7)
        pg.rows[p] = (uint64_t)(r + 0.5);
        pg.base[p] = table_bytes;
        table_bytes += pg.rows[p] * pitch;
    }
    const uint32_t nq = 40000, nqg = nq / 8, tiles = 104;
    uint8_t *table, *scores;
    if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&scores, (uint64_t)nq * 100352u) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    hipMemset(table, 0x5a, table_bytes);
    hipMemset(scores, 0, (uint64_t)nq * 100352u);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Mix { const char* name; uint32_t trips, wlines; int local; } mixes[] = {
        {"gather only, 20 lines per (query, tile), anywhere          ", 20, 0, 0},
        {"gather only, 20 lines per (query, tile), the tile's column ", 20, 0, 1},
        {"50-bp reads: 20 lines read, 8 lines (1 KiB) written, anywhere", 20, 8, 0},
        {"50-bp reads: 20 lines read, 8 lines (1 KiB) written, column  ", 20, 8, 1},
        {"50-bp reads: 20 read, 8 written, COLUMN-MAJOR table          ", 20, 8, 2},
        {"100-bp reads: 70 lines read, 8 lines written, column         ", 70, 8, 1},
        {"100-bp reads: 70 read, 8 written, COLUMN-MAJOR table         ", 70, 8, 2},
        {"150-bp reads: 120 lines read, 8 lines written, column        ", 120, 8, 1},
        {"1000-k-mer queries: 1000 lines read, 16 lines (u16) written  ", 1000, 16, 1},
        {"1000-k-mer queries: 1000 read, 16 written, COLUMN-MAJOR table", 1000, 16, 2},
        {"gather only, 1000 lines per (query, tile), the tile's column ", 1000, 0, 1},
        {"gather only, 1000 lines, COLUMN-MAJOR table                  ", 1000, 0, 2},
        {"stores only: 8 lines per (query, tile)                       ", 0, 8, 1},
    };
    if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("# table %.1f GB (rows of %u bytes), %u queries x %u tiles, one wave per 8 queries and tile; best of 3; %zu bytes of LDS per wave%s\n",
           table_bytes / 1e9, pitch, nq, tiles, lds, lds ? " (occupancy capped)" : "");
    for (const Mix& m : mixes) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(tiles * nqg), dim3(64), lds, 0, table, table_bytes, pg, pitch, scores, nqg,
                               m.trips, m.wlines, m.local, (uint64_t)rep * 7919u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double rd = (double)tiles * nqg * 8.0 * m.trips * 128.0;
        const double wr = (double)tiles * nqg * 8.0 * m.wlines * 128.0;
        printf("%s  %8.3f ms   read %7.2f GB  written %6.2f GB (%4.1f %%)   %7.1f GB/s\n", m.name, best, rd / 1e9, wr / 1e9,
               100.0 * wr / (rd + wr), (rd + wr) / best / 1e6);
    }
    // ---- row ranges sized for the L2: the four smallest sub-indexes, 10 000 queries (the headline batch) ----
    {
        const uint32_t nq2 = 10000, nqg2 = nq2 / 8;
        printf("# row ranges of <= 3 MB per tile column, (tile, range)-major, %u queries; per sub-index: today's pattern vs ranges\n", nq2);
        double sum_now = 0, sum_rng = 0;
        for (int p = 0; p < 4; ++p) {
            const uint32_t R = (uint32_t)((pg.rows[p] * 128ull + (3u << 20) - 1) / (3u << 20));
            const uint32_t trips_r = ((1000u + R - 1) / R + 7u) / 8u * 8u;       // a range's share of the 1000 terms, in blocks of 8
            float now = 1e30f, rng = 1e30f, rng_nw = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                float ms = 0;
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe_ranges, dim3(13u * 1u * nqg2), dim3(64), lds, 0, table, pg.base[p], pg.rows[p], pitch, scores,
                                   nqg2, 1u, 1000u, 16u, (uint64_t)rep * 7919u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < now) now = ms;
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe_ranges, dim3(13u * R * nqg2), dim3(64), lds, 0, table, pg.base[p], pg.rows[p], pitch, scores,
                                   nqg2, R, trips_r, 16u, (uint64_t)rep * 104729u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < rng) rng = ms;
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe_ranges, dim3(13u * R * nqg2), dim3(64), lds, 0, table, pg.base[p], pg.rows[p], pitch, scores,
                                   nqg2, R, trips_r, 0u, (uint64_t)rep * 15485863u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < rng_nw) rng_nw = ms;
            }
            printf("sub-index %d (%llu rows, column %.0f MB): today %7.3f ms | %2u ranges x %3u lines + 16 written each %7.3f ms | the same without the partial-score writes %7.3f ms\n",
                   p, (unsigned long long)pg.rows[p], pg.rows[p] * 128.0 / 1e6, now, R, trips_r, rng, rng_nw);
            sum_now += now;
            sum_rng += rng;
        }
        printf("four smallest sub-indexes: today %.3f ms, in L2-sized row ranges %.3f ms (%+.1f %%)\n", sum_now, sum_rng,
               100.0 * (sum_rng - sum_now) / sum_now);
    }
    return 0;
}
