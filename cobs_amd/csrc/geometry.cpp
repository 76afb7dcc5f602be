// cobs_amd/csrc/geometry.cpp -- the launch geometry of the scan kernel (tile width, waves per work-group, multi-query
// groups) as a function of the chunk and the batch.  Kept apart from plan.cpp (shards, chunks, budgets) because this
// function, together with kernels.hip, is what the measured memory traffic of a workload depends on: bench.py stamps
// its lines with a hash of exactly these two files and replays profiles/traffic.json only for the same hash.
#include <algorithm>
#include <cstdint>

#include "engine.hpp"

namespace cobs_amd {

// Geometry of a scan launch: tile width W (16-byte column chunks per tile: 64, 32, 16, 8 or 4)
// and waves per work-group NW (1, 2 or 4).  A query's 8-term blocks are split over
// NV = NW * (64 / W) "virtual waves" (lane groups).
// * Narrow tiles: with W < 64 one wave-load fetches 64/W different rows, and a tile of one
//   sub-index is signature_size x W*16 bytes.  All queries of a batch work on the same tile
//   before the grid moves on (tile-major order), so narrow tiles turn the repeated lookups of
//   a batch into Infinity-Cache hits, and W = 8 makes every row slice exactly one 128-byte
//   line.  Interleaved A/B on MI355X, 10k x 1000-k-mer queries: W = 8 vs 64: C3 -7 % scan
//   time, 512-byte pages -9 %, 128-byte pages -12 %, 30 M-row sub-indexes that cannot be
//   cached -3 %; W = 4 (64-byte slices) halves throughput.
// * Every virtual wave should keep about two to four blocks (merging and expansion cost per
//   tile is fixed): NV = largest power of two <= blocks / 1.5, at most 32.  Measured (round 2,
//   after the in-register 8-bit epilogue) for 125/150/175/200/250-bp reads (12/15/18/21/28
//   blocks): (NW 2, W 16) -- 10 % faster than (2, 32) --, (2, 16), (1, 8), (1, 8), (2, 8).
// * Indexes narrower than a wave get the smallest tile that covers them (no idle lanes).
// Tuning hooks (per handle): tile_w, waves, mq force a value.



ScanGeom scan_geometry(const Chunk& c, uint64_t mean_blocks, uint64_t max_blocks, uint64_t num_hashes,
                       uint32_t forced_waves, int planes, bool idx64, const Tuning& tune) {
    uint32_t nv = 1;
    while (nv < 32 && (uint64_t)nv * 2 * 3 <= mean_blocks * 2) nv <<= 1;     // blocks / NV >= 1.5
    ScanGeom g;
    if (nv >= 16) { g.nwaves = (int)(nv / 8); g.tile_w = 8; }
    else if (nv == 8) {
        if (mean_blocks >= 17) { g.nwaves = 1; g.tile_w = 8; }
        else { g.nwaves = 2; g.tile_w = 16; }
    }
    else if (nv == 4) { g.nwaves = 2; g.tile_w = 32; }
    else if (nv == 2) { g.nwaves = 2; g.tile_w = 64; }
    else { g.nwaves = 1; g.tile_w = 64; }
    if (num_hashes > 1 && g.tile_w < 16) {     // generic-H kernel: 16 measured best
        g.tile_w = 16;
        g.nwaves = std::min(4, g.nwaves * 2);
    }
    // The generic-H row loop works in half blocks since round 6 (kernels.hip): its one- and two-wave instantiations fit the
    // 128 VGPRs of four waves per SIMD (123-125 at 10 planes; 133-149 above: three), the four-wave one does not without
    // spilling into its row loop (45 ms against 22.3 for C3 with three hash functions).  Two waves per group cost a long
    // query nothing: 22.32 ms with two, 22.34 with one (profiles/r06_generic_h_ab.txt).
    if (num_hashes > 1 && !idx64) g.nwaves = std::min(g.nwaves, 2);
    if (g.tile_w < 16) {
        // when even the largest sub-index fits the Infinity Cache with 256-byte slices, 16-chunk
        // tiles win (half the merge/expand work; C2: 7.3 vs 6.8 TB/s); otherwise 128-byte slices
        uint64_t max_sig = 0;
        for (const PageDev& pd : c.pages) max_sig = std::max<uint64_t>(max_sig, pd.sig);
        if (max_sig * 256ull <= (256ull << 20) && g.nwaves >= 2) { g.tile_w = 16; }
    }
    if (forced_waves) g.nwaves = (int)forced_waves;
    if (tune.waves) g.nwaves = (int)tune.waves;
    if (c.total_chunks < g.tile_w) {           // index narrower than the tile
        uint32_t cover = 4;
        while (cover < c.total_chunks) cover <<= 1;
        g.tile_w = std::min<uint32_t>(g.tile_w, std::max<uint32_t>(cover, 8));
        if (c.total_chunks <= 4) g.tile_w = 4;
    }
    if (tune.tile_w) g.tile_w = tune.tile_w;
    // Very short queries (<= 10 blocks: reads up to ~110 bp): the lane groups of a wave serve 8
    // different queries instead of splitting one query's few blocks.  Interleaved A/B on the C3
    // index (with the in-register 8-bit epilogue, which only the first wave runs): 50-bp reads
    // 2.27 ms with one wave per group vs 2.38 with two, 75 bp equal, 100 bp 6.04 ms with two vs
    // 6.11 with one; from 125 bp on the one-query geometry above is faster.
    g.multi_query = false;
    if (tune.mq != 0 && !idx64 && forced_waves == 0 && mean_blocks <= 10 && max_blocks <= 20 && c.total_chunks >= 8 &&
        scan_has_multi_query(planes, (uint32_t)num_hashes, 8)) {
        g.multi_query = true;
        g.tile_w = 8;
        g.nwaves = mean_blocks >= 8 ? 2 : 1;
        if (tune.waves) g.nwaves = (int)tune.waves;
        if (tune.tile_w && tune.tile_w < 64) g.tile_w = tune.tile_w;
    }
    if (tune.mq == 1 && !idx64) g.multi_query = true;
    if (g.multi_query && !scan_has_multi_query(planes, (uint32_t)num_hashes, g.tile_w)) g.multi_query = false;
    return g;
}

}  // namespace cobs_amd
