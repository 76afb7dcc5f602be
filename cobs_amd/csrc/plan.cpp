// cobs_amd/csrc/plan.cpp -- host arithmetic of an opened index: which slices of which sub-indexes a shard holds
// (held_slices), how they are cut into chunks (resident: one per slice width; streamed: at most a stream buffer
// each), their layout in HBM (row pitch, zero rows, PageDev), what stays resident under an HBM budget
// (plan_index), and the geometry of a scan launch (scan_geometry).  No device work happens here.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

using namespace cobs_amd;

namespace cobs_amd {

// ---------------------------------------------------------------------------
// index staging

cobs_gpu_status select_device(const cobs_gpu_options* o, int* device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(COBS_GPU_ERR_NO_DEVICE,
                    "no HIP device visible; libcobs_gpu has no CPU fallback");
    }
    int dev = 0;
    if (o && o->device >= 0) {
        dev = o->device;
        if (dev >= n) return fail(COBS_GPU_ERR_ARG, "device ordinal out of range");
        HIP_TRY(hipSetDevice(dev));
    } else {
        HIP_TRY(hipGetDevice(&dev));
    }
    *device = dev;
    return COBS_GPU_OK;
}

// Rows are made of 16-byte chunks.  Starting every row on a 128-byte cache-line
// boundary removes the partial lines at both ends of a gathered row (measured on
// MI355X: 1568-byte rows, 1664-byte pitch: -8.5 % scan time); it is applied when
// it costs at most 12.5 % more HBM.  Tuning::row_align overrides.
uint32_t pitch_for(uint64_t ncols, const Tuning& tune) {
    uint64_t align = 16;
    for (uint64_t a : {128ull, 64ull, 32ull}) {
        if (round_up(ncols, a) * 8 <= ncols * 9) { align = a; break; }
    }
    if (tune.row_align) align = tune.row_align;
    return (uint32_t)round_up(ncols, align);
}

uint64_t slice_bytes(uint64_t sig, uint64_t ncols, const Tuning& tune) {
    return round_up((sig + 1) * (uint64_t)pitch_for(ncols, tune), 256);     // +1: the all-zero row
}

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

// fill pages / geometry of a chunk whose slices (equal ncols) are already listed
void layout_chunk(const Part& pt, Chunk& c, const Tuning& tune) {
    const IndexMeta& m = pt.meta;
    const uint64_t prb = m.page_row_bytes();
    const uint64_t ncols = c.vp.empty() ? 0 : c.vp[0].ncols;
    c.pitch = pitch_for(ncols, tune);
    c.cpp = c.pitch / 16;
    c.total_chunks = (uint32_t)c.vp.size() * c.cpp;
    c.pages.resize(c.vp.size());
    uint64_t off = 0, packed = 0;
    for (size_t i = 0; i < c.vp.size(); ++i) {
        const VPage& v = c.vp[i];
        PageDev& pd = c.pages[i];
        const uint64_t file_slot = ((m.kind == IndexKind::Compact ? (uint64_t)v.fp * prb : 0) + v.col0) * 8;
        pd.base = off;
        pd.sig = v.nrows ? v.nrows : m.signature_sizes[v.fp];      // rows in the buffer (its zero row sits at index sig)
        pd.row0 = v.nrows ? v.row0 : 0;
        pd.magic = ~0ull / pd.sig;
        pd.slot0 = (uint32_t)(file_slot - pt.slot_begin);
        pd.doc0 = (uint32_t)file_slot;
        pd.valid_bytes = (uint32_t)v.ncols;
        pd.tpage = v.fp - pt.first_page;
        off += round_up((pd.sig + 1) * (uint64_t)c.pitch, 256);
        packed += pd.sig * v.ncols;
    }
    c.bytes = off;
    c.stage_bytes = packed;
}

cobs_gpu_status check_meta(const IndexMeta& m) {
    if (m.term_size == 0) return fail(COBS_GPU_ERR_FORMAT, "term_size is zero");
    if (m.num_hashes == 0 || m.num_hashes > 64)
        return fail(COBS_GPU_ERR_UNSUPPORTED, "num_hashes must be in 1..64");
    if (m.canonicalize > 1)
        return fail(COBS_GPU_ERR_FORMAT, "Unknown canonicalize value " + std::to_string(m.canonicalize));
    for (uint64_t s : m.signature_sizes)
        if (s == 0 || s > (1ull << 46))
            return fail(COBS_GPU_ERR_UNSUPPORTED, "signature_size must be in 1..2^46");
    const uint64_t prb = m.page_row_bytes();
    if (prb == 0 || prb > (1ull << 28)) return fail(COBS_GPU_ERR_UNSUPPORTED, "row too wide");
    if (m.signature_sizes.empty()) return fail(COBS_GPU_ERR_FORMAT, "index holds no sub-index");
    // every byte count derived from the geometry stays far below 2^64 (a procedural index has no
    // file length to bound it): one sub-index at most 2^47 bytes, the file at most 2^50
    uint64_t total = 0;
    for (uint64_t s : m.signature_sizes) {
        uint64_t bytes = 0;
        if (__builtin_mul_overflow(s + 1, round_up(prb, 128), &bytes) || bytes > (1ull << 47) ||
            __builtin_add_overflow(total, bytes, &total) || total > (1ull << 50))
            return fail(COBS_GPU_ERR_UNSUPPORTED, "index geometry too large (a sub-index beyond 128 TiB or a file beyond 1 PiB)");
    }
    if ((uint64_t)m.num_pages() * prb > 0xFFFFFFF0ull / 8)
        return fail(COBS_GPU_ERR_UNSUPPORTED, "more than 2^32 score slots in one file");
    return COBS_GPU_OK;
}



// The slices of a file that shard `rank` of `count` holds (documents of different
// sub-indexes / row-byte columns never combine, so any cut of the (sub-index, column) space
// gives independent shards; reference compact_index/mmap_search_file.cpp:22-27,
// search_file.cpp:30-32).  The unit is one 16-byte column chunk of one sub-index.
//   mode 0 (default): equal WORK per shard.  The scan is a gather: a query term looks up ONE row in every sub-index,
//     so what a shard does per term is the row BYTES it holds (its columns), whatever the number of rows behind them.
//     A column chunk costs 1, or 1.1 in a sub-index whose 128-byte tile column (rows x 128 B) exceeds half the
//     256 MiB Infinity Cache (measured, scripts/shard_times.py: the 8 sub-indexes of C3 one per GPU scan a 10k-query
//     batch in 2.17 / 2.21 / 2.25 / 2.28 | 2.53 / 2.43 / 2.41 / 2.41 ms).  A cut may fall inside a sub-index, on a
//     multiple of 8 chunks (whole 128-byte lines); one within 3 % of a shard's share of a sub-index boundary snaps to it.
//     (Rounds 1-3 balanced the shards' BYTES in HBM, i.e. rows x columns: the shard with the small sub-indexes held
//     3.4x the columns of the one with the largest and took 7.6 ms against 0.95 ms -- a 2.5x speed-up on 8 GPUs.)
//   mode 1: whole sub-indexes, equal COUNT per shard (compact), 16-byte columns (classic).
//   mode 2: equal BYTES in HBM per shard (rows x columns) -- for an index that only fits when its footprint is spread
//     evenly; the scan times are then as uneven as the sub-indexes' signature sizes.
// The held slices are contiguous in score-slot order: [tail columns of the first sub-index]
// [whole sub-indexes] [head columns of the last].
std::vector<VPage> held_slices(const IndexMeta& m, uint32_t rank, uint32_t count, uint32_t mode) {
    const uint64_t prb = m.page_row_bytes();
    const uint32_t P = m.num_pages();
    const uint64_t nch = (prb + 15) / 16;                       // 16-byte chunks per row
    std::vector<VPage> out;
    if (count <= 1) {
        for (uint32_t p = 0; p < P; ++p) out.push_back(VPage{p, 0, prb});
        return out;
    }
    // cost of one column chunk of sub-index p
    auto weight = [&](uint32_t p) -> long double {
        if (mode == 2) return (long double)m.signature_sizes[p];
        return m.signature_sizes[p] * 128ull > (128ull << 20) ? 1.1L : 1.0L;
    };
    // a cut is a global chunk position in [0, P * nch]
    auto cut_of = [&](uint32_t r) -> uint64_t {
        if (r == 0) return 0;
        if (r >= count) return (uint64_t)P * nch;
        if (mode == 1) {
            if (m.kind == IndexKind::Compact) return (uint64_t)((uint64_t)P * r / count) * nch;
            return nch * r / count;
        }
        long double total = 0;
        for (uint32_t p = 0; p < P; ++p) total += weight(p) * nch;
        const long double share = total / count, ideal = share * r;
        long double acc = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const long double w = weight(p) * nch;
            if (acc + w < ideal) { acc += w; continue; }
            // the cut falls into sub-index p
            const long double tol = 0.03L * share;
            if (ideal - acc <= tol) return (uint64_t)p * nch;
            if (acc + w - ideal <= tol) return (uint64_t)(p + 1) * nch;
            uint64_t c = (uint64_t)((ideal - acc) / weight(p) + 0.5L);
            if (mode == 0 && nch >= 16) c = (c + 4) / 8 * 8;     // whole 128-byte lines on both sides of the cut
            if (c > nch) c = nch;
            return (uint64_t)p * nch + c;
        }
        return (uint64_t)P * nch;
    };
    const uint64_t c0 = cut_of(rank), c1 = std::max(cut_of(rank + 1), c0);
    for (uint64_t c = c0; c < c1;) {
        const uint32_t p = (uint32_t)(c / nch);
        const uint64_t in = c - (uint64_t)p * nch;
        const uint64_t end = std::min<uint64_t>(nch, in + (c1 - c));
        const uint64_t b0 = in * 16, b1 = std::min<uint64_t>(prb, end * 16);
        if (b1 > b0) out.push_back(VPage{p, b0, b1 - b0});
        c += end - in;
    }
    return out;
}



// Which slices this shard holds, their score-slot range and what they need in HBM.
cobs_gpu_status plan_part(Part& pt, const cobs_gpu_index* ix) {
    const IndexMeta& m = pt.meta;
    cobs_gpu_status st = check_meta(m);
    if (st != COBS_GPU_OK) return st;
    // row indices are 32-bit unless a sub-index (plus its zero row) does not fit them
    pt.idx64 = false;
    for (uint64_t s : m.signature_sizes)
        if (s >= 0xFFFFFFFFull) pt.idx64 = true;
    const uint64_t prb = m.page_row_bytes();
    pt.held = held_slices(m, ix->shard_rank, ix->shard_count, ix->shard_mode);
    pt.chunks.clear();
    pt.resident_bytes = 0;
    if (pt.held.empty()) {
        pt.first_page = pt.end_page = 0;
        pt.slot_begin = pt.slot_count = 0;
        return COBS_GPU_OK;
    }
    pt.first_page = pt.held.front().fp;
    pt.end_page = pt.held.back().fp + 1;
    const uint64_t page_slots = m.kind == IndexKind::Compact ? 8 * prb : 0;
    pt.slot_begin = (uint64_t)pt.held.front().fp * page_slots + pt.held.front().col0 * 8;
    pt.slot_count = 0;
    for (const VPage& v : pt.held) {
        pt.slot_count += v.ncols * 8;
        pt.resident_bytes += slice_bytes(m.signature_sizes[v.fp], v.ncols, ix->tune);
    }
    pt.tpages.assign(pt.end_page - pt.first_page, PageDev{});
    for (uint32_t fp = pt.first_page; fp < pt.end_page; ++fp) {
        PageDev& t = pt.tpages[fp - pt.first_page];
        t.sig = m.signature_sizes[fp];
        t.magic = ~0ull / t.sig;
        t.tpage = fp - pt.first_page;
    }
    return COBS_GPU_OK;
}

// Cut the held slices into chunks: resident (cap == 0) = one chunk per run of equal-width
// slices; streamed = chunks of at most `cap` bytes each (two device buffers of `cap` bytes).
cobs_gpu_status chunk_part(Part& pt, uint64_t cap, const Tuning& tune) {
    const IndexMeta& m = pt.meta;
    pt.chunks.clear();
    pt.streamed = cap != 0;
    pt.has_row_ranges = false;
    Chunk cur;
    uint64_t cur_bytes = 0;
    auto flush = [&]() {
        if (!cur.vp.empty()) {
            layout_chunk(pt, cur, tune);
            pt.chunks.push_back(std::move(cur));
            cur = Chunk();
            cur_bytes = 0;
        }
    };
    for (const VPage& v : pt.held) {
        const uint64_t sig = m.signature_sizes[v.fp];
        const uint64_t full = slice_bytes(sig, v.ncols, tune);
        if (!cur.vp.empty() && cur.vp[0].ncols != v.ncols) flush();
        if (cap == 0 || full <= cap) {
            if (cap != 0 && cur_bytes + full > cap) flush();
            cur.vp.push_back(v);
            cur_bytes += full;
            continue;
        }
        flush();
        // A single slice exceeds a buffer.  One hash function: cut it by ROWS -- every chunk holds whole rows of a range
        // [row0, row0 + n) and counts the terms whose row falls into it (the others read the chunk's zero row); counts
        // are sums over terms, so the ranges' partial scores add up (pass.cpp).  Whole rows cross PCIe at 56.2 GB/s,
        // column slices at 50-55 (below).  With several hash functions the H rows of a term are ANDed before they are
        // counted and may lie in different ranges: columns.  (Procedural indexes regenerate chunks: columns.)
        if (m.num_hashes == 1 && tune.row_ranges != 0 && !pt.synthetic) {
            const uint64_t pitch = pitch_for(v.ncols, tune);
            const uint64_t rows_fit = (cap / 256 * 256) / pitch;                 // a buffer is a multiple of 256 bytes
            const uint64_t fit = rows_fit > 1 ? rows_fit - 1 : 0;                // rows per buffer, beside the zero row
            if (fit >= tune.row_range_min) {
                const uint64_t nr = (sig + fit - 1) / fit;
                const uint64_t per = (sig + nr - 1) / nr;
                uint32_t no = 0;
                for (uint64_t r0 = 0; r0 < sig; r0 += per, ++no) {
                    VPage rv = v;
                    rv.row0 = r0;
                    rv.nrows = std::min<uint64_t>(per, sig - r0);
                    cur.vp.push_back(rv);
                    cur.row_range = true;
                    cur.range_no = no;
                    pt.has_row_ranges = true;
                    flush();
                }
                continue;
            }
        }
        // Cut it by columns (all rows, fewer documents) -- into the FEWEST slices that fit, of about equal width.  A slice
        // crosses PCIe as a 2-D copy of `width` bytes out of every row of the file, and what the link delivers falls with
        // the width (MI355X, 1568-byte rows, profiles/r04_h2d_probe.txt: 56.2 GB/s whole rows, 55.1 at 1024 bytes, 53.6 at
        // 640, 50.6 at 544, 43.0 at 288, 9.5 at 32): the widest slice a buffer holds plus a narrow remainder (round 3:
        // 1536 + 32, 1024 + 544, 640 + 640 + 288 for the three largest sub-indexes of C3 under a 6 GB budget) spent 4 % of
        // a pass on the remainders.
        uint64_t wmax = cap / (sig + 1);
        wmax = wmax >= 128 ? wmax / 128 * 128 : wmax / 16 * 16;
        while (wmax >= 16 && slice_bytes(sig, wmax, tune) > cap) wmax -= 16;
        if (wmax < 16)
            return fail(COBS_GPU_ERR_CAPACITY,
                        "hbm budget too small: a 16-byte column slice of the largest sub-index needs " +
                        std::to_string(2 * slice_bytes(sig, 16, tune)) + " bytes of streaming buffers");
        const uint64_t nsl = (v.ncols + wmax - 1) / wmax;
        uint64_t w = (v.ncols + nsl - 1) / nsl;                   // equal share ...
        const uint64_t al = w >= 256 ? 128 : 16;                  // ... rounded up to whole cache lines where that is cheap
        w = std::min(wmax, (w + al - 1) / al * al);
        for (uint64_t c0 = 0; c0 < v.ncols; c0 += w) {
            cur.vp.push_back(VPage{v.fp, v.col0 + c0, std::min<uint64_t>(w, v.ncols - c0)});
            flush();
        }
    }
    flush();
    return COBS_GPU_OK;
}

// Decide residency for all files of the handle together (the budget is one number for the
// whole handle): everything resident if it fits; otherwise the smallest files stay resident
// while they use at most half the budget and all other files are streamed through ONE pair of
// device buffers sized from what is left.
cobs_gpu_status plan_index(cobs_gpu_index* ix) {
    for (auto& pt : ix->parts) {
        cobs_gpu_status st = plan_part(pt, ix);
        if (st != COBS_GPU_OK) return st;
    }
    uint64_t total = 0;
    for (auto& pt : ix->parts) total += pt.resident_bytes;
    std::vector<bool> resident(ix->parts.size(), true);
    uint64_t cap = 0;
    if (ix->hbm_budget && total > ix->hbm_budget) {
        std::vector<size_t> order(ix->parts.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            return ix->parts[a].resident_bytes < ix->parts[b].resident_bytes;
        });
        uint64_t kept = 0;
        std::fill(resident.begin(), resident.end(), false);
        for (size_t i : order) {
            if (kept + ix->parts[i].resident_bytes > ix->hbm_budget / 2) break;
            kept += ix->parts[i].resident_bytes;
            resident[i] = true;
        }
        cap = (ix->hbm_budget - kept) / 2;
        if (cap == 0) return fail(COBS_GPU_ERR_CAPACITY, "hbm budget too small");
    }
    ix->stream.cap = 0;
    for (size_t i = 0; i < ix->parts.size(); ++i) {
        Part& pt = ix->parts[i];
        const bool res = resident[i] || pt.held.empty();
        cobs_gpu_status st = chunk_part(pt, res ? 0 : cap, ix->tune);
        if (st != COBS_GPU_OK) return st;
        pt.hbm_bytes = res ? pt.resident_bytes : 0;
        if (!res) ix->stream.cap = cap;
    }
    // the shared buffers are accounted to the first streamed file
    for (auto& pt : ix->parts)
        if (pt.streamed) { pt.hbm_bytes = 2 * cap; break; }
    uint64_t g = 0, l = 0;
    for (auto& p : ix->parts) {
        p.doc_offset = g;
        p.local_offset = l;
        g += p.meta.counts_size();
        l += p.slot_count;
    }
    ix->total_counts = g;
    ix->local_counts = l;
    return COBS_GPU_OK;
}


}  // namespace cobs_amd
