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
    if (tune.packed_width != 0 && ncols == tune.packed_width) align = 16;       // a streamed chunk of whole file rows
    return (uint32_t)round_up(ncols, align);
}

uint64_t slice_bytes(uint64_t sig, uint64_t ncols, const Tuning& tune) {
    return round_up((sig + 1) * (uint64_t)pitch_for(ncols, tune), 256);     // +1: the all-zero row
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
//   mode 0 (default): equal scan TIME per shard.  The scan is a gather: a query term looks up ONE row in every sub-index,
//     so what a shard does per term follows the 128-byte LINES of a row it holds (its columns), whatever the number of
//     rows behind them -- and a line of a sub-index whose tile column (rows x 128 B) fits the 256 MiB Infinity Cache
//     beside its neighbours comes in faster than one that leaves the HBM pins every time.  Measured
//     (scripts/shard_times.py: the 8 sub-indexes of C3 one per GPU, 13 lines each, scan a 10k-query batch in 2.14 / 2.20 /
//     2.24 / 2.29 | 2.53 / 2.52 / 2.53 / 2.58 ms; tile columns 32 ... 105 | 156 ... 512 MB; least squares over that run
//     and the 8-way splits of rounds 5 and 6, 24 shards: 0.845 / 0.885 / 0.905 / 0.906 | 1.00 of a large sub-index's line,
//     residuals up to 3 % = what two boxes differ by): a line costs min(0.905, 0.765 + 0.0025 per MB of column) up to
//     128 MiB of column, 1 above.  The unit of a cut is a whole line
//     (8 chunks; rows below 256 bytes: a chunk) -- a line shared by two shards is fetched by both -- and a row's last,
//     partial line costs a whole one.  The cuts are the contiguous partition of the lines with the smallest maximum
//     (binary search over the bound + greedy packing), each cut then as close to its equal share as that bound allows;
//     a sub-index boundary within 3 % of a share is preferred.  [Until round 6: a chunk cost 1 or 1.1, cuts at the
//     share rounded to 8 chunks -- shards of 12 to 14 lines, C3 over 8 GPUs 2.31 ... 2.55 ms, balance 0.942.]
//     (Rounds 1-3 balanced the shards' BYTES in HBM, i.e. rows x columns: the shard with the small sub-indexes held
//     3.4x the columns of the one with the largest and took 7.6 ms against 0.95 ms -- a 2.5x speed-up on 8 GPUs.)
//   mode 1: whole sub-indexes, equal COUNT per shard (compact), 16-byte columns (classic).
//   mode 2: equal BYTES in HBM per shard (rows x columns) -- for an index that only fits when its footprint is spread
//     evenly; the scan times are then as uneven as the sub-indexes' signature sizes.
// The held slices are contiguous in score-slot order: [tail columns of the first sub-index]
// [whole sub-indexes] [head columns of the last].
namespace {

// mode 0: the cuts (global chunk positions, [count + 1]) of the min-max partition of the lines
std::vector<uint64_t> time_balanced_cuts(const IndexMeta& m, uint32_t count) {
    const uint64_t prb = m.page_row_bytes();
    const uint32_t P = m.num_pages();
    const uint64_t nch = (prb + 15) / 16;
    const uint64_t u = nch >= 16 ? 8 : 1;                       // chunks per tile: a 128-byte line, or a chunk of a narrow row
    const uint64_t tp = (nch + u - 1) / u;                      // tiles per sub-index
    const uint64_t n = (uint64_t)P * tp;
    std::vector<long double> w(P), before(P + 1, 0.0L);         // cost of a tile of p; of all tiles of the sub-indexes before p
    for (uint32_t p = 0; p < P; ++p) {
        const long double col_mb = (long double)m.signature_sizes[p] * 128.0L / 1e6L;
        const long double rel = m.signature_sizes[p] * 128ull > (128ull << 20) ? 1.0L : std::min(0.905L, 0.765L + 0.0025L * col_mb);
        w[p] = rel * (long double)u;
        before[p + 1] = before[p] + w[p] * (long double)tp;
    }
    const long double total = before[P];
    auto pre = [&](uint64_t t) -> long double {                 // cost of the tiles [0, t)
        const uint64_t p = t / tp;
        return p >= P ? total : before[p] + w[p] * (long double)(t - p * tp);
    };
    // the furthest tile position from `t` whose tiles [t, .) cost at most `budget`
    auto advance = [&](uint64_t t, long double budget) -> uint64_t {
        budget *= 1.0L + 1e-12L;
        while (t < n) {
            const uint64_t p = t / tp, left = (p + 1) * tp - t;
            const long double all = w[p] * (long double)left;
            if (all <= budget) { budget -= all; t += left; continue; }
            return t + (uint64_t)(budget / w[p]);
        }
        return n;
    };
    // the earliest tile position before `t` whose tiles [., t) cost at most `budget`
    auto retreat = [&](uint64_t t, long double budget) -> uint64_t {
        budget *= 1.0L + 1e-12L;
        while (t > 0) {
            const uint64_t p = (t - 1) / tp, have = t - p * tp;
            const long double all = w[p] * (long double)have;
            if (all <= budget) { budget -= all; t -= have; continue; }
            return t - (uint64_t)(budget / w[p]);
        }
        return 0;
    };
    auto fits = [&](long double bound) {
        uint64_t t = 0;
        for (uint32_t r = 0; r < count && t < n; ++r) t = advance(t, bound);
        return t >= n;
    };
    long double lo = 0.0L, hi = total;
    for (uint32_t p = 0; p < P; ++p) lo = std::max(lo, w[p]);
    if (!fits(lo)) {
        for (int it = 0; it < 80; ++it) {
            const long double mid = (lo + hi) / 2;
            if (fits(mid)) hi = mid; else lo = mid;
        }
    } else {
        hi = lo;
    }
    const long double bound = hi;
    // earliest start of shard r when the shards behind it are packed from the end
    std::vector<uint64_t> start_min(count + 1, n);
    for (uint32_t r = count; r-- > 0;) start_min[r] = retreat(start_min[r + 1], bound);
    const long double share = total / count, tol = 0.03L * share;
    std::vector<uint64_t> tiles(count + 1, 0);
    tiles[count] = n;
    for (uint32_t r = 1; r < count; ++r) {
        const long double ideal = share * r;
        uint64_t t = advance(0, ideal);                                          // pre(t) <= ideal < pre(t + 1)
        if (t < n && pre(t + 1) - ideal < ideal - pre(t)) ++t;
        // a sub-index boundary close enough to the share: whole sub-indexes on both sides
        const uint64_t b0 = t / tp * tp, b1 = std::min<uint64_t>(n, b0 + tp);
        if (ideal - pre(b0) <= tol) t = b0;
        else if (pre(b1) - ideal <= tol) t = b1;
        const uint64_t least = std::max(start_min[r], tiles[r - 1]), most = advance(tiles[r - 1], bound);
        tiles[r] = std::min(std::max(t, least), std::max(most, least));
    }
    std::vector<uint64_t> cuts(count + 1, 0);
    for (uint32_t r = 0; r <= count; ++r) {
        const uint64_t p = tiles[r] / tp, k = tiles[r] - p * tp;
        cuts[r] = p >= P ? (uint64_t)P * nch : p * nch + std::min(nch, k * u);
    }
    return cuts;
}

}  // namespace

std::vector<VPage> held_slices(const IndexMeta& m, uint32_t rank, uint32_t count, uint32_t mode) {
    const uint64_t prb = m.page_row_bytes();
    const uint32_t P = m.num_pages();
    const uint64_t nch = (prb + 15) / 16;                       // 16-byte chunks per row
    std::vector<VPage> out;
    if (count <= 1) {
        for (uint32_t p = 0; p < P; ++p) out.push_back(VPage{p, 0, prb});
        return out;
    }
    // a cut is a global chunk position in [0, P * nch]
    auto cut_of = [&](uint32_t r) -> uint64_t {
        if (r == 0) return 0;
        if (r >= count) return (uint64_t)P * nch;
        if (mode == 1) {
            if (m.kind == IndexKind::Compact) return (uint64_t)((uint64_t)P * r / count) * nch;
            return nch * r / count;
        }
        // mode 2: equal bytes in HBM (a chunk of sub-index p weighs its rows)
        long double total = 0;
        for (uint32_t p = 0; p < P; ++p) total += (long double)m.signature_sizes[p] * nch;
        const long double share = total / count, ideal = share * r;
        long double acc = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const long double w = (long double)m.signature_sizes[p] * nch;
            if (acc + w < ideal) { acc += w; continue; }
            // the cut falls into sub-index p
            const long double tol = 0.03L * share;
            if (ideal - acc <= tol) return (uint64_t)p * nch;
            if (acc + w - ideal <= tol) return (uint64_t)(p + 1) * nch;
            uint64_t c = (uint64_t)((ideal - acc) / (long double)m.signature_sizes[p] + 0.5L);
            if (c > nch) c = nch;
            return (uint64_t)p * nch + c;
        }
        return (uint64_t)P * nch;
    };
    uint64_t c0, c1;
    if (mode == 0 || mode > 2) {
        const std::vector<uint64_t> cuts = time_balanced_cuts(m, count);
        c0 = cuts[std::min(rank, count)];
        c1 = std::max(cuts[std::min(rank + 1, count)], c0);
    } else {
        c0 = cut_of(rank);
        c1 = std::max(cut_of(rank + 1), c0);
    }
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
// keep_bytes (streamed parts, round 5): HBM left over beside the two stream buffers.  Residency is decided per SLICE,
// not per file [VERDICT r4 item 4: a single file above the budget crossed PCIe whole in every pass]: whole slices
// that fit keep_bytes stay resident (the smallest first, see below) and are scanned where they lie; the rest is
// streamed as before.
// Chunks stay in the order of the held slices (document order: what a top-k pass without score rows relies on), a
// change of residency ends a chunk.  *kept = bytes of the resident chunks.
cobs_gpu_status chunk_part(Part& pt, uint64_t cap, const Tuning& tune_in, uint64_t keep_bytes, uint64_t* kept) {
    // A streamed chunk of WHOLE file rows keeps the file's row pitch where that is a multiple of 16 (no padding to
    // 128-byte lines): it crosses PCIe as ONE linear copy (57.6 GB/s on MI355X against 56.2 for the 2-D copy that
    // re-pitches 1568-byte rows to 1664, profiles/r04_h2d_probe.txt).  Its scan pays for rows that straddle cache lines
    // (+8.5 % scan time) -- 2 ms of a pass that waits 320 ms for its copies.  Column slices are 2-D copies either way
    // and keep the line-aligned pitch (packed, they measured 3-9 % slower: profiles/r04_shard_times.txt).
    Tuning tune = tune_in;
    if (cap != 0 && !pt.synthetic && tune.row_align == 0 && tune.stream_packed != 0 && pt.meta.page_row_bytes() % 16 == 0)
        tune.packed_width = (uint32_t)pt.meta.page_row_bytes();
    const IndexMeta& m = pt.meta;
    pt.chunks.clear();
    pt.streamed = cap != 0;
    pt.has_row_ranges = false;
    if (kept) *kept = 0;
    // Which slices stay resident: the SMALLEST first, then whatever else still fits (largest first).  What a streamed
    // slice costs a pass is min(its bytes, the rows the batch looks up in it x pitch) -- a small batch fetches rows, and
    // then every slice costs about the same whatever its size, so the number of resident slices counts; a large batch
    // copies chunks whole, and then only the resident bytes count, which the second sweep fills up.  (Round 5's first
    // version maximised resident bytes alone: under 64 GB it kept the ONE largest sub-index of the 184 GB file, 62.7 GB
    // of which a 10k-query batch looks up 15.7 GB, and streamed the five it reads whole.)
    std::vector<bool> keep(pt.held.size(), false);
    if (cap != 0 && keep_bytes != 0 && !pt.held.empty()) {
        std::vector<size_t> order(pt.held.size());
        std::iota(order.begin(), order.end(), 0);
        auto bytes_of = [&](size_t i) { return slice_bytes(m.signature_sizes[pt.held[i].fp], pt.held[i].ncols, tune); };
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return bytes_of(x) < bytes_of(y); });
        uint64_t left = keep_bytes;
        size_t k = 0;
        for (; k < order.size() && bytes_of(order[k]) <= left; ++k) {
            keep[order[k]] = true;
            left -= bytes_of(order[k]);
        }
        for (size_t j = order.size(); j-- > k;)
            if (bytes_of(order[j]) <= left) {
                keep[order[j]] = true;
                left -= bytes_of(order[j]);
            }
    }
    Chunk cur;
    uint64_t cur_bytes = 0;
    auto flush = [&]() {
        if (!cur.vp.empty()) {
            layout_chunk(pt, cur, tune);
            if (cur.resident && kept) *kept += cur.bytes;
            pt.chunks.push_back(std::move(cur));
            cur = Chunk();
            cur_bytes = 0;
        }
    };
    for (size_t hi = 0; hi < pt.held.size(); ++hi) {
        const VPage& v = pt.held[hi];
        const uint64_t sig = m.signature_sizes[v.fp];
        const uint64_t full = slice_bytes(sig, v.ncols, tune);
        if (!cur.vp.empty() && (cur.vp[0].ncols != v.ncols || cur.resident != keep[hi])) flush();
        if (keep[hi]) {             // stays in HBM: runs of equal width share a chunk, as in a resident part
            cur.resident = true;
            cur.vp.push_back(v);
            cur_bytes += full;
            continue;
        }
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
    // The stream buffers need not be larger than what keeps the link busy: beyond stream_buf_kib (256 MiB: a chunk
    // crosses PCIe in ~5 ms, its scan takes 2-3 ms) the rest of the budget holds slices of the streamed files RESIDENT.
    // (configs[4] on one GPU, 18.4 GB under 6 GB: buffers of 3 GB / 512 MiB / 256 MiB -> 323 / 239 / 229 ms per pass.)
    uint64_t spare = 0;
    // (Small buffers are for chunks of whole ROWS.  A file that has to be cut by COLUMNS -- several hash functions, a
    // procedural index, row ranges switched off -- crosses PCIe as 2-D copies whose rate falls with the slice width
    // (profiles/r04_h2d_probe.txt: 56 GB/s whole rows, 43 at 288 bytes, 9.5 at 32): it keeps the wide buffers.)
    bool by_columns = false;
    for (size_t i = 0; i < ix->parts.size(); ++i)
        if (!resident[i] && !ix->parts[i].held.empty() &&
            (ix->parts[i].meta.num_hashes != 1 || ix->tune.row_ranges == 0 || ix->parts[i].synthetic))
            by_columns = true;
    if (cap && !by_columns) {
        const uint64_t want = (uint64_t)ix->tune.stream_buf_kib << 10;
        if (want && cap > want) {
            spare = 2 * (cap - want);
            cap = want;
        }
    }
    ix->stream.cap = 0;
    ix->stream.resident_bytes = ix->stream.pass_bytes = 0;
    for (size_t i = 0; i < ix->parts.size(); ++i) {
        Part& pt = ix->parts[i];
        const bool res = resident[i] || pt.held.empty();
        uint64_t kept_here = 0;
        cobs_gpu_status st = chunk_part(pt, res ? 0 : cap, ix->tune, res ? 0 : spare, &kept_here);
        if (st != COBS_GPU_OK) return st;
        pt.hbm_bytes = res ? pt.resident_bytes : kept_here;
        if (!res) {
            ix->stream.cap = cap;
            spare -= std::min(spare, kept_here);
            ix->stream.resident_bytes += kept_here;
            for (const Chunk& c : pt.chunks)
                if (!c.resident) ix->stream.pass_bytes += c.stage_bytes;
        }
    }
    // the shared buffers are accounted to the first streamed file
    for (auto& pt : ix->parts)
        if (pt.streamed) { pt.hbm_bytes += 2 * cap; break; }
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
