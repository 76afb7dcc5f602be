// cobs_amd/csrc/engine.cpp -- host side of libcobs_gpu.so: index staging into
// HBM, batch workspaces, kernel orchestration, result ranking, and the C ABI of
// include/cobs_gpu.h.  There is no CPU fallback: without a HIP device every entry
// point that needs one fails with COBS_GPU_ERR_NO_DEVICE.
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

// ---------------------------------------------------------------------------
// errors

namespace {
thread_local std::string g_last_error;
}

namespace cobs_amd {

cobs_gpu_status fail(cobs_gpu_status st, const std::string& msg) {
    g_last_error = msg;
    return st;
}

cobs_gpu_status hip_fail(hipError_t e, const char* what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    const bool nodev = e == hipErrorNoDevice || e == hipErrorInvalidDevice ||
                       e == hipErrorInsufficientDriver;
    (void)hipGetLastError();
    return fail(nodev ? COBS_GPU_ERR_NO_DEVICE : COBS_GPU_ERR_HIP, m);
}

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

Tuning Tuning::from_env() {
    Tuning t;
    if (const char* e = getenv("COBS_GPU_ROW_ALIGN")) {
        const uint64_t v = std::strtoull(e, nullptr, 10);
        if (v >= 16 && v <= 4096 && v % 16 == 0) t.row_align = (uint32_t)v;
    }
    if (const char* e = getenv("COBS_GPU_WAVES")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) t.waves = (uint32_t)v;
    }
    if (const char* e = getenv("COBS_GPU_TILE_W")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) t.tile_w = (uint32_t)v;
    }
    if (const char* e = getenv("COBS_GPU_MQ")) t.mq = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_PASS_BYTES")) t.pass_bytes = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    if (const char* e = getenv("COBS_GPU_PIPE_CHARS")) t.pipe_chars = std::strtoull(e, nullptr, 10);
    if (getenv("COBS_GPU_NO_PIN")) t.no_pin = true;
    if (const char* e = getenv("COBS_GPU_GRAPH")) t.graph = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_LDS_STAGED")) t.lds_staged = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_DEVICE_RANK")) t.device_rank = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_TRACE")) t.trace = atoi(e) != 0;
    return t;
}

Part::~Part() {
    for (Chunk& c : chunks) {
        if (c.d_data) (void)hipFree(c.d_data);
        if (c.d_pages) (void)hipFree(c.d_pages);
        if (c.d_src) (void)hipFree(c.d_src);
        for (auto* p2 : c.d_pages2) if (p2) (void)hipFree(p2);
    }
    if (d_tpages) (void)hipFree(d_tpages);
    if (file_pinned && file) (void)hipHostUnregister(const_cast<uint8_t*>(file->data()));
}

StreamBufs::~StreamBufs() {
    for (auto& e : copied) if (e) (void)hipEventDestroy(e);
    for (auto& e : scanned) if (e) (void)hipEventDestroy(e);
    if (hashed) (void)hipEventDestroy(hashed);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
}

}  // namespace cobs_amd

cobs_gpu_batch::~cobs_gpu_batch() {
    if (xchg) destroy_exchange(xchg);
    if (rank) destroy_rank_work(rank);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    for (auto& e : graph_more) if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (graph_stream) (void)hipStreamDestroy(graph_stream);
    for (auto& r : ev) for (auto& e : r) if (e) (void)hipEventDestroy(e);
    if (run_done) (void)hipEventDestroy(run_done);
    if (done) (void)hipEventDestroy(done);
    if (own_stream) (void)hipStreamDestroy(own_stream);
}

cobs_gpu_index::~cobs_gpu_index() { for (auto* b : scratch) delete b; }

namespace {

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
struct ScanGeom { uint32_t tile_w; int nwaves; bool multi_query; };

// a replayed small pass brings this many hit-pool entries home inside the graph
constexpr size_t kGraphPoolPrefix = 2048;

// K3 orders up to this many survivors per (query, file) on the device (8-byte keys in 64 KB of LDS)
constexpr size_t kTopkSortLimit = 8192;
// largest k for which K2 selects per tile (a tile holds 512 or more documents; the pool is queries x tiles x k entries)
constexpr size_t kTileTopkMax = 128;

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
        pd.sig = m.signature_sizes[v.fp];
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

}  // namespace

namespace cobs_amd {

// The slices of a file that shard `rank` of `count` holds (documents of different
// sub-indexes / row-byte columns never combine, so any cut of the (sub-index, column) space
// gives independent shards; reference compact_index/mmap_search_file.cpp:22-27,
// search_file.cpp:30-32).  The unit is one 16-byte column chunk of one sub-index; its cost is
// the sub-index's signature size (rows).
//   mode 0 (default): equal BYTES per shard -- a cut may fall inside a sub-index (8 sub-indexes
//     whose sizes differ 16x would otherwise give 8 GPUs a 3x speed-up at best); a cut within
//     3 % of a shard's share of a sub-index boundary snaps to it.
//   mode 1: whole sub-indexes, equal COUNT per shard (compact), 16-byte columns (classic).
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
    // a cut is a global chunk position in [0, P * nch]
    auto cut_of = [&](uint32_t r) -> uint64_t {
        if (r == 0) return 0;
        if (r >= count) return (uint64_t)P * nch;
        if (mode == 1) {
            if (m.kind == IndexKind::Compact) return (uint64_t)((uint64_t)P * r / count) * nch;
            return nch * r / count;
        }
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

}  // namespace cobs_amd

namespace {

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
        // a single slice exceeds a buffer: cut it by columns (all rows, fewer documents)
        uint64_t w = cap / (sig + 1);
        w = w >= 128 ? w / 128 * 128 : w / 16 * 16;
        while (w >= 16 && slice_bytes(sig, w, tune) > cap) w -= 16;
        if (w < 16)
            return fail(COBS_GPU_ERR_CAPACITY,
                        "hbm budget too small: a 16-byte column slice of the largest sub-index needs " +
                        std::to_string(2 * slice_bytes(sig, 16, tune)) + " bytes of streaming buffers");
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

cobs_gpu_status alloc_part(cobs_gpu_index* ix, Part& pt) {
    for (Chunk& c : pt.chunks) {
        HIP_TRY(hipMalloc((void**)&c.d_pages, sizeof(PageDev) * c.pages.size()));
        HIP_TRY(hipMemcpy(c.d_pages, c.pages.data(), sizeof(PageDev) * c.pages.size(), hipMemcpyHostToDevice));
    }
    if (pt.chunks.empty()) return COBS_GPU_OK;
    HIP_TRY(hipMalloc((void**)&pt.d_tpages, sizeof(PageDev) * pt.tpages.size()));
    HIP_TRY(hipMemcpy(pt.d_tpages, pt.tpages.data(), sizeof(PageDev) * pt.tpages.size(), hipMemcpyHostToDevice));
    if (!pt.streamed) {
        for (Chunk& c : pt.chunks) HIP_TRY(hipMalloc((void**)&c.d_data, c.bytes));
        return COBS_GPU_OK;
    }
    StreamBufs& sb = ix->stream;
    size_t dev = 0, host = 0;
    for (const Chunk& c : pt.chunks) { dev = std::max(dev, c.bytes); host = std::max(host, c.stage_bytes); }
    sb.stage_need = std::max(sb.stage_need, host);
    for (int i = 0; i < 2; ++i) {
        if (sb.sbuf[i].cap < dev) {
            // grow keeping nothing: buffers are only (re)allocated while the index is opened
            HIP_TRY(sb.sbuf[i].reserve(dev));
        }
        if (!sb.copied[i]) HIP_TRY(hipEventCreateWithFlags(&sb.copied[i], hipEventDisableTiming));
        if (!sb.scanned[i]) HIP_TRY(hipEventCreateWithFlags(&sb.scanned[i], hipEventDisableTiming));
    }
    if (!sb.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&sb.copy_stream, hipStreamNonBlocking));
    return COBS_GPU_OK;
}

// Resident chunks: copy the held columns of every held sub-index from the mapped file into HBM.
// The index file (an mmap of the page cache) -> HBM.  Rows travel in slabs of up to 256 MiB: host
// threads copy a slab from the mapping into one of two pinned buffers while the previous slab is on
// its way over PCIe (and, when the device pitch differs from the file's row size, through the
// re-pitch kernel) -- a plain hipMemcpy from pageable memory measured 10-25 GB/s here.
cobs_gpu_status upload_resident(Part& pt, const uint8_t* file) {
    const IndexMeta& m = pt.meta;
    const uint64_t src_pitch = m.page_row_bytes();
    constexpr uint64_t kSlab = 256ull << 20;
    struct Slab {
        PinnedBuf<uint8_t> host;
        DevBuf<uint8_t> dev;                 // raw rows on the device, only when they are re-pitched
        hipEvent_t done = nullptr;
        bool busy = false;
        ~Slab() { if (done) (void)hipEventDestroy(done); }
    } slab[2];
    hipStream_t stream = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } sg{stream};
    for (Slab& sl : slab) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    const size_t nthreads = std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    auto copy_in = [&](uint8_t* dst, const uint8_t* src, uint64_t bytes) {
        if (bytes < (8u << 20) || nthreads == 1) { std::memcpy(dst, src, (size_t)bytes); return; }
        std::vector<std::thread> pool;
        const uint64_t per = (bytes / nthreads + 4095) / 4096 * 4096;
        for (size_t t = 0; t < nthreads; ++t) {
            const uint64_t o = t * per;
            if (o >= bytes) break;
            pool.emplace_back([=]() { std::memcpy(dst + o, src + o, (size_t)std::min(per, bytes - o)); });
        }
        for (auto& t : pool) t.join();
    };
    int cur = 0;
    for (Chunk& c : pt.chunks) {
        for (size_t lp = 0; lp < c.vp.size(); ++lp) {
            const PageDev& pd = c.pages[lp];
            const VPage& v = c.vp[lp];
            const uint8_t* src = file + m.page_offset(v.fp);
            uint8_t* dst = c.d_data + pd.base;
            const bool straight = src_pitch == c.pitch && v.col0 == 0;      // rows already have the device pitch
            const uint64_t rows_per = std::max<uint64_t>(1, kSlab / src_pitch);
            for (uint64_t r = 0; r < pd.sig; r += rows_per) {
                const uint64_t n = std::min(rows_per, pd.sig - r), bytes = n * src_pitch;
                Slab& sl = slab[cur];
                cur ^= 1;
                if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.done)); sl.busy = false; }
                HIP_TRY(sl.host.reserve((size_t)(std::min(rows_per, pd.sig) * src_pitch)));
                copy_in(sl.host.p, src + r * src_pitch, bytes);
                if (straight) {
                    HIP_TRY(hipMemcpyAsync(dst + r * src_pitch, sl.host.p, (size_t)bytes, hipMemcpyHostToDevice, stream));
                } else {
                    HIP_TRY(sl.dev.reserve(sl.host.cap));
                    HIP_TRY(hipMemcpyAsync(sl.dev.p, sl.host.p, (size_t)bytes, hipMemcpyHostToDevice, stream));
                    RepitchArgs ra;
                    ra.src = sl.dev.p;
                    ra.dst = dst + r * c.pitch;
                    ra.rows = n;
                    ra.src_pitch = (uint32_t)src_pitch;
                    ra.dst_pitch = c.pitch;
                    ra.copy_bytes = (uint32_t)v.ncols;
                    ra.src_col0 = (uint32_t)v.col0;
                    HIP_TRY(launch_repitch(ra, stream));
                }
                HIP_TRY(hipEventRecord(sl.done, stream));
                sl.busy = true;
            }
            HIP_TRY(hipMemsetAsync(dst + pd.sig * (uint64_t)c.pitch, 0, c.pitch, stream));   // zero row
        }
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return COBS_GPU_OK;
}

SynthArgs synth_args(const Part& pt, const Chunk& c, uint8_t* data) {
    SynthArgs sa;
    sa.blob = data;
    sa.pages = c.d_pages;
    sa.seed = pt.synth_seed;
    sa.row_bytes = pt.meta.page_row_bytes();
    sa.col0 = c.vp[0].col0;
    sa.num_docs = pt.meta.doc_names.size();
    sa.page_docs = pt.meta.kind == IndexKind::Compact ? 8 * pt.meta.header_page_size : 0;
    sa.npages = (uint32_t)c.vp.size();
    sa.first_page = c.vp[0].fp;
    sa.pitch = c.pitch;
    return sa;
}

// Streamed chunk: bring it into device buffer `buf` on the copy stream (file-backed:
// DMA from the pinned mapping, or pack the needed columns into pinned staging first;
// procedural: regenerate).
cobs_gpu_status stream_chunk_in(cobs_gpu_index* ix, Part& pt, const Chunk& c, int buf) {
    StreamBufs& sb = ix->stream;
    uint8_t* dev = sb.sbuf[buf].p;
    if (pt.synthetic) {
        HIP_TRY(launch_synth(synth_args(pt, c, dev), sb.copy_stream));
        return COBS_GPU_OK;
    }
    const IndexMeta& m = pt.meta;
    const uint64_t prb = m.page_row_bytes();
    if (!pt.file_pinned) HIP_TRY(sb.stage[buf].reserve(sb.stage_need));
    uint8_t* host = sb.stage[buf].p;
    uint64_t hoff = 0;
    for (size_t i = 0; i < c.vp.size(); ++i) {
        const VPage& v = c.vp[i];
        const PageDev& pd = c.pages[i];
        const uint8_t* src = pt.file->data() + m.page_offset(v.fp);
        uint8_t* dst = dev + pd.base;
        if (pt.file_pinned) {
            HIP_TRY(hipMemcpy2DAsync(dst, c.pitch, src + v.col0, (size_t)prb, (size_t)v.ncols, (size_t)pd.sig,
                                     hipMemcpyHostToDevice, sb.copy_stream));
            HIP_TRY(hipMemsetAsync(dst + pd.sig * (uint64_t)c.pitch, 0, c.pitch, sb.copy_stream));
            continue;
        }
        uint8_t* hp = host + hoff;
        {   // pack the needed columns into pinned staging with a few host threads
            const unsigned nthr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, pd.sig * v.ncols >> 24));
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nthr; ++t) {
                const uint64_t r0 = pd.sig * t / nthr, r1 = pd.sig * (t + 1) / nthr;
                pool.emplace_back([=]() {
                    if (v.ncols == prb) {
                        std::memcpy(hp + r0 * prb, src + r0 * prb, (size_t)((r1 - r0) * prb));
                    } else {
                        for (uint64_t r = r0; r < r1; ++r)
                            std::memcpy(hp + r * v.ncols, src + r * prb + v.col0, (size_t)v.ncols);
                    }
                });
            }
            for (auto& th : pool) th.join();
        }
        if (c.pitch == v.ncols)
            HIP_TRY(hipMemcpyAsync(dst, hp, (size_t)(pd.sig * v.ncols), hipMemcpyHostToDevice, sb.copy_stream));
        else
            HIP_TRY(hipMemcpy2DAsync(dst, c.pitch, hp, (size_t)v.ncols, (size_t)v.ncols, (size_t)pd.sig,
                                     hipMemcpyHostToDevice, sb.copy_stream));
        HIP_TRY(hipMemsetAsync(dst + pd.sig * (uint64_t)c.pitch, 0, c.pitch, sb.copy_stream));
        hoff += pd.sig * v.ncols;
    }
    return COBS_GPU_OK;
}

// options fields appended after the first release are honoured only if the caller's struct has them
bool has_field(const cobs_gpu_options* o, size_t end_offset) { return o && o->struct_size >= end_offset; }

cobs_gpu_status read_options(const cobs_gpu_options* o, cobs_gpu_index* ix) {
    ix->tune = Tuning::from_env();
    ix->shard_rank = 0;
    ix->shard_count = 1;
    ix->shard_mode = 0;
    ix->hbm_budget = 0;
    if (!o) return COBS_GPU_OK;
    if (o->shard_count > 1) {
        if (o->shard_rank >= o->shard_count) return fail(COBS_GPU_ERR_ARG, "shard_rank >= shard_count");
        ix->shard_rank = o->shard_rank;
        ix->shard_count = o->shard_count;
    }
    if (o->waves_per_group == 1 || o->waves_per_group == 2 || o->waves_per_group == 4)
        ix->waves_per_group = o->waves_per_group;
    if (o->shard_mode > 1) return fail(COBS_GPU_ERR_ARG, "unknown shard_mode");
    ix->shard_mode = o->shard_mode;
    if (has_field(o, offsetof(cobs_gpu_options, hbm_budget_bytes) + sizeof(uint64_t))) ix->hbm_budget = o->hbm_budget_bytes;
    return COBS_GPU_OK;
}

// row bytes one hash lookup gathers from this part (all held slices)
uint64_t gathered_row_bytes(const Part& p) {
    uint64_t n = 0;
    for (const VPage& v : p.held) n += v.ncols;
    return n;
}

}  // namespace

namespace cobs_amd {

// ---------------------------------------------------------------------------
// ranking (counts_to_result, reference classic_search.cpp:109-202)

bool hit_before(const cobs_gpu_hit& a, const cobs_gpu_hit& b) {
    if (a.score != b.score) return a.score > b.score;
    if (a.file_no != b.file_no) return a.file_no < b.file_no;
    return a.doc < b.doc;
}

bool doc_before(const cobs_gpu_hit& a, const cobs_gpu_hit& b) {
    if (a.file_no != b.file_no) return a.file_no < b.file_no;
    return a.doc < b.doc;
}

// total number of hashes of query `q` over all files: the reference's max_counts
uint64_t total_hashes(const cobs_gpu_batch* b, size_t q) {
    uint64_t n = 0;
    for (const Part& p : b->ix->parts)
        n += (uint64_t)(b->lens[q] - p.meta.term_size + 1) * p.meta.num_hashes;
    return n;
}

uint32_t threshold_for(double threshold, uint64_t terms) {
    // classic_search.cpp:446-448: std::ceil(threshold * T) in double
    const double v = std::ceil(threshold * (double)terms);
    if (!(v > 0)) return 0;
    if (v >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

}  // namespace cobs_amd

// shared with build.cpp
__attribute__((visibility("hidden"))) cobs_gpu_status cobs_gpu_set_error(cobs_gpu_status st, const char* msg) { return fail(st, msg ? msg : ""); }
// ===========================================================================
// C ABI

extern "C" {

uint32_t cobs_gpu_abi_version(void) { return COBS_GPU_ABI_VERSION; }

const char* cobs_gpu_last_error(void) { return g_last_error.c_str(); }

int cobs_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// stage every part: resident chunks are uploaded (file) or generated (procedural), streamed
// parts keep their source and get the handle's shared buffers
static cobs_gpu_status stage_index(cobs_gpu_index* ix, std::vector<std::unique_ptr<MappedFile>>& files) {
    for (size_t i = 0; i < ix->parts.size(); ++i) {
        Part& pt = ix->parts[i];
        cobs_gpu_status st = alloc_part(ix, pt);
        if (st != COBS_GPU_OK) return st;
        if (pt.chunks.empty()) continue;
        if (pt.built) {          // rows are written by the caller (index construction straight into HBM)
            for (Chunk& c : pt.chunks) HIP_TRY(hipMemset(c.d_data, 0, c.bytes));
            continue;
        }
        if (pt.synthetic) {
            if (!pt.streamed) {
                for (Chunk& c : pt.chunks) HIP_TRY(launch_synth(synth_args(pt, c, c.d_data), nullptr));
                HIP_TRY(hipStreamSynchronize(nullptr));
            }
            continue;
        }
        if (pt.streamed) {
            pt.file = std::move(files[i]);       // chunks are read from the mapping at every pass
            // Pin the read-only mapping so that the copy engine reads it directly (no packing
            // through staging buffers).  Not every kernel/driver allows pinning file pages;
            // if it fails the staged path is used.
            if (!ix->tune.no_pin) {
                hipError_t pe = hipHostRegister(const_cast<uint8_t*>(pt.file->data()), pt.file->size(),
                                                hipHostRegisterReadOnly);
                if (pe != hipSuccess) {
                    (void)hipGetLastError();
                    pe = hipHostRegister(const_cast<uint8_t*>(pt.file->data()), pt.file->size(), hipHostRegisterDefault);
                }
                if (pe == hipSuccess) pt.file_pinned = true;
                else (void)hipGetLastError();
            }
            if (pt.file_pinned) {
                // the device-visible address of the mapping and, per chunk, where its slices start in the
                // file: the row-selective pass (fetch_kernels.hip) reads looked-up rows straight from it
                void* dp = nullptr;
                if (hipHostGetDevicePointer(&dp, const_cast<uint8_t*>(pt.file->data()), 0) == hipSuccess && dp) {
                    pt.file_dev = static_cast<const uint8_t*>(dp);
                    for (Chunk& c : pt.chunks) {
                        std::vector<uint64_t> src(c.vp.size());
                        for (size_t k = 0; k < c.vp.size(); ++k) src[k] = pt.meta.page_offset(c.vp[k].fp) + c.vp[k].col0;
                        HIP_TRY(hipMalloc((void**)&c.d_src, 8 * src.size()));
                        HIP_TRY(hipMemcpy(c.d_src, src.data(), 8 * src.size(), hipMemcpyHostToDevice));
                        for (auto& p2 : c.d_pages2) HIP_TRY(hipMalloc((void**)&p2, sizeof(PageDev) * c.vp.size()));
                    }
                    if (!ix->stream.hashed) HIP_TRY(hipEventCreateWithFlags(&ix->stream.hashed, hipEventDisableTiming));
                } else {
                    (void)hipGetLastError();
                }
            }
        } else {
            st = upload_resident(pt, files[i]->data());
            if (st != COBS_GPU_OK) return st;
        }
    }
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_open(const char* const* paths, size_t n_paths,
                              const cobs_gpu_options* opts, cobs_gpu_index** out) {
    if (!out) return fail(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!paths || n_paths == 0) return fail(COBS_GPU_ERR_ARG, "no index paths");
    return guarded([&]() -> cobs_gpu_status {
        // parse all headers first: format errors are reported even without a device
        std::vector<std::unique_ptr<MappedFile>> files;
        std::unique_ptr<cobs_gpu_index> ix(new cobs_gpu_index);
        for (size_t i = 0; i < n_paths; ++i) {
            if (!paths[i]) return fail(COBS_GPU_ERR_ARG, "NULL path");
            std::string err;
            files.emplace_back(new MappedFile);
            if (!files.back()->open(paths[i], err)) return fail(COBS_GPU_ERR_OPEN, err);
            Part pt;
            if (!parse_index_header(files.back()->data(), files.back()->size(), pt.meta, err))
                return fail(COBS_GPU_ERR_FORMAT, std::string("Could not open index path \"") + paths[i] + "\": " + err);
            ix->parts.push_back(std::move(pt));
        }
        cobs_gpu_status st = read_options(opts, ix.get());
        if (st != COBS_GPU_OK) return st;
        st = plan_index(ix.get());
        if (st != COBS_GPU_OK) return st;
        st = select_device(opts, &ix->device);
        if (st != COBS_GPU_OK) return st;
        st = stage_index(ix.get(), files);
        if (st != COBS_GPU_OK) return st;
        *out = ix.release();
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_open_synthetic(const cobs_gpu_synth* d, const cobs_gpu_options* opts,
                                        cobs_gpu_index** out) {
    if (!out) return fail(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!d || !d->signature_sizes || d->num_pages == 0 || d->kind > 1)
        return fail(COBS_GPU_ERR_ARG, "bad synthetic index description");
    if (d->kind == 1 && d->page_size == 0) return fail(COBS_GPU_ERR_ARG, "page_size is zero");
    if (d->kind == 0 && d->num_pages != 1) return fail(COBS_GPU_ERR_ARG, "classic index has one sub-index");
    if (d->kind == 1 && d->num_docs > (uint64_t)d->num_pages * 8 * d->page_size)
        return fail(COBS_GPU_ERR_ARG, "more documents than sub-index slots");
    if (d->num_docs == 0 || d->num_docs > 0xFFFFFFF0ull) return fail(COBS_GPU_ERR_ARG, "bad num_docs");
    return guarded([&]() -> cobs_gpu_status {
        std::unique_ptr<cobs_gpu_index> ix(new cobs_gpu_index);
        Part pt;
        pt.meta.kind = d->kind ? IndexKind::Compact : IndexKind::Classic;
        pt.meta.term_size = d->term_size;
        pt.meta.canonicalize = (uint8_t)d->canonicalize;
        pt.meta.num_hashes = d->num_hashes;
        pt.meta.header_page_size = d->kind ? d->page_size : 0;
        pt.meta.signature_sizes.assign(d->signature_sizes, d->signature_sizes + d->num_pages);
        {   // geometry first: an absurd size must not cost a multi-gigabyte name table
            IndexMeta probe = pt.meta;
            if (!d->kind) probe.doc_names.resize(1);
            const uint64_t prb = d->kind ? d->page_size : (d->num_docs + 7) / 8;
            if (prb == 0 || prb > (1ull << 28)) return fail(COBS_GPU_ERR_UNSUPPORTED, "row too wide");
            for (uint64_t sg : probe.signature_sizes) {
                uint64_t bytes = 0;
                if (sg == 0 || sg > (1ull << 46) || __builtin_mul_overflow(sg + 1, round_up(prb, 128), &bytes) ||
                    bytes > (1ull << 47))
                    return fail(COBS_GPU_ERR_UNSUPPORTED, "index geometry too large (a sub-index beyond 128 TiB)");
            }
        }
        pt.meta.doc_names.resize(d->num_docs);
        char nm[32];
        for (uint64_t i = 0; i < d->num_docs; ++i) {      // names as classic_construct_random, classic_index.cpp:668-670
            std::snprintf(nm, sizeof nm, "file_%06u", (unsigned)i);
            pt.meta.doc_names[i] = nm;
        }
        pt.synthetic = true;
        pt.synth_seed = d->seed;
        ix->parts.push_back(std::move(pt));
        cobs_gpu_status st = read_options(opts, ix.get());
        if (st != COBS_GPU_OK) return st;
        st = plan_index(ix.get());
        if (st != COBS_GPU_OK) return st;
        st = select_device(opts, &ix->device);
        if (st != COBS_GPU_OK) return st;
        std::vector<std::unique_ptr<MappedFile>> none;
        st = stage_index(ix.get(), none);
        if (st != COBS_GPU_OK) return st;
        *out = ix.release();
        return COBS_GPU_OK;
    });
}

void cobs_gpu_close(cobs_gpu_index* ix) { delete ix; }

}  // extern "C"

// An all-zero resident index of the given geometry (unsharded, no budget): index construction
// fills its rows in place (build.cpp), so that build -> query needs no file.
cobs_gpu_status cobs_amd::open_zeroed(IndexMeta&& meta, const cobs_gpu_options* opts, cobs_gpu_index** out) {
    return guarded([&]() -> cobs_gpu_status {
        std::unique_ptr<cobs_gpu_index> ix(new cobs_gpu_index);
        Part pt;
        pt.meta = std::move(meta);
        pt.built = true;
        ix->parts.push_back(std::move(pt));
        cobs_gpu_status st = read_options(opts, ix.get());
        if (st != COBS_GPU_OK) return st;
        if (ix->shard_count > 1 || ix->hbm_budget)
            return fail(COBS_GPU_ERR_UNSUPPORTED, "an index is built into HBM whole: no shard, no budget");
        st = plan_index(ix.get());
        if (st != COBS_GPU_OK) return st;
        st = select_device(opts, &ix->device);
        if (st != COBS_GPU_OK) return st;
        std::vector<std::unique_ptr<MappedFile>> none;
        st = stage_index(ix.get(), none);
        if (st != COBS_GPU_OK) return st;
        *out = ix.release();
        return COBS_GPU_OK;
    });
}

extern "C" {

cobs_gpu_status cobs_gpu_plan_shards(const char* path, uint32_t shard_count, uint32_t shard_mode,
                                     uint64_t* slot_begin, uint64_t* slot_count, uint64_t* bytes) {
    if (!path || !slot_begin || !slot_count || shard_count == 0 || shard_mode > 1)
        return fail(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        MappedFile file;
        std::string err;
        if (!file.open(path, err)) return fail(COBS_GPU_ERR_OPEN, err);
        cobs_gpu_index ix;
        ix.tune = Tuning::from_env();
        ix.shard_count = shard_count;
        ix.shard_mode = shard_mode;
        for (uint32_t r = 0; r < shard_count; ++r) {
            Part pt;
            if (!parse_index_header(file.data(), file.size(), pt.meta, err))
                return fail(COBS_GPU_ERR_FORMAT, std::string("Could not open index path \"") + path + "\": " + err);
            ix.shard_rank = r;
            cobs_gpu_status st = plan_part(pt, &ix);
            if (st != COBS_GPU_OK) return st;
            slot_begin[r] = pt.slot_begin;
            slot_count[r] = pt.slot_count;
            if (bytes) bytes[r] = pt.resident_bytes;
        }
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_set_tuning(cobs_gpu_index* ix, const char* key, int64_t value) {
    if (!ix || !key) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    const std::string k = key;
    Tuning& t = ix->tune;
    if (k == "waves") {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(COBS_GPU_ERR_ARG, "waves: 0, 1, 2 or 4");
        t.waves = (uint32_t)value;
    } else if (k == "tile_w") {
        if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64)
            return fail(COBS_GPU_ERR_ARG, "tile_w: 0, 4, 8, 16, 32 or 64");
        t.tile_w = (uint32_t)value;
    } else if (k == "mq") {
        t.mq = value < 0 ? -1 : value != 0;
    } else if (k == "pass_bytes") {
        t.pass_bytes = value > 0 ? (uint64_t)value : 16ull << 30;
    } else if (k == "pipe_chars") {
        t.pipe_chars = value < 0 ? 4ull << 20 : (uint64_t)value;
    } else if (k == "graph") {
        t.graph = value < 0 ? -1 : value != 0;
    } else if (k == "lds_staged") {
        t.lds_staged = value > 0;
    } else if (k == "device_rank") {
        t.device_rank = value != 0;
    } else if (k == "tile_topk") {
        t.tile_topk = value != 0;
    } else if (k == "row_fetch") {
        t.row_fetch = value != 0;
    } else if (k == "row_fetch_alpha") {
        t.row_fetch_alpha = value >= 0 ? (uint32_t)std::min<int64_t>(value, 1 << 20) : 1;   // 0: whenever the rows fit
    } else if (k == "phase_slots") {
        t.phase_slots = value > 0 ? (uint32_t)std::min<int64_t>(value, 1 << 20) : 0;
    } else {
        return fail(COBS_GPU_ERR_ARG, "unknown tuning key (waves, tile_w, mq, pass_bytes, pipe_chars, graph, lds_staged, device_rank, tile_topk, row_fetch, row_fetch_alpha)");
    }
    return COBS_GPU_OK;
}

size_t cobs_gpu_num_files(const cobs_gpu_index* ix) { return ix ? ix->parts.size() : 0; }

cobs_gpu_status cobs_gpu_info(const cobs_gpu_index* ix, size_t f, cobs_gpu_index_info* o) {
    if (!ix || !o || f >= ix->parts.size()) return fail(COBS_GPU_ERR_ARG, "bad file number");
    const Part& p = ix->parts[f];
    std::memset(o, 0, sizeof *o);
    o->kind = (uint32_t)p.meta.kind;
    o->term_size = p.meta.term_size;
    o->canonicalize = p.meta.canonicalize;
    o->num_pages = p.meta.num_pages();
    o->num_hashes = p.meta.num_hashes;
    o->page_size = p.meta.page_size();
    o->row_size = p.meta.row_size();
    o->counts_size = p.meta.counts_size();
    o->num_docs = p.meta.doc_names.size();
    o->doc_offset = p.doc_offset;
    o->hbm_bytes = p.hbm_bytes;
    o->first_page = p.first_page;
    o->end_page = p.end_page;
    o->slot_begin = p.slot_begin;
    o->slot_count = p.slot_count;
    o->local_offset = p.local_offset;
    return COBS_GPU_OK;
}

uint64_t cobs_gpu_signature_size(const cobs_gpu_index* ix, size_t f, uint32_t page) {
    if (!ix || f >= ix->parts.size() || page >= ix->parts[f].meta.num_pages()) return 0;
    return ix->parts[f].meta.signature_sizes[page];
}

const char* cobs_gpu_doc_name(const cobs_gpu_index* ix, size_t f, uint64_t doc) {
    if (!ix || f >= ix->parts.size() || doc >= ix->parts[f].meta.doc_names.size()) return "";
    return ix->parts[f].meta.doc_names[doc].c_str();
}

uint64_t cobs_gpu_total_counts(const cobs_gpu_index* ix) { return ix ? ix->total_counts : 0; }
uint64_t cobs_gpu_local_counts(const cobs_gpu_index* ix) { return ix ? ix->local_counts : 0; }

// resident slice of file-level sub-index `page` (streamed chunks cannot be read back)
static cobs_gpu_status find_resident(const cobs_gpu_index* ix, size_t f, uint32_t page, const Chunk** c,
                                     const PageDev** pd) {
    if (!ix || f >= ix->parts.size()) return fail(COBS_GPU_ERR_ARG, "bad argument");
    const Part& p = ix->parts[f];
    if (p.streamed) return fail(COBS_GPU_ERR_UNSUPPORTED, "index is streamed, rows are not resident");
    for (const Chunk& ch : p.chunks)
        for (size_t i = 0; i < ch.vp.size(); ++i)
            if (ch.vp[i].fp == page) {
                *c = &ch;
                *pd = &ch.pages[i];
                return COBS_GPU_OK;
            }
    return fail(COBS_GPU_ERR_ARG, "sub-index not held by this shard");
}

cobs_gpu_status cobs_gpu_page_columns(const cobs_gpu_index* ix, size_t f, uint32_t page, uint64_t* col0,
                                      uint64_t* ncols) {
    if (!ix || f >= ix->parts.size() || !col0 || !ncols) return fail(COBS_GPU_ERR_ARG, "bad argument");
    *col0 = *ncols = 0;
    for (const VPage& v : ix->parts[f].held)
        if (v.fp == page) {
            *col0 = v.col0;
            *ncols = v.ncols;
            return COBS_GPU_OK;
        }
    return COBS_GPU_OK;      // not held: zero columns
}

cobs_gpu_status cobs_gpu_read_row(const cobs_gpu_index* ix, size_t f, uint32_t page, uint64_t row,
                                  uint8_t* out, size_t n) {
    const Chunk* c = nullptr;
    const PageDev* pd = nullptr;
    if (!out) return fail(COBS_GPU_ERR_ARG, "bad argument");
    cobs_gpu_status st = find_resident(ix, f, page, &c, &pd);
    if (st != COBS_GPU_OK) return st;
    if (row > pd->sig || n > c->pitch) return fail(COBS_GPU_ERR_ARG, "row or length out of range");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipMemcpy(out, c->d_data + pd->base + row * (uint64_t)c->pitch, n, hipMemcpyDeviceToHost));
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_read_rows(const cobs_gpu_index* ix, size_t f, uint32_t page, uint64_t row0,
                                   uint64_t nrows, uint8_t* out, size_t out_pitch) {
    const Chunk* c = nullptr;
    const PageDev* pd = nullptr;
    if (!out) return fail(COBS_GPU_ERR_ARG, "bad argument");
    cobs_gpu_status st = find_resident(ix, f, page, &c, &pd);
    if (st != COBS_GPU_OK) return st;
    if (row0 + nrows > pd->sig + 1 || out_pitch < pd->valid_bytes) return fail(COBS_GPU_ERR_ARG, "rows or pitch out of range");
    HIP_TRY(hipSetDevice(ix->device));
    if (nrows)
        HIP_TRY(hipMemcpy2D(out, out_pitch, c->d_data + pd->base + row0 * (uint64_t)c->pitch, c->pitch,
                            (size_t)pd->valid_bytes, (size_t)nrows, hipMemcpyDeviceToHost));
    return COBS_GPU_OK;
}

// ---------------------------------------------------------------------------
// batches

cobs_gpu_status cobs_gpu_batch_create(cobs_gpu_index* ix, size_t max_queries, size_t max_query_len,
                                      cobs_gpu_batch** out) {
    if (!ix || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    *out = nullptr;
    return guarded([&]() -> cobs_gpu_status {
    HIP_TRY(hipSetDevice(ix->device));
    std::unique_ptr<cobs_gpu_batch> b(new cobs_gpu_batch);
    b->ix = ix;
    b->max_queries = max_queries;
    b->max_len = max_query_len;
    b->work.resize(ix->parts.size());
    for (auto& r : b->ev) for (auto& e : r) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&b->run_done, hipEventDisableTiming));
    HIP_TRY(b->flags.reserve(4));
    *out = b.release();
    return COBS_GPU_OK;
    });
}

void cobs_gpu_batch_destroy(cobs_gpu_batch* b) { delete b; }

// Uploads go through `up` (asynchronously where the source is pinned); wait = false leaves them
// in flight: the caller orders its kernels after them on the same stream.
}  // extern "C"

cobs_gpu_status cobs_amd::set_queries_on(cobs_gpu_batch* b, const char* const* queries, const size_t* lens,
                                         size_t nq, hipStream_t up, bool wait, size_t* bad_query, size_t index_base) {
    if (!b || (nq && (!queries || !lens))) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    cobs_gpu_index* ix = b->ix;
    HIP_TRY(hipSetDevice(ix->device));
    // the upload overwrites buffers a run still in flight would read: wait for a run nobody
    // synced -- on that run's own event, other handles' streams on the device keep going
    if (b->ran && !b->synced) HIP_TRY(hipEventSynchronize(b->run_done));
    b->ran = false;
    b->nq = 0;
    if (nq >= 0xFFFFFFFEull) return fail(COBS_GPU_ERR_ARG, "too many queries");
    // reference checks, classic_search.cpp:431-433 and :453-504
    uint32_t max_term = 0, min_term = 0xFFFFFFFFu;
    for (const Part& p : ix->parts) {
        max_term = std::max(max_term, p.meta.term_size);
        min_term = std::min(min_term, p.meta.term_size);
    }
    uint64_t max_terms = 1;
    for (size_t q = 0; q < nq; ++q) {
        if (bad_query) *bad_query = q;
        if (!queries[q]) return fail(COBS_GPU_ERR_ARG, "NULL query (query " + std::to_string(index_base + q) + ")");
        if (lens[q] < max_term)
            return fail(COBS_GPU_ERR_QUERY_TOO_SHORT, "query too short, needs to be at least " +
                        std::to_string(max_term) + " characters long (query " + std::to_string(index_base + q) + ")");
        if (lens[q] - max_term >= 0xFFFFFFFFull || lens[q] >= 0xFFFFFFF0ull)
            return fail(COBS_GPU_ERR_QUERY_TOO_LONG, "query too long (query " + std::to_string(index_base + q) + ")");
        max_terms = std::max<uint64_t>(max_terms, lens[q] - min_term + 1);
    }
    if (bad_query) *bad_query = 0;
    const int planes = scan_planes_for(max_terms);
    if (planes < 0) return fail(COBS_GPU_ERR_QUERY_TOO_LONG, "query too long");
    b->planes = planes;
    b->max_terms = max_terms;
    b->elem_bytes = scan_score_bytes(planes);

    // thread spans of K1: every character and every (padded) term of every file
    b->lens.resize(nq);
    b->span_off.resize(nq + 1);
    uint64_t off = 0;
    for (size_t q = 0; q < nq; ++q) {
        b->lens[q] = (uint32_t)lens[q];
        b->span_off[q] = off;
        uint64_t span = lens[q];
        for (const Part& p : ix->parts)
            span = std::max<uint64_t>(span, round_up(lens[q] - p.meta.term_size + 1, 8) + 8);   // + padding block
        off += round_up(span, 8);
    }
    b->span_off[nq] = off;
    // upload layout: text (+ 64: K1 reads whole dwords around a k-mer) | span_off | q_len | blk_off per file
    const size_t o_span = (size_t)round_up(off + 64, 16);
    const size_t o_qlen = o_span + (size_t)round_up(8 * (nq + 1), 16);
    const size_t o_blk = o_qlen + (size_t)round_up(4 * std::max<size_t>(nq, 1), 16);
    const size_t blk_stride = (size_t)round_up(8 * (nq + 1), 16);
    const size_t upload_bytes = o_blk + blk_stride * ix->parts.size();
    HIP_TRY(b->h_text.reserve(upload_bytes));
    HIP_TRY(b->text.reserve(upload_bytes));
    std::memset(b->h_text.p, 0, o_span);
    for (size_t q = 0; q < nq; ++q) std::memcpy(b->h_text.p + b->span_off[q], queries[q], lens[q]);
    std::memcpy(b->h_text.p + o_span, b->span_off.data(), 8 * (nq + 1));
    if (nq) std::memcpy(b->h_text.p + o_qlen, b->lens.data(), 4 * nq);
    b->d_span_off = reinterpret_cast<const uint64_t*>(b->text.p + o_span);
    b->d_qlen = reinterpret_cast<const uint32_t*>(b->text.p + o_qlen);

    uint64_t algo_bytes = 0, lookups = 0, table_bytes = 0;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        const Part& p = ix->parts[f];
        PartWork& w = b->work[f];
        w.h_blk_off.resize(nq + 1);
        uint64_t blk = 0;
        for (size_t q = 0; q < nq; ++q) {
            w.h_blk_off[q] = blk;
            const uint64_t T = lens[q] - p.meta.term_size + 1;
            blk += (T + 7) / 8;
            lookups += T;
            // T * H * (row bytes gathered) + score bytes written
            algo_bytes += T * p.meta.num_hashes * gathered_row_bytes(p);
        }
        w.h_blk_off[nq] = blk;
        // per (query, sub-index): its 8-term blocks plus one padding block
        const uint64_t idx_words = p.idx64 ? 2 : 1;      // u32 words per table entry
        w.table_entries = (blk + nq) * 8 * p.meta.num_hashes * p.num_tpages() * idx_words;
        table_bytes += w.table_entries * 4;
        if (w.table_entries >= (1ull << 40)) return fail(COBS_GPU_ERR_CAPACITY, "batch too large");
        HIP_TRY(w.table.reserve((size_t)w.table_entries));
        HIP_TRY(w.thr.reserve(nq));
        std::memcpy(b->h_text.p + o_blk + f * blk_stride, w.h_blk_off.data(), 8 * (nq + 1));
        w.blk_off = reinterpret_cast<const uint64_t*>(b->text.p + o_blk + f * blk_stride);
    }
    HIP_TRY(hipMemcpyAsync(b->text.p, b->h_text.p, upload_bytes, hipMemcpyHostToDevice, up));
    b->algo_row_bytes = algo_bytes;                          // the score bytes are added by the run that writes them
    // selection pool: room for 1024 hits per query, at least 1 Mi entries
    const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(1u << 20, nq * 1024ull), 1ull << 26);
    HIP_TRY(b->hits.reserve((size_t)want));
    b->hit_cap = (uint32_t)b->hits.cap;
    HIP_TRY(b->h_thr_stage.reserve(std::max<size_t>(nq * ix->parts.size(), 1)));
    b->stats[0] = algo_bytes + (uint64_t)nq * ix->local_counts * b->elem_bytes;      // until a run says otherwise
    b->stats[1] = 0;
    b->stats[2] = lookups;
    b->stats[3] = table_bytes;
    b->nq = nq;
    if (wait) HIP_TRY(hipStreamSynchronize(up));
    return COBS_GPU_OK;
}

extern "C" cobs_gpu_status cobs_gpu_batch_set_queries(cobs_gpu_batch* b, const char* const* queries,
                                                      const size_t* lens, size_t nq) {
    return guarded([&]() { return set_queries_on(b, queries, lens, nq, nullptr, true, nullptr); });
}

// want_counts = false: the caller only needs the selected hits (threshold > 0, no top-k), so the
// scan does not write the score rows (for reads they are up to a third of the traffic).
// what a run leaves behind on the host side of the batch (a replayed graph sets the same)
static void set_run_state(cobs_gpu_batch* b, double threshold, size_t topk, bool want_counts) {
    cobs_gpu_index* ix = b->ix;
    b->ran = false;
    b->synced = false;
    b->pool_fetched = false;
    b->topk_fetched = false;
    b->rows_q0 = b->rows_q1 = 0;
    b->view_global = false;
    b->pool_global = false;
    b->topk_stride = 0;
    b->graph_run = false;
    b->threshold = threshold;
    // K3 (exact top-k on the device, every score width) needs a bounded k
    const bool use_topk = topk > 0 && topk <= 65536 &&
                          (uint64_t)topk * std::max<size_t>(b->nq, 1) * ix->parts.size() <= (1ull << 27);
    b->topk_k = use_topk ? (uint32_t)topk : 0;
    b->topk_sorted = use_topk && topk <= kTopkSortLimit;
    // with K3 the threshold is applied there; otherwise K2 selects into the hit pool
    b->selected = threshold > 0.0 && !use_topk;
    // a top-k pass whose caller does not want the score rows: K2 leaves the k best of every tile and K3 merges
    // those (no score matrix at all) -- where that epilogue exists, for a k a tile can hold, and unless a query
    // has a single hash in total (its result is index order, which only the rows give: classic_search.cpp:136,179)
    b->topk_direct = false;
    if (use_topk && !want_counts && topk <= kTileTopkMax && !ix->tune.lds_staged && ix->tune.tile_topk != 0) {
        bool ok = b->nq > 0;
        for (const Part& p : ix->parts) ok = ok && scan_has_tile_topk((uint32_t)p.meta.num_hashes, p.idx64);
        for (size_t q = 0; ok && q < b->nq; ++q) ok = total_hashes(b, q) > 1;
        b->topk_direct = ok;
    }
    b->have_counts = want_counts || (!b->selected && !b->topk_direct);
}

// The per-(file, query) thresholds ceil(threshold * T) (classic_search.cpp:444-449) in the pinned
// buffer the H2D copies of a run read.  A captured graph holds those copies as nodes that read the
// buffer when the graph is LAUNCHED, and the buffer is shared by every shape of the batch: a replay
// has to write its own thresholds first (whatever ran in between left its own there).
static void stage_thresholds(cobs_gpu_batch* b, double threshold) {
    const cobs_gpu_index* ix = b->ix;
    const size_t nq = b->nq;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        uint32_t* stage = b->h_thr_stage.p + f * nq;
        for (size_t q = 0; q < nq; ++q)
            stage[q] = threshold_for(threshold, (uint64_t)b->lens[q] - ix->parts[f].meta.term_size + 1);
    }
}

cobs_gpu_status cobs_amd::run_impl(cobs_gpu_batch* b, double threshold, size_t topk, void* hip_stream,
                                   bool want_counts) {
    if (!b) return fail(COBS_GPU_ERR_ARG, "NULL batch");
    cobs_gpu_index* ix = b->ix;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(ix->device));
    set_run_state(b, threshold, topk, want_counts);
    const size_t nq = b->nq;
    const bool use_topk = b->topk_k != 0;
    // score rows are allocated by the first run that writes them (a hits-only caller never pays
    // for them: 100k reads x 100k documents would be 10 GB)
    if (b->have_counts) HIP_TRY(b->counts.reserve((size_t)(nq * ix->local_counts * b->elem_bytes)));
    const bool need_thr = threshold > 0.0;
    if (use_topk) {
        HIP_TRY(b->topk_out.reserve((size_t)topk * std::max<size_t>(nq, 1) * ix->parts.size()));
        HIP_TRY(b->topk_cnt.reserve(std::max<size_t>(nq, 1) * ix->parts.size()));
    }
    // device flags: first invalid query = none, selected hits = 0
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->flags.p, 0, 4, st));      // all zero: one fill
    if (need_thr) {
        stage_thresholds(b, threshold);
        for (size_t f = 0; f < ix->parts.size(); ++f)
            if (nq) HIP_TRY(hipMemcpyAsync(b->work[f].thr.p, b->h_thr_stage.p + f * nq, 4 * nq, hipMemcpyHostToDevice, st));
    }
    // scan geometry of every (file, chunk); with tile-level top-k also the files' places in the candidate pool
    std::vector<std::vector<ScanGeom>> geoms(ix->parts.size());
    std::vector<uint64_t> cand_off(ix->parts.size() + 1, 0);
    std::vector<uint32_t> cand_tiles(ix->parts.size(), 0), cand_stride(ix->parts.size(), 0);
    for (size_t f = 0; f < ix->parts.size() && nq; ++f) {
        const Part& p = ix->parts[f];
        for (const Chunk& c : p.chunks) {
            geoms[f].push_back(scan_geometry(c, b->work[f].h_blk_off[nq] / nq, (b->max_terms + 7) / 8, p.meta.num_hashes,
                                             ix->waves_per_group, b->planes, p.idx64, ix->tune));
            cand_tiles[f] += (c.total_chunks + geoms[f].back().tile_w - 1) / geoms[f].back().tile_w;
        }
        cand_stride[f] = (uint32_t)round_up((uint64_t)cand_tiles[f] * topk, 8);
        cand_off[f + 1] = cand_off[f] + (b->topk_direct ? (uint64_t)nq * cand_stride[f] : 0);
    }
    if (b->topk_direct) {
        if (cand_off.back() > (1ull << 31)) {            // 16 GiB of candidates: take the score rows instead
            b->topk_direct = false;
            b->have_counts = true;
            HIP_TRY(b->counts.reserve((size_t)(nq * ix->local_counts * b->elem_bytes)));
        } else {
            HIP_TRY(b->cand.reserve((size_t)cand_off.back()));
        }
    }
    hipEvent_t* ev = b->ev[b->run_seq % cobs_gpu_batch::kRing];
    HIP_TRY(hipEventRecord(ev[0], st));
    bool hash_marked = false;
    uint64_t launches = 0;
    StreamBufs& sbufs = ix->stream;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        Part& p = ix->parts[f];
        if (nq == 0 || p.chunks.empty()) continue;
        {   // K1 once per file and pass: the row-index table covers every held sub-index,
            // the chunks (launches) of the file pick their sub-indexes by PageDev::tpage
            HashArgs ha;
            ha.text = b->text.p;
            ha.span_off = b->d_span_off;
            ha.q_len = b->d_qlen;
            ha.blk_off = b->work[f].blk_off;
            ha.pages = p.d_tpages;
            ha.table = b->work[f].table.p;
            ha.err_query = b->flags.p;
            ha.nq = (uint32_t)nq;
            ha.npages = p.num_tpages();
            ha.term_size = p.meta.term_size;
            ha.canonicalize = p.meta.canonicalize;
            ha.num_hashes = (uint32_t)p.meta.num_hashes;
            ha.idx64 = p.idx64 ? 1u : 0u;
            HIP_TRY(launch_hash(ha, b->span_off[nq], st));
            if (!hash_marked) {      // K1 / K2 split of the timing events: first file only
                HIP_TRY(hipEventRecord(ev[1], st));
                hash_marked = true;
            }
        }
        uint32_t tile_base = 0;
        bool fetch_ready = false;
        if (p.streamed && p.file_dev && ix->tune.row_fetch != 0) {
            // a row-selective chunk gets its own row-index table (one per stream buffer); sized before the
            // chunk loop, when no scan of this handle is reading the old ones any more
            const size_t need = (size_t)b->work[f].table_entries * 4;
            for (int i = 0; i < 2; ++i) {
                if (sbufs.table2[i].cap >= need) continue;
                if (sbufs.used[i]) HIP_TRY(hipEventSynchronize(sbufs.scanned[i]));
                HIP_TRY(sbufs.table2[i].reserve(need));
            }
        }
        for (size_t ci = 0; ci < p.chunks.size(); ++ci) {
            const Chunk& c = p.chunks[ci];
            const uint8_t* data = c.d_data;
            int buf = 0;
            const PageDev* pages_dev = c.d_pages;
            const void* table_dev = b->work[f].table.p;
            if (p.streamed) {
                // double buffer shared by all streamed files: the next chunk goes to the buffer
                // whose last scan is done
                buf = (int)(sbufs.seq++ & 1);
                if (sbufs.used[buf]) HIP_TRY(hipEventSynchronize(sbufs.scanned[buf]));
                // Whole chunk, or only the rows this batch looks up?  The table holds E entries per sub-index;
                // fetching them row by row moves E x (slices) x pitch bytes over PCIe at the rate random rows
                // come in, copying the chunk moves all of its rows at the slab rate (row_fetch_alpha prices
                // the difference).  The reference's mmap / AIO back-ends always take the first form
                // (compact_index/mmap_search_file.cpp:34-67, aio_search_file.cpp:58-97).
                const uint64_t E = (b->work[f].h_blk_off[nq] + nq) * 8ull * p.meta.num_hashes;
                const uint64_t gathered = (E * c.vp.size() + 1) * (uint64_t)c.pitch;
                const bool fetch = ix->tune.row_fetch != 0 && p.file_dev && c.d_src && !p.synthetic &&
                                   gathered <= sbufs.sbuf[buf].cap && E * c.vp.size() < 0xFFFFFFF0ull &&
                                   (gathered - c.pitch) * ix->tune.row_fetch_alpha <= c.bytes;
                if (fetch) {
                    if (!fetch_ready) {          // the fetch kernel reads K1's table: once per file and pass
                        HIP_TRY(hipEventRecord(sbufs.hashed, st));
                        HIP_TRY(hipStreamWaitEvent(sbufs.copy_stream, sbufs.hashed, 0));
                        fetch_ready = true;
                    }
                    FetchArgs fa;
                    fa.file = p.file_dev;
                    fa.table = b->work[f].table.p;
                    fa.table2 = sbufs.table2[buf].p;
                    fa.blk_off = b->work[f].blk_off;
                    fa.pages = c.d_pages;
                    fa.pages2 = c.d_pages2[buf];
                    fa.page_src = c.d_src;
                    fa.dst = sbufs.sbuf[buf].p;
                    fa.entries = E;
                    fa.src_pitch = p.meta.page_row_bytes();
                    fa.nq = (uint32_t)nq;
                    fa.npages = (uint32_t)c.vp.size();
                    fa.table_npages = p.num_tpages();
                    fa.num_hashes = (uint32_t)p.meta.num_hashes;
                    fa.pitch = c.pitch;
                    fa.ncols = (uint32_t)c.vp[0].ncols;
                    HIP_TRY(launch_fetch_rows(fa, p.idx64, sbufs.copy_stream));
                    pages_dev = c.d_pages2[buf];
                    table_dev = sbufs.table2[buf].p;
                    ++sbufs.fetched_chunks;
                } else {
                    cobs_gpu_status cs = stream_chunk_in(ix, p, c, buf);
                    if (cs != COBS_GPU_OK) return cs;
                    ++sbufs.streamed_chunks;
                }
                HIP_TRY(hipEventRecord(sbufs.copied[buf], sbufs.copy_stream));
                HIP_TRY(hipStreamWaitEvent(st, sbufs.copied[buf], 0));
                data = sbufs.sbuf[buf].p;
            }
            ScanArgs sa;
            sa.blob = data;
            sa.pages = pages_dev;
            sa.table = table_dev;
            sa.blk_off = b->work[f].blk_off;
            sa.counts = b->counts.p;
            sa.thresholds = b->selected ? b->work[f].thr.p : nullptr;
            sa.hits = b->hits.p;
            sa.hit_count = reinterpret_cast<unsigned long long*>(b->flags.p + 2);
            sa.counts_stride = ix->local_counts;
            sa.counts_offset = p.local_offset;
            sa.hit_cap = b->hit_cap;
            sa.nq = (uint32_t)nq;
            sa.npages = (uint32_t)c.vp.size();
            sa.table_npages = p.num_tpages();
            sa.pitch = c.pitch;
            sa.cpp = c.cpp;
            sa.total_chunks = c.total_chunks;
            sa.num_hashes = (uint32_t)p.meta.num_hashes;
            sa.num_docs = (uint32_t)p.meta.doc_names.size();
            sa.part = (uint32_t)f;
            sa.write_counts = b->have_counts ? 1 : 0;
            sa.idx64 = p.idx64 ? 1u : 0u;
            const ScanGeom geom = geoms[f][ci];
            const int nwaves = geom.nwaves;
            sa.tile_w = geom.tile_w;
            sa.cand = b->topk_direct ? b->cand.p + cand_off[f] : nullptr;
            sa.topk_k = b->topk_direct ? (uint32_t)topk : 0u;
            sa.cand_stride = cand_stride[f];
            sa.tile_base = tile_base;
            // K2 filters by threshold only when it selects: into the hit pool, or the tile's k best
            if (b->topk_direct) sa.thresholds = need_thr ? b->work[f].thr.p : nullptr;
            sa.dbg = nullptr;
            sa.dbg_every = 1;
            sa.dbg_slots = 0;
            if (ix->tune.phase_slots) {          // tuning builds: phase stamps of sampled work-groups (last launch wins)
                HIP_TRY(b->phase.reserve((size_t)ix->tune.phase_slots * 32));
                HIP_TRY(hipMemsetAsync(b->phase.p, 0, (size_t)ix->tune.phase_slots * 32 * 8, st));
                sa.dbg = b->phase.p;
                sa.dbg_slots = ix->tune.phase_slots;
            }
            // measured variant (A/B only): rows staged through LDS, where that kernel exists
            sa.lds_staged = ix->tune.lds_staged && !geom.multi_query && !p.idx64 &&
                            scan_has_lds_staged(b->planes, (uint32_t)p.meta.num_hashes, nwaves) ? 1u : 0u;
            sa.chunk_begin = 0;
            sa.chunk_end = c.total_chunks;
            // one launch covers at most 2^31-1 work-groups
            const uint32_t ntiles = (c.total_chunks + sa.tile_w - 1) / sa.tile_w;
            if ((uint64_t)ntiles * nq > 0x7FFFFFFFull)
                return fail(COBS_GPU_ERR_CAPACITY, "batch too large for one scan launch; use fewer queries");
            if (sa.dbg) {
                const uint64_t groups = geom.multi_query ? (uint64_t)ntiles * ((nq + 64 / sa.tile_w - 1) / (64 / sa.tile_w))
                                                          : (uint64_t)ntiles * nq;
                sa.dbg_every = (uint32_t)std::max<uint64_t>(1, groups / sa.dbg_slots);
            }
            HIP_TRY(launch_scan(sa, ntiles, b->planes, nwaves, geom.multi_query, st));
            tile_base += ntiles;
            ++launches;
            if (p.streamed) {
                HIP_TRY(hipEventRecord(sbufs.scanned[buf], st));
                sbufs.used[buf] = true;
            }
        }
    }
    if (!hash_marked) HIP_TRY(hipEventRecord(ev[1], st));
    HIP_TRY(hipEventRecord(ev[2], st));
    if (use_topk && nq) {
        for (size_t f = 0; f < ix->parts.size(); ++f) {
            const Part& p = ix->parts[f];
            TopkArgs ta;
            ta.counts = b->counts.p;
            ta.score_bytes = b->elem_bytes;
            ta.thresholds = need_thr ? b->work[f].thr.p : nullptr;
            ta.from_pool = 0;
            ta.out = b->topk_out.p + (uint64_t)f * nq * topk;
            ta.out_count = b->topk_cnt.p + f * nq;
            ta.counts_stride = ix->local_counts;
            ta.counts_offset = p.local_offset;
            ta.nslots = (uint32_t)p.slot_count;
            ta.doc_base = (uint32_t)p.slot_begin;
            ta.num_docs = (uint32_t)p.meta.doc_names.size();
            ta.k = (uint32_t)topk;
            ta.nq = (uint32_t)nq;
            ta.score_bits = (uint32_t)b->planes;
            ta.levels = ((uint32_t)b->planes + 11u) / 12u;                       // radix levels of <= 12 bits
            ta.level_bits = ((uint32_t)b->planes + ta.levels - 1u) / ta.levels;
            ta.sort_limit = topk <= kTopkSortLimit ? (uint32_t)topk : 0u;        // survivors ordered on the device
            if (b->topk_direct) {           // merge the tiles' candidates (threshold already applied by K2)
                ta.from_pool = 1;
                ta.counts = b->cand.p + cand_off[f];
                ta.counts_stride = cand_stride[f];
                ta.counts_offset = 0;
                ta.nslots = cand_tiles[f] * (uint32_t)topk;
                ta.thresholds = nullptr;
            }
            HIP_TRY(launch_topk(ta, st));
        }
    }
    HIP_TRY(hipEventRecord(b->run_done, st));
    b->run_seq++;
    b->stats[1] = launches;
    // T * H * (row bytes gathered) + score bytes WRITTEN (a hits-only pass writes none)
    b->stats[0] = b->algo_row_bytes + (b->have_counts ? (uint64_t)nq * ix->local_counts * b->elem_bytes : 0);
    b->ran = true;
    return COBS_GPU_OK;
}

extern "C" {

cobs_gpu_status cobs_gpu_batch_run(cobs_gpu_batch* b, double threshold, void* hip_stream) {
    return guarded([&]() { return run_impl(b, threshold, 0, hip_stream); });
}

cobs_gpu_status cobs_gpu_batch_run_hits(cobs_gpu_batch* b, double threshold, void* hip_stream) {
    if (!(threshold > 0.0)) return fail(COBS_GPU_ERR_ARG, "a hits-only pass needs a threshold > 0");
    return guarded([&]() { return run_impl(b, threshold, 0, hip_stream, false); });
}

cobs_gpu_status cobs_gpu_batch_run_topk(cobs_gpu_batch* b, double threshold, size_t num_results,
                                        void* hip_stream) {
    return guarded([&]() { return run_impl(b, threshold, num_results, hip_stream); });
}

cobs_gpu_status cobs_gpu_batch_run_topk_only(cobs_gpu_batch* b, double threshold, size_t num_results,
                                             void* hip_stream) {
    if (num_results == 0) return fail(COBS_GPU_ERR_ARG, "a top-k pass needs num_results > 0");
    return guarded([&]() { return run_impl(b, threshold, num_results, hip_stream, false); });
}

cobs_gpu_status cobs_gpu_batch_sync(cobs_gpu_batch* b, void* hip_stream, size_t* bad_query) {
    if (!b) return fail(COBS_GPU_ERR_ARG, "NULL batch");
    if (!b->ran) return fail(COBS_GPU_ERR_ARG, "batch has not been run");
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(b->ix->device));
    if (b->graph_run && b->h_res.p) {            // the graph already copied the flags (and the results) home
        HIP_TRY(hipStreamSynchronize(st));
        std::memcpy(b->h_flags, b->h_res.p, sizeof b->h_flags);
    } else {
        HIP_TRY(hipMemcpyAsync(b->h_flags, b->flags.p, sizeof b->h_flags, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    b->synced = true;
    if (b->h_flags[0] != 0u) {           // K1 keeps 2^32-1 - (first query with a non-ACGT character)
        const uint32_t bad = 0xFFFFFFFFu - b->h_flags[0];
        if (bad_query) *bad_query = bad;
        return fail(COBS_GPU_ERR_INVALID_BASE,
                    "Invalid DNA base pair in query string. Only ACGT are allowed. (query " +
                    std::to_string(bad) + ")");
    }
    return COBS_GPU_OK;
}

void* cobs_gpu_batch_counts_device(cobs_gpu_batch* b, uint32_t* elem_bytes, uint64_t* row_stride_bytes) {
    if (!b) return nullptr;
    if (elem_bytes) *elem_bytes = b->elem_bytes;
    if (row_stride_bytes) *row_stride_bytes = b->ix->local_counts * b->elem_bytes;
    // the rows are allocated lazily (see run_impl); a caller that asks for them before the first
    // run (to size an exchange buffer, say) gets them now
    if (hipSetDevice(b->ix->device) != hipSuccess ||
        b->counts.reserve((size_t)(b->nq * b->ix->local_counts * b->elem_bytes)) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return b->counts.p;
}

// Raw local score row of query q of the last run, through a pinned host window of up to 64 MiB
// of consecutive rows (callers walk the queries in order: one DMA per window, not per query).
// After an exchange (comm.cpp) the batch may expose GLOBAL rows instead: queries
// [g_q0, g_q0 + g_qn), every row total_counts elements in global document order.
static cobs_gpu_status fetch_row(cobs_gpu_batch* b, size_t q, const uint8_t** row) {
    const bool glob = b->view_global;
    const size_t row_bytes = (size_t)((glob ? b->ix->total_counts : b->ix->local_counts) * b->elem_bytes);
    if (glob && (q < b->g_q0 || q >= b->g_q0 + b->g_qn))
        return fail(COBS_GPU_ERR_ARG, "this rank does not hold the exchanged row of that query");
    if (!glob && b->graph_run && b->res_rows && b->rows_q1 == 0) {
        b->rows_q0 = 0;                 // the replayed graph copied all rows of this small pass into the window
        b->rows_q1 = b->nq;
    }
    if (q < b->rows_q0 || q >= b->rows_q1) {
        const size_t per = std::max<size_t>(1, (64u << 20) / std::max<size_t>(row_bytes, 1));
        const size_t q1 = std::min(glob ? (size_t)(b->g_q0 + b->g_qn) : b->nq, q + per);
        HIP_TRY(b->h_rows.reserve(std::max<size_t>((q1 - q) * row_bytes, 1)));
        const uint8_t* src = glob ? b->g_rows + (q - b->g_q0) * row_bytes : b->counts.p + q * row_bytes;
        if (row_bytes)
            HIP_TRY(hipMemcpy(b->h_rows.p, src, (q1 - q) * row_bytes, hipMemcpyDeviceToHost));
        b->rows_q0 = q;
        b->rows_q1 = q1;
    }
    *row = b->h_rows.p + (q - b->rows_q0) * row_bytes;
    return COBS_GPU_OK;
}

static inline uint32_t score_at(const uint8_t* row, uint32_t elem_bytes, uint64_t i) {
    if (elem_bytes == 1) return row[i];
    if (elem_bytes == 2) return reinterpret_cast<const uint16_t*>(row)[i];
    return reinterpret_cast<const uint32_t*>(row)[i];
}

// local count row of query q, widened to u32, scattered into a global-layout vector
static cobs_gpu_status fetch_counts(cobs_gpu_batch* b, size_t q, uint32_t* counts) {
    cobs_gpu_index* ix = b->ix;
    if (!b->have_counts) return fail(COBS_GPU_ERR_ARG, "the last run did not keep the score rows");
    const uint8_t* raw = nullptr;
    cobs_gpu_status st = fetch_row(b, q, &raw);
    if (st != COBS_GPU_OK) return st;
    if (b->view_global) {
        for (uint64_t i = 0; i < ix->total_counts; ++i) counts[i] = score_at(raw, b->elem_bytes, i);
        return COBS_GPU_OK;
    }
    std::fill(counts, counts + ix->total_counts, 0u);
    for (const Part& p : ix->parts) {
        uint32_t* dst = counts + p.doc_offset + p.slot_begin;
        if (b->elem_bytes == 1) {
            const uint8_t* s = raw + p.local_offset;
            for (uint64_t i = 0; i < p.slot_count; ++i) dst[i] = s[i];
        } else if (b->elem_bytes == 2) {
            const uint16_t* s = reinterpret_cast<const uint16_t*>(raw) + p.local_offset;
            for (uint64_t i = 0; i < p.slot_count; ++i) dst[i] = s[i];
        } else {
            const uint32_t* s = reinterpret_cast<const uint32_t*>(raw) + p.local_offset;
            for (uint64_t i = 0; i < p.slot_count; ++i) dst[i] = s[i];
        }
    }
    return COBS_GPU_OK;
}

// counts_to_result over a whole score row (threshold <= 0: every document is a result):
// the result order (score desc, then (file, doc) asc; classic_search.cpp:134-145, :179-188) is a
// stable counting sort by score of the documents taken in (file, doc) order -- O(documents),
// where std::partial_sort of 100 000 documents costs ~9 ms per query.  Writes the first `want`
// results straight into `hits` (when it is large enough) and returns their number.
// (rank_raw touches nothing of the batch but `hist`: several host threads rank different queries of
// one row window at the same time, see rank_window)
static cobs_gpu_status rank_raw(const cobs_gpu_batch* b, size_t q, const uint8_t* raw, std::vector<uint32_t>& hist,
                                size_t num_results, cobs_gpu_hit* hits, size_t cap, size_t* n_hits) {
    const cobs_gpu_index* ix = b->ix;
    const uint32_t eb = b->elem_bytes;
    const bool glob = b->view_global;
    const bool by_score = total_hashes(b, q) > 1;       // max_counts <= 1: index order, no sort (:134, :177)
    // pass 1: passing documents per score
    uint64_t max_score = 0;
    for (const Part& p : ix->parts)
        max_score = std::max<uint64_t>(max_score, (uint64_t)b->lens[q] - p.meta.term_size + 1);
    if (max_score > (1u << 24)) return COBS_GPU_ERR_UNSUPPORTED;      // caller falls back to the generic sort
    hist.assign((size_t)max_score + 2, 0u);
    size_t passing = 0;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        const Part& p = ix->parts[f];
        const uint32_t thr = threshold_for(b->threshold, (uint64_t)b->lens[q] - p.meta.term_size + 1);
        const uint64_t d0 = glob ? 0 : p.slot_begin;
        const uint64_t d1 = glob ? p.meta.doc_names.size()
                                 : std::min<uint64_t>(p.slot_begin + p.slot_count, p.meta.doc_names.size());
        const uint64_t base = glob ? p.doc_offset : p.local_offset;
        for (uint64_t d = d0; d < d1; ++d) {
            const uint32_t s = score_at(raw, eb, base + d - d0);
            if (s >= thr) { ++hist[by_score ? std::min<uint64_t>(s, max_score) : 0]; ++passing; }
        }
    }
    size_t want = num_results == 0 ? (size_t)ix->total_counts : std::min<size_t>(num_results, (size_t)ix->total_counts);
    want = std::min(want, passing);
    *n_hits = want;
    if (want > cap) return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small");
    if (want && !hits) return fail(COBS_GPU_ERR_ARG, "NULL hit buffer");
    // start position of every score, highest first
    uint32_t pos = 0;
    for (size_t s = hist.size(); s-- > 0;) {
        const uint32_t c = hist[s];
        hist[s] = pos;
        pos += c;
    }
    // pass 2: scatter in (file, doc) order; positions >= want are dropped
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        const Part& p = ix->parts[f];
        const uint32_t thr = threshold_for(b->threshold, (uint64_t)b->lens[q] - p.meta.term_size + 1);
        const uint64_t d0 = glob ? 0 : p.slot_begin;
        const uint64_t d1 = glob ? p.meta.doc_names.size()
                                 : std::min<uint64_t>(p.slot_begin + p.slot_count, p.meta.doc_names.size());
        const uint64_t base = glob ? p.doc_offset : p.local_offset;
        for (uint64_t d = d0; d < d1; ++d) {
            const uint32_t s = score_at(raw, eb, base + d - d0);
            if (s < thr) continue;
            const uint32_t at = hist[by_score ? std::min<uint64_t>(s, max_score) : 0]++;
            if (at < want) hits[at] = cobs_gpu_hit{(uint32_t)f, (uint32_t)d, s};
        }
    }
    return COBS_GPU_OK;
}

static cobs_gpu_status rank_row(cobs_gpu_batch* b, size_t q, size_t num_results, cobs_gpu_hit* hits, size_t cap,
                                size_t* n_hits) {
    if (!b->have_counts) return fail(COBS_GPU_ERR_ARG, "the last run did not keep the score rows");
    const uint8_t* raw = nullptr;
    cobs_gpu_status st = fetch_row(b, q, &raw);
    if (st != COBS_GPU_OK) return st;
    return rank_raw(b, q, raw, b->rank_hist, num_results, hits, cap, n_hits);
}

// The reference's default call (threshold 0, no limit) ranks EVERY document of every query: with
// thousands of queries per pass that is host work worth spreading.  Queries [q0, q1) of the last run,
// every one yielding exactly `per_query` hits (threshold <= 0: all real documents pass), written to
// hits + (q - q0) * per_query by up to 16 host threads, one row window (one DMA) at a time.
static cobs_gpu_status rank_window(cobs_gpu_batch* b, size_t q0, size_t q1, size_t per_query, cobs_gpu_hit* hits) {
    const size_t row_bytes = (size_t)(b->ix->local_counts * b->elem_bytes);
    for (size_t q = q0; q < q1;) {
        const uint8_t* raw0 = nullptr;
        cobs_gpu_status st = fetch_row(b, q, &raw0);               // loads the window that starts at q
        if (st != COBS_GPU_OK) return st;
        const size_t qe = std::min(q1, b->rows_q1);
        const unsigned nthr = (unsigned)std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), qe - q);
        std::vector<cobs_gpu_status> res(nthr, COBS_GPU_OK);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthr; ++t)
            pool.emplace_back([=, &res]() {
                std::vector<uint32_t> hist;
                for (size_t i = q + t; i < qe; i += nthr) {
                    size_t n = 0;
                    const cobs_gpu_status r = rank_raw(b, i, raw0 + (i - q) * row_bytes, hist, 0, hits + (i - q0) * per_query,
                                                       per_query, &n);
                    if (r != COBS_GPU_OK || n != per_query) { res[t] = r != COBS_GPU_OK ? r : COBS_GPU_ERR_ARG; return; }
                }
            });
        for (auto& th : pool) th.join();
        for (cobs_gpu_status r : res)
            if (r != COBS_GPU_OK) return r == COBS_GPU_ERR_UNSUPPORTED ? r : fail(r, "ranking a row window failed");
        q = qe;
    }
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_batch_counts_host(cobs_gpu_batch* b, size_t q, uint32_t* counts, size_t cap) {
    if (!b || !counts) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (!b->ran || !b->synced) return fail(COBS_GPU_ERR_ARG, "run and sync the batch first");
    if (q >= b->nq) return fail(COBS_GPU_ERR_ARG, "query number out of range");
    if (cap < b->ix->total_counts) return fail(COBS_GPU_ERR_CAPACITY, "counts buffer too small");
    HIP_TRY(hipSetDevice(b->ix->device));
    return fetch_counts(b, q, counts);
}

static cobs_gpu_status hits_host_impl(cobs_gpu_batch* b, size_t q, size_t num_results,
                                      cobs_gpu_hit* hits, size_t cap, size_t* n_hits);

cobs_gpu_status cobs_gpu_batch_hits_host(cobs_gpu_batch* b, size_t q, size_t num_results,
                                         cobs_gpu_hit* hits, size_t cap, size_t* n_hits) {
    return guarded([&]() { return hits_host_impl(b, q, num_results, hits, cap, n_hits); });
}

static cobs_gpu_status hits_host_impl(cobs_gpu_batch* b, size_t q, size_t num_results,
                                      cobs_gpu_hit* hits, size_t cap, size_t* n_hits) {
    if (!b || !n_hits) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (!b->ran || !b->synced) return fail(COBS_GPU_ERR_ARG, "run and sync the batch first");
    if (q >= b->nq) return fail(COBS_GPU_ERR_ARG, "query number out of range");
    cobs_gpu_index* ix = b->ix;
    HIP_TRY(hipSetDevice(ix->device));
    std::vector<cobs_gpu_hit>& sel = b->sel_scratch;     // reused: no allocation per query
    sel.clear();
    const bool pool_ok = b->selected && (b->pool_global || b->h_nhits() <= b->hit_cap);
    const bool topk_ok = b->topk_k > 0 && num_results > 0 && num_results <= b->topk_k && total_hashes(b, q) > 1;
    if (topk_ok) {
        // K3 left the k best documents of every file on the device: fetch once, merge per query
        const size_t k = b->topk_k, nparts = ix->parts.size();
        if (!b->topk_fetched) {
            b->h_topk.resize(k * b->nq * nparts);
            b->h_topk_cnt.resize(b->nq * nparts);
            if (b->graph_run && b->h_res.p) {
                std::memcpy(b->h_topk_cnt.data(), b->h_res.p + 16, 4 * b->h_topk_cnt.size());
                std::memcpy(b->h_topk.data(), b->h_res.p + b->res_topk, sizeof(uint2) * b->h_topk.size());
            } else {
                HIP_TRY(hipMemcpy(b->h_topk.data(), b->topk_out.p, sizeof(uint2) * b->h_topk.size(), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(b->h_topk_cnt.data(), b->topk_cnt.p, 4 * b->h_topk_cnt.size(), hipMemcpyDeviceToHost));
            }
            b->topk_fetched = true;
        }
        const size_t stride = b->topk_stride ? b->topk_stride : k;     // ranks * k after an exchange
        if (nparts == 1 && b->topk_sorted && !b->topk_stride) {
            // one file, one shard: K3 already left the survivors in result order
            const uint2* e = b->h_topk.data() + q * k;
            const size_t want1 = std::min<size_t>(std::min<size_t>(num_results, (size_t)ix->total_counts), b->h_topk_cnt[q]);
            *n_hits = want1;
            if (want1 > cap) return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small");
            if (want1 && !hits) return fail(COBS_GPU_ERR_ARG, "NULL hit buffer");
            for (size_t i = 0; i < want1; ++i) hits[i] = cobs_gpu_hit{0u, e[i].x, e[i].y};
            return COBS_GPU_OK;
        }
        for (size_t f = 0; f < nparts; ++f) {
            const uint2* e = b->h_topk.data() + (f * b->nq + q) * stride;
            const uint32_t cnt = b->h_topk_cnt[f * b->nq + q];
            for (uint32_t i = 0; i < cnt; ++i) sel.push_back(cobs_gpu_hit{(uint32_t)f, e[i].x, e[i].y});
        }
    } else if (pool_ok) {
        if (!b->pool_fetched) {
            // the pool arrives in arbitrary order: bucket it by query with a counting scatter
            std::vector<HitDev> raw((size_t)b->h_nhits());
            if (!raw.empty()) {
                if (b->graph_run && b->h_res.p && raw.size() <= b->res_pool_n)
                    std::memcpy(raw.data(), b->h_res.p + b->res_pool, sizeof(HitDev) * raw.size());
                else
                    HIP_TRY(hipMemcpy(raw.data(), b->hits.p, sizeof(HitDev) * raw.size(), hipMemcpyDeviceToHost));
            }
            b->h_hit_off.assign(b->nq + 1, 0);
            for (const HitDev& h : raw) b->h_hit_off[h.query + 1]++;
            for (size_t i = 0; i < b->nq; ++i) b->h_hit_off[i + 1] += b->h_hit_off[i];
            b->h_hits.resize(raw.size());
            std::vector<size_t> cur(b->h_hit_off.begin(), b->h_hit_off.end() - 1);
            for (const HitDev& h : raw) b->h_hits[cur[h.query]++] = h;
            b->pool_fetched = true;
        }
        for (size_t i = b->h_hit_off[q]; i < b->h_hit_off[q + 1]; ++i)
            sel.push_back(cobs_gpu_hit{b->h_hits[i].part, b->h_hits[i].doc, b->h_hits[i].score});
    } else {
        // threshold <= 0 (every document is a hit) or pool overflow: rank the score row on the host
        cobs_gpu_status rs = rank_row(b, q, num_results, hits, cap, n_hits);
        if (rs != COBS_GPU_ERR_UNSUPPORTED) return rs;
        // scores too wide for a counting sort: generic path
        std::vector<uint32_t> counts((size_t)ix->total_counts);
        cobs_gpu_status st = fetch_counts(b, q, counts.data());
        if (st != COBS_GPU_OK) return st;
        for (size_t f = 0; f < ix->parts.size(); ++f) {
            const Part& p = ix->parts[f];
            const uint32_t thr = threshold_for(b->threshold, (uint64_t)b->lens[q] - p.meta.term_size + 1);
            // only documents whose slots this shard computed
            const uint64_t d0 = b->view_global ? 0 : p.slot_begin;
            const uint64_t d1 = b->view_global ? p.meta.doc_names.size()
                                               : std::min<uint64_t>(p.slot_begin + p.slot_count, p.meta.doc_names.size());
            for (uint64_t d = d0; d < d1; ++d) {
                const uint32_t s = counts[p.doc_offset + d];
                if (s >= thr) sel.push_back(cobs_gpu_hit{(uint32_t)f, (uint32_t)d, s});
            }
        }
    }
    // classic_search.cpp:450-451,134-145
    size_t want = num_results == 0 ? (size_t)ix->total_counts : std::min<size_t>(num_results, (size_t)ix->total_counts);
    want = std::min(want, sel.size());
    if (total_hashes(b, q) > 1)
        std::partial_sort(sel.begin(), sel.begin() + want, sel.end(), hit_before);
    else
        std::partial_sort(sel.begin(), sel.begin() + want, sel.end(), doc_before);
    *n_hits = want;
    if (want > cap) return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small");
    if (want && !hits) return fail(COBS_GPU_ERR_ARG, "NULL hit buffer");
    std::copy(sel.begin(), sel.begin() + want, hits);
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_batch_phase_stamps(cobs_gpu_batch* b, uint64_t* out, size_t cap_words, size_t* n_words) {
    if (!b || !n_words) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    const size_t n = (size_t)b->ix->tune.phase_slots * 32;
    *n_words = n;
    if (!b->phase.p || n == 0) { *n_words = 0; return COBS_GPU_OK; }
    if (cap_words < n || !out) return fail(COBS_GPU_ERR_CAPACITY, "stamp buffer too small");
    HIP_TRY(hipSetDevice(b->ix->device));
    HIP_TRY(hipMemcpy(out, b->phase.p, n * 8, hipMemcpyDeviceToHost));
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_batch_stats(const cobs_gpu_batch* b, uint64_t out[4]) {
    if (!b || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    std::memcpy(out, b->stats, sizeof b->stats);
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_batch_kernel_ms(cobs_gpu_batch* b, float* scan_ms, float* hash_ms) {
    if (!b) return fail(COBS_GPU_ERR_ARG, "NULL batch");
    if (!b->ran || !b->synced) return fail(COBS_GPU_ERR_ARG, "run and sync the batch first");
    // average over the runs since the previous call (at most the last kRing runs)
    uint64_t first = b->read_seq;
    if (b->run_seq - first > (uint64_t)cobs_gpu_batch::kRing) first = b->run_seq - cobs_gpu_batch::kRing;
    if (first == b->run_seq) first = b->run_seq - 1;       // nothing new: report the last run again
    double h = 0, s = 0;
    for (uint64_t r = first; r < b->run_seq; ++r) {
        hipEvent_t* ev = b->ev[r % cobs_gpu_batch::kRing];
        float a = 0, c = 0;
        HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&c, ev[1], ev[2]));
        h += a;
        s += c;
    }
    const double n = (double)(b->run_seq - first);
    b->read_seq = b->run_seq;
    if (hash_ms) *hash_ms = (float)(h / n);
    if (scan_ms) *scan_ms = (float)(s / n);
    return COBS_GPU_OK;
}

// ---------------------------------------------------------------------------
// host-buffer search API

// One pass of the host-buffer API on scratch batch `slot`, in two halves so that passes can
// overlap: begin = stage the queries, upload them and launch K1/K2(/K3) on the slot's own stream
// (asynchronous; the kernels are ordered after `after`, the previous pass), end = wait for it,
// repeat it with score rows if the hit pool overflowed, book the timers.
static cobs_gpu_status host_pass_begin(cobs_gpu_index* ix, int slot, const char* const* queries, const size_t* lens,
                                       size_t nq, double threshold, size_t topk, hipEvent_t after,
                                       size_t* bad_at = nullptr, size_t index_base = 0) {
    HIP_TRY(hipSetDevice(ix->device));
    if (!ix->scratch[slot]) {
        cobs_gpu_status st = cobs_gpu_batch_create(ix, 0, 0, &ix->scratch[slot]);
        if (st != COBS_GPU_OK) return st;
        HIP_TRY(hipStreamCreateWithFlags(&ix->scratch[slot]->own_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ix->scratch[slot]->done, hipEventDisableTiming));
    }
    cobs_gpu_batch* b = ix->scratch[slot];
    double t0 = now_s();
    size_t bad_local = 0;
    cobs_gpu_status st = set_queries_on(b, queries, lens, nq, b->own_stream, false, &bad_local, index_base);
    if (st != COBS_GPU_OK && bad_at) *bad_at = bad_local;
    if (st != COBS_GPU_OK) return st;
    ix->timers[1] += now_s() - t0;
    if (after) HIP_TRY(hipStreamWaitEvent(b->own_stream, after, 0));
    // with a threshold and no limit only the selected hits travel back: skip the score rows,
    // unless the hit pool overflows (then the pass is repeated with them and ranked on the host)
    // ... and with a limit K2 / K3 select on the device (tile-level top-k where it applies): no score rows either
    const bool hits_only = (threshold > 0.0 && topk == 0) || topk > 0;
    // Small calls (a single query is the reference's own entry point, search.hpp:39-42) are
    // launch-bound: fill + K1 + K2 (+ K3) are four launches for ~15 us of work.  The second time
    // the same shape comes along (same query lengths, parameters and buffers) the pass is captured
    // into a hipGraph and from then on replayed with one launch.
    bool any_streamed = false;
    for (const auto& p : ix->parts) any_streamed = any_streamed || p.streamed;
    if (nq > 0 && nq <= 16 && ix->tune.graph != 0 && !any_streamed && !ix->tune.phase_slots) {
        // the shape of the pass and every address the captured nodes hold
        auto make_key = [&]() {
            uint64_t key = 1469598103934665603ull;
            auto mixin = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
            mixin(nq);
            for (size_t q = 0; q < nq; ++q) mixin(lens[q]);
            uint64_t tb;
            std::memcpy(&tb, &threshold, 8);
            mixin(tb); mixin(topk); mixin(hits_only);
            mixin((uint64_t)(uintptr_t)b->text.p); mixin((uint64_t)(uintptr_t)b->counts.p); mixin((uint64_t)(uintptr_t)b->hits.p);
            mixin((uint64_t)(uintptr_t)b->topk_out.p); mixin((uint64_t)(uintptr_t)b->topk_cnt.p); mixin((uint64_t)(uintptr_t)b->cand.p);
            for (auto& w : b->work) { mixin((uint64_t)(uintptr_t)w.table.p); mixin((uint64_t)(uintptr_t)w.thr.p); }
            mixin((uint64_t)(uintptr_t)b->h_res.p); mixin((uint64_t)(uintptr_t)b->h_rows.p);      // the graph writes there
            mixin((uint64_t)(uintptr_t)b->h_text.p); mixin((uint64_t)(uintptr_t)b->h_thr_stage.p);
            mixin(ix->tune.waves); mixin(ix->tune.tile_w); mixin((uint64_t)(int64_t)ix->tune.mq); mixin(ix->tune.lds_staged);
            mixin((uint64_t)ix->tune.tile_topk);      // decides which buffers the pass needs (a capture must not allocate)
            return key;
        };
        const uint64_t key = make_key();
        if (!(b->graph_exec && b->graph_key == key)) {
            // captured earlier, displaced by other shapes since?  make it the current one again
            for (auto& e : b->graph_more) {
                if (!e.exec || e.key != key) continue;
                std::swap(e.exec, b->graph_exec);
                std::swap(e.key, b->graph_key);
                std::swap(e.res_topk, b->res_topk);
                std::swap(e.res_pool, b->res_pool);
                std::swap(e.res_pool_n, b->res_pool_n);
                std::swap(e.res_rows, b->res_rows);
                e.used = ++b->graph_clock;
                break;
            }
        }
        if (b->graph_exec && b->graph_key == key) {
            set_run_state(b, threshold, topk, !hits_only);
            if (threshold > 0.0) stage_thresholds(b, threshold);     // the graph's H2D nodes read them now
            HIP_TRY(hipGraphLaunch(b->graph_exec, b->own_stream));
            b->graph_run = true;
            b->run_seq++;
            b->ran = true;
            ix->graph_replays++;
            HIP_TRY(hipEventRecord(b->done, b->own_stream));
            return COBS_GPU_OK;
        }
        bool seen_before = b->graph_candidate == key;
        for (uint64_t k : b->graph_recent) seen_before = seen_before || (k != 0 && k == key);
        if (seen_before) {
            // same shape twice in a row: every buffer already has its size (no allocation inside the capture)
            hipGraph_t graph = nullptr;
            // the results travel back inside the graph too (pinned buffers sized before the capture):
            // flags | top-k counts and survivors | a prefix of the hit pool; score rows of an
            // all-documents call go to the row window
            const size_t np = ix->parts.size();
            const bool will_topk = topk > 0 && topk <= 65536 && (uint64_t)topk * nq * np <= (1ull << 27);
            const bool will_select = threshold > 0.0 && !will_topk;
            const size_t res_topk_cnt = 16, res_topk = res_topk_cnt + (will_topk ? 4 * np * nq : 0);
            const size_t res_pool = (res_topk + (will_topk ? 8 * np * nq * topk : 0) + 15) / 16 * 16;
            const size_t pool_n = will_select ? std::min<size_t>(b->hit_cap, kGraphPoolPrefix) : 0;
            const size_t row_bytes_all = (!will_topk && !will_select) ? (size_t)(nq * ix->local_counts * b->elem_bytes) : 0;
            bool pre_ok = b->h_res.reserve(res_pool + pool_n * sizeof(HitDev) + 16) == hipSuccess;
            if (row_bytes_all) pre_ok = pre_ok && row_bytes_all <= (64u << 20) && b->h_rows.reserve(row_bytes_all) == hipSuccess;
            if (pre_ok && hipStreamBeginCapture(b->own_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                cobs_gpu_status cs = run_impl(b, threshold, topk, b->own_stream, !hits_only);
                if (cs == COBS_GPU_OK) {
                    hipError_t ce = hipMemcpyAsync(b->h_res.p, b->flags.p, 16, hipMemcpyDeviceToHost, b->own_stream);
                    if (ce == hipSuccess && b->topk_k) {
                        ce = hipMemcpyAsync(b->h_res.p + res_topk_cnt, b->topk_cnt.p, 4 * np * nq, hipMemcpyDeviceToHost, b->own_stream);
                        if (ce == hipSuccess)
                            ce = hipMemcpyAsync(b->h_res.p + res_topk, b->topk_out.p, 8 * np * nq * topk, hipMemcpyDeviceToHost, b->own_stream);
                    }
                    if (ce == hipSuccess && pool_n)
                        ce = hipMemcpyAsync(b->h_res.p + res_pool, b->hits.p, pool_n * sizeof(HitDev), hipMemcpyDeviceToHost, b->own_stream);
                    if (ce == hipSuccess && row_bytes_all && b->have_counts)
                        ce = hipMemcpyAsync(b->h_rows.p, b->counts.p, row_bytes_all, hipMemcpyDeviceToHost, b->own_stream);
                    if (ce != hipSuccess) cs = COBS_GPU_ERR_HIP;
                }
                const hipError_t ee = hipStreamEndCapture(b->own_stream, &graph);
                hipGraphExec_t exec = nullptr;
                if (cs == COBS_GPU_OK && ee == hipSuccess && graph &&
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    if (b->graph_exec) {
                        // the current graph moves to the least recently used of the older slots
                        cobs_gpu_batch::GraphEntry* lru = &b->graph_more[0];
                        for (auto& e : b->graph_more)
                            if (!e.exec || (lru->exec && e.used < lru->used)) { lru = &e; if (!e.exec) break; }
                        if (lru->exec) (void)hipGraphExecDestroy(lru->exec);
                        lru->exec = b->graph_exec;
                        lru->key = b->graph_key;
                        lru->res_topk = b->res_topk;
                        lru->res_pool = b->res_pool;
                        lru->res_pool_n = b->res_pool_n;
                        lru->res_rows = b->res_rows;
                        lru->used = ++b->graph_clock;
                    }
                    b->graph_exec = exec;
                    b->graph_key = make_key();          // with the addresses as they are now
                    b->res_topk = res_topk;
                    b->res_pool = res_pool;
                    b->res_pool_n = pool_n;
                    b->res_rows = row_bytes_all != 0;
                    (void)hipGraphDestroy(graph);
                    HIP_TRY(hipGraphLaunch(b->graph_exec, b->own_stream));
                    b->graph_run = true;
                    HIP_TRY(hipEventRecord(b->done, b->own_stream));
                    return COBS_GPU_OK;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
            ix->tune.graph = 0;              // capture is not possible here: never try again on this handle
        }
        st = run_impl(b, threshold, topk, b->own_stream, !hits_only);
        if (st != COBS_GPU_OK) return st;
        b->graph_candidate = make_key();                // buffers have their sizes (and addresses) now
        b->graph_recent[b->graph_clock++ % 4] = b->graph_candidate;
        HIP_TRY(hipEventRecord(b->done, b->own_stream));
        return COBS_GPU_OK;
    }
    st = run_impl(b, threshold, topk, b->own_stream, !hits_only);
    if (st != COBS_GPU_OK) return st;
    HIP_TRY(hipEventRecord(b->done, b->own_stream));
    return COBS_GPU_OK;
}

static cobs_gpu_status host_pass_end(cobs_gpu_index* ix, int slot, double threshold, size_t topk, size_t* bad_query) {
    cobs_gpu_batch* b = ix->scratch[slot];
    cobs_gpu_status st = cobs_gpu_batch_sync(b, b->own_stream, bad_query);
    if (st == COBS_GPU_OK && !b->have_counts && b->h_nhits() > b->hit_cap) {
        st = run_impl(b, threshold, topk, b->own_stream, true);
        if (st != COBS_GPU_OK) return st;
        HIP_TRY(hipEventRecord(b->done, b->own_stream));
        st = cobs_gpu_batch_sync(b, b->own_stream, bad_query);
    }
    if (st == COBS_GPU_OK || st == COBS_GPU_ERR_INVALID_BASE) {
        float sm = 0, hm = 0;
        // (a replayed graph re-records the events of the run it was captured from: no per-kernel split)
        if (b->ran && !b->graph_run && cobs_gpu_batch_kernel_ms(b, &sm, &hm) == COBS_GPU_OK) {
            ix->timers[0] += hm * 1e-3;
            ix->timers[2] += sm * 1e-3;
        }
    }
    return st;
}

static cobs_gpu_status run_host_batch(cobs_gpu_index* ix, const char* const* queries, const size_t* lens,
                                      size_t nq, double threshold, size_t* bad_query, size_t topk = 0) {
    cobs_gpu_status st = host_pass_begin(ix, 0, queries, lens, nq, threshold, topk, nullptr);
    if (st != COBS_GPU_OK) return st;
    return host_pass_end(ix, 0, threshold, topk, bad_query);
}

static cobs_gpu_status search_batch_impl(cobs_gpu_index* ix, const char* const* queries, const size_t* lens,
                                         size_t nq, double threshold, size_t num_results,
                                         cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets,
                                         size_t* bad_query) {
    if (!ix || !hit_offsets) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (nq && (!queries || !lens)) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    size_t used = 0;
    hit_offsets[0] = 0;
    bool overflow = false;
    // Large batches are cut into device passes whose score rows and row-index tables stay
    // below a limit each (the caller sees one call; results are concatenated).
    // (16 GiB: a small part of 288 GB of HBM, and large passes keep more lookups per cached line.)
    const uint64_t kLimit = ix->tune.pass_bytes;
    uint32_t min_term = 0xFFFFFFFFu;
    for (const auto& p : ix->parts) min_term = std::min(min_term, p.meta.term_size);
    uint64_t terms_per_char = 0;                      // table bytes per query character, all files
    for (const auto& p : ix->parts) terms_per_char += 4ull * p.meta.num_hashes * std::max<uint32_t>(p.num_tpages(), 1) * (p.idx64 ? 2 : 1);
    // Passes are pipelined over up to three scratch batches: while the GPU scans pass i the host
    // stages and uploads pass i+1 and ranks pass i-1 (kernels of consecutive passes are chained by
    // events, so they never share the GPU).  A call with 4 MiB of query text or more is cut into at
    // least four passes for that.  Streamed (out-of-core) files share their chunk buffers: one pass at a time.
    bool any_streamed = false;
    for (const auto& p : ix->parts) any_streamed = any_streamed || p.streamed;
    const size_t depth = any_streamed ? 1 : (size_t)cobs_gpu_index::kScratch;
    uint64_t total_chars = 0;
    for (size_t q = 0; q < nq; ++q) total_chars += lens[q];
    const uint64_t pipe_chars = ix->tune.pipe_chars;   // 0 = never cut for pipelining
    const size_t max_pass = (!any_streamed && pipe_chars && total_chars >= pipe_chars && nq >= 64)
                                ? (nq + 3) / 4 : std::max<size_t>(nq, 1);
    const size_t topk = num_results < ix->total_counts ? num_results : 0;   // bounded: K3 selects on the device
    struct Pass { size_t g0, g1; int slot; };
    std::vector<Pass> inflight;                        // FIFO, at most `depth` entries
    auto drain = [&]() {                               // error paths: nothing may still use the scratch batches
        for (const Pass& ps : inflight) (void)hipStreamSynchronize(ix->scratch[ps.slot]->own_stream);
        inflight.clear();
    };
    auto collect = [&](const Pass& ps) -> cobs_gpu_status {
        size_t bad = 0;
        const double te0 = now_s();
        cobs_gpu_status st = host_pass_end(ix, ps.slot, threshold, topk, &bad);
        if (ix->tune.trace) std::fprintf(stderr, "[cobs_gpu] pass of queries %zu..%zu: waited %.3f ms for the device\n", ps.g0, ps.g1, (now_s() - te0) * 1e3);
        if (st != COBS_GPU_OK) {
            if (bad_query) *bad_query = ps.g0 + bad;
            if (st == COBS_GPU_ERR_INVALID_BASE)          // the message names the query by its index in the call
                return fail(st, "Invalid DNA base pair in query string. Only ACGT are allowed. (query " +
                                std::to_string(ps.g0 + bad) + ")");
            return st;
        }
        cobs_gpu_batch* sb = ix->scratch[ps.slot];
        // results that have to come from whole score rows -- the reference's default call (threshold 0, no
        // limit: every document of every query, src/cobs.cpp:618-626), a limit too large for K3, a hit pool
        // that overflowed -- are ordered on the device and cross PCIe as finished records (rank.cpp)
        if (ix->tune.device_rank != 0 && rank_on_device_applies(sb, ps.g1 - ps.g0)) {
            double t0 = now_s();
            st = rank_on_device(sb, ps.g1 - ps.g0, num_results, hits, cap, &used, hit_offsets + ps.g0, &overflow);
            ix->timers[4] += now_s() - t0;
            return st;
        }
        // all documents of every query (the reference's default call): every query yields the same
        // number of hits, so the queries of the pass are ranked by several host threads at once
        if (!overflow && threshold <= 0.0 && num_results == 0 && sb->have_counts && !sb->selected && sb->topk_k == 0 &&
            !sb->view_global && ps.g1 - ps.g0 >= 4 && sb->max_terms <= (1u << 24)) {
            size_t per_query = 0;
            for (const Part& p : ix->parts) {
                const uint64_t d1 = std::min<uint64_t>(p.slot_begin + p.slot_count, p.meta.doc_names.size());
                per_query += d1 > p.slot_begin ? (size_t)(d1 - p.slot_begin) : 0;
            }
            // (a single-hash query is not ordered by score but yields the same number of hits: rank_raw handles it)
            if (per_query * (ps.g1 - ps.g0) <= cap - used) {
                double t0 = now_s();
                st = rank_window(sb, 0, ps.g1 - ps.g0, per_query, hits + used);
                ix->timers[4] += now_s() - t0;
                if (st == COBS_GPU_OK) {
                    for (size_t q = ps.g0; q < ps.g1; ++q) {
                        used += per_query;
                        hit_offsets[q + 1] = used;
                    }
                    return COBS_GPU_OK;
                }
                if (st != COBS_GPU_ERR_UNSUPPORTED) return st;      // else: scores too wide for the counting sort
            }
        }
        for (size_t q = ps.g0; q < ps.g1; ++q) {
            size_t n = 0;
            if (sb->selected && sb->pool_fetched && sb->h_nhits() <= sb->hit_cap &&
                sb->h_hit_off[q - ps.g0] == sb->h_hit_off[q - ps.g0 + 1]) {
                hit_offsets[q + 1] = used;       // no document of this query reached the threshold
                continue;
            }
            double t0 = now_s();
            st = cobs_gpu_batch_hits_host(sb, q - ps.g0, num_results, overflow ? nullptr : hits + used,
                                          overflow ? 0 : cap - used, &n);
            ix->timers[4] += now_s() - t0;
            if (st == COBS_GPU_ERR_CAPACITY || (overflow && st == COBS_GPU_ERR_ARG)) overflow = true;
            else if (st != COBS_GPU_OK) return st;
            used += n;
            hit_offsets[q + 1] = used;
        }
        return COBS_GPU_OK;
    };
    size_t g0 = 0, pass_no = 0;
    hipEvent_t prev_done = nullptr;
    while (g0 < nq || (nq == 0 && g0 == 0)) {
        size_t g1 = g0;
        uint64_t table_bytes = 0, max_terms = 1;
        while (g1 < nq && g1 - g0 < max_pass) {
            // score rows of the pass: queries x slots x the score width its longest query needs
            const uint64_t terms = lens[g1] >= min_term ? lens[g1] - min_term + 1 : 1;
            const uint64_t mt = std::max(max_terms, terms);
            const int planes = scan_planes_for(mt);
            const uint64_t sb = (uint64_t)(g1 - g0 + 1) * ix->local_counts * (planes > 0 ? scan_score_bytes(planes) : 4u);
            const uint64_t tb = (uint64_t)(lens[g1] + 16) * terms_per_char;
            if (g1 > g0 && (sb > kLimit || table_bytes + tb > kLimit)) break;
            max_terms = mt;
            table_bytes += tb;
            ++g1;
        }
        if (inflight.size() == depth) {                // the slot about to be reused must be collected first
            const Pass oldest = inflight.front();
            inflight.erase(inflight.begin());
            cobs_gpu_status st = collect(oldest);
            if (st != COBS_GPU_OK) { drain(); return st; }
        }
        const int slot = (int)(pass_no % depth);
        size_t bad_local = 0;
        const double tb0 = now_s();
        cobs_gpu_status st = host_pass_begin(ix, slot, queries + g0, lens + g0, g1 - g0, threshold, topk, prev_done,
                                             &bad_local, g0);
        if (ix->tune.trace) std::fprintf(stderr, "[cobs_gpu] pass %zu: %zu queries staged + launched in %.3f ms\n", pass_no, g1 - g0, (now_s() - tb0) * 1e3);
        if (st != COBS_GPU_OK) {
            // passes before this one come first in the caller's order: report their error if they have one
            const size_t first_bad = g0 + bad_local;
            cobs_gpu_status earlier = COBS_GPU_OK;
            while (!inflight.empty() && earlier == COBS_GPU_OK) {
                const Pass ps = inflight.front();
                inflight.erase(inflight.begin());
                const std::string keep = g_last_error;
                earlier = collect(ps);
                if (earlier == COBS_GPU_OK) g_last_error = keep;
            }
            drain();
            if (earlier != COBS_GPU_OK) return earlier;
            if (bad_query) *bad_query = first_bad;
            return st;
        }
        prev_done = ix->scratch[slot]->done;
        inflight.push_back(Pass{g0, g1, slot});
        ++pass_no;
        if (nq == 0) break;
        g0 = g1;
    }
    while (!inflight.empty()) {
        const Pass ps = inflight.front();
        inflight.erase(inflight.begin());
        cobs_gpu_status st = collect(ps);
        if (st != COBS_GPU_OK) { drain(); return st; }
    }
    if (overflow) return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small; hit_offsets[nq] holds the needed size");
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_search_batch(cobs_gpu_index* ix, const char* const* queries, const size_t* lens,
                                      size_t nq, double threshold, size_t num_results,
                                      cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets,
                                      size_t* bad_query) {
    return guarded([&]() {
        return search_batch_impl(ix, queries, lens, nq, threshold, num_results, hits, cap, hit_offsets, bad_query);
    });
}

cobs_gpu_status cobs_gpu_search(cobs_gpu_index* ix, const char* query, size_t len, double threshold,
                                size_t num_results, cobs_gpu_hit* hits, size_t cap, size_t* n_hits) {
    if (!ix || !query || !n_hits) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    size_t offs[2] = {0, 0};
    cobs_gpu_status st = cobs_gpu_search_batch(ix, &query, &len, 1, threshold, num_results, hits, cap, offs, nullptr);
    *n_hits = offs[1];
    return st;
}

cobs_gpu_status cobs_gpu_counts(cobs_gpu_index* ix, const char* query, size_t len, uint32_t* counts, size_t cap) {
    if (!ix || !query || !counts) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (cap < ix->total_counts) return fail(COBS_GPU_ERR_CAPACITY, "counts buffer too small");
    return guarded([&]() -> cobs_gpu_status {
        cobs_gpu_status st = run_host_batch(ix, &query, &len, 1, 0.0, nullptr);
        if (st != COBS_GPU_OK) return st;
        double t0 = now_s();
        st = fetch_counts(ix->scratch[0], 0, counts);
        ix->timers[3] += now_s() - t0;
        return st;
    });
}

uint64_t cobs_gpu_graph_replays(const cobs_gpu_index* ix) { return ix ? ix->graph_replays : 0; }

cobs_gpu_status cobs_gpu_stream_counters(const cobs_gpu_index* ix, uint64_t out[2]) {
    if (!ix || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    out[0] = ix->stream.fetched_chunks;
    out[1] = ix->stream.streamed_chunks;
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_timers(cobs_gpu_index* ix, double out[5], int reset) {
    if (!ix) return fail(COBS_GPU_ERR_ARG, "NULL index");
    if (out) std::memcpy(out, ix->timers, sizeof ix->timers);
    if (reset) std::memset(ix->timers, 0, sizeof ix->timers);
    return COBS_GPU_OK;
}

}  // extern "C"
