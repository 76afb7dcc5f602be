// cobs_amd/csrc/pass.cpp -- one pass of the hot path over a batch (ClassicSearch::search's body, reference
// cobs/query/classic_search.cpp:403-505, for many queries at once): batch workspaces, the single upload of the
// query text, K1 per file, then per chunk K2 -- on resident data, on a chunk streamed in whole, or on the rows a
// row-selective fetch brought in --, K3 where a limit is given, and the events / counters a caller reads afterwards.
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

// row bytes one hash lookup gathers from this part (all held slices)
uint64_t gathered_row_bytes(const Part& p) {
    uint64_t n = 0;
    for (const VPage& v : p.held) n += v.ncols;
    return n;
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

extern "C" {

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
    int planes = scan_planes_for(max_terms);
    if (planes < 0) return fail(COBS_GPU_ERR_QUERY_TOO_LONG, "query too long");
    // the reference's classic_search_disable_8bit / _16bit switches (classic_search.cpp:207-209, :453-504; its tests
    // run one query under every Score width, tests/compact_index_query.cpp:54-140): a wider score type on request
    if (ix->tune.min_score_bytes >= 2 && planes < 10) planes = 10;
    if (ix->tune.min_score_bytes >= 4 && planes < 20) planes = 20;
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
    // upload layout: span_off | q_len | blk_off per file | text (+ 64: K1 reads whole dwords around a k-mer).
    // The tables come first so that their device addresses depend on the NUMBER of queries only, not on their
    // lengths: a captured graph of a small pass (host_api.cpp) bakes those addresses in and is replayed for
    // every batch of its shape class, whatever the exact lengths.
    const size_t o_span = 0;
    const size_t o_qlen = o_span + (size_t)round_up(8 * (nq + 1), 16);
    const size_t o_blk = o_qlen + (size_t)round_up(4 * std::max<size_t>(nq, 1), 16);
    const size_t blk_stride = (size_t)round_up(8 * (nq + 1), 16);
    const size_t o_text = o_blk + blk_stride * ix->parts.size();
    const size_t upload_bytes = o_text + (size_t)round_up(off + 64, 16);
    // small passes keep one allocation across lengths (a grown buffer would re-key their graphs)
    const size_t upload_cap = nq <= 16 ? std::max<size_t>(upload_bytes, 256u << 10) : upload_bytes;
    HIP_TRY(b->h_text.reserve(upload_cap));
    HIP_TRY(b->text.reserve(upload_cap));
    std::memset(b->h_text.p + o_text, 0, upload_bytes - o_text);
    for (size_t q = 0; q < nq; ++q) std::memcpy(b->h_text.p + o_text + b->span_off[q], queries[q], lens[q]);
    std::memcpy(b->h_text.p + o_span, b->span_off.data(), 8 * (nq + 1));
    if (nq) std::memcpy(b->h_text.p + o_qlen, b->lens.data(), 4 * nq);
    b->d_span_off = reinterpret_cast<const uint64_t*>(b->text.p + o_span);
    b->d_qlen = reinterpret_cast<const uint32_t*>(b->text.p + o_qlen);
    b->d_text = b->text.p + o_text;

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
        // (small passes: room for longer queries of the same class, so that the table keeps its address)
        HIP_TRY(w.table.reserve(nq <= 16 ? std::max<size_t>((size_t)w.table_entries, 1u << 18) : (size_t)w.table_entries));
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
// The shape class of the batch's pass: everything a captured graph of it bakes in besides buffer addresses --
// the number of queries, the score planes, K1's (rounded) grid, every chunk's launch geometry, and whether some
// query has a single hash in total (such a batch keeps its score rows in a top-k pass).  Batches of one class
// differ in their query lengths only, which the kernels read from device tables: one graph serves them all.
uint64_t cobs_amd::pass_shape_class(const cobs_gpu_batch* b) {
    const cobs_gpu_index* ix = b->ix;
    uint64_t key = 1469598103934665603ull;
    auto mixin = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    const size_t nq = b->nq;
    mixin(nq); mixin((uint64_t)b->planes); mixin(b->elem_bytes);
    mixin(round_up(nq ? b->span_off[nq] : 0, 1024));
    bool single = false;
    for (size_t q = 0; q < nq && !single; ++q) single = total_hashes(b, q) <= 1;
    mixin(single);
    for (size_t f = 0; f < ix->parts.size() && nq; ++f) {
        const Part& p = ix->parts[f];
        for (const Chunk& c : p.chunks) {
            const ScanGeom g = scan_geometry(c, b->work[f].h_blk_off[nq] / nq, (b->max_terms + 7) / 8, p.meta.num_hashes,
                                             ix->waves_per_group, b->planes, p.idx64, ix->tune);
            mixin(g.tile_w); mixin((uint64_t)g.nwaves); mixin(g.multi_query);
        }
    }
    return key;
}

void cobs_amd::set_run_state(cobs_gpu_batch* b, double threshold, size_t topk, bool want_counts) {
    cobs_gpu_index* ix = b->ix;
    b->ran = false;
    b->synced = false;
    b->pool_fetched = false;
    b->pool_sorted = false;
    b->topk_fetched = false;
    b->rows_q0 = b->rows_q1 = 0;
    b->view_global = false;
    b->pool_global = false;
    b->pool_owned = false;
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
    // (A streamed sub-index cut into ROW ranges is counted range by range: K2 sees partial counts there.  Its ranges add
    // up in a scratch matrix of the sub-index's own width and the selection runs over that after the last range --
    // run_impl, `acc_mode` --, so such a handle selects like any other.  Until round 6 it kept score rows of the whole
    // index instead and answered hits and limits from them.)
    b->have_counts = want_counts || (!b->selected && !b->topk_direct);
}

// The per-(file, query) thresholds ceil(threshold * T) (classic_search.cpp:444-449) in the pinned
// buffer the H2D copies of a run read.  A captured graph holds those copies as nodes that read the
// buffer when the graph is LAUNCHED, and the buffer is shared by every shape of the batch: a replay
// has to write its own thresholds first (whatever ran in between left its own there).
void cobs_amd::stage_thresholds(cobs_gpu_batch* b, double threshold) {
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
    // K1 on the batch's own stream (tuning key hash_stream; device-resident batches only -- the scratch batches of the
    // host-buffer API live on one stream each and may be under graph capture): it is ordered after the previous run of
    // THIS batch (and the exchange that followed it: both end with run_done), whose K2 read the row-index tables and
    // the flags K1 is about to overwrite, and before this run's K2 by the event `hashed` -- nothing else.  The
    // hashing of one sub-batch then runs under the scan / exchange of another (bench.py's sharded flow).
    const bool split = ix->tune.hash_stream != 0 && b->own_stream == nullptr && nq > 0;
    hipStream_t hs = st;
    if (split) {
        if (!b->hash_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&b->hash_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&b->hashed, hipEventDisableTiming));
        }
        hs = b->hash_stream;
        if (b->run_seq) HIP_TRY(hipStreamWaitEvent(hs, b->run_done, 0));
    }
    // device flags: first invalid query = none, selected hits = 0
    HIP_TRY(launch_clear_flags(b->flags.p, hs));      // (a kernel: a captured memset node is not safe to replay, fetch_kernels.hip)
    if (need_thr) {
        stage_thresholds(b, threshold);
        for (size_t f = 0; f < ix->parts.size(); ++f)
            if (nq) HIP_TRY(hipMemcpyAsync(b->work[f].thr.p, b->h_thr_stage.p + f * nq, 4 * nq, hipMemcpyHostToDevice, st));
    }
    std::vector<std::vector<bool>> fetch_unit(ix->parts.size());
    std::vector<std::vector<uint32_t>> unit_merge(ix->parts.size());     // row ranges of one sub-index merged into the unit (>= 1)
    hipEvent_t* ev = b->ev[b->run_seq % cobs_gpu_batch::kRing];
    b->ev_split[b->run_seq % cobs_gpu_batch::kRing] = split;
    HIP_TRY(hipEventRecord(ev[0], hs));
    StreamBufs& sbufs = ix->stream;
    // ---- K1 once per file and pass: the row-index table covers every held sub-index, the chunks (launches) of the file
    // pick their sub-indexes by PageDev::tpage.  All files first: what a streamed file's pass looks like -- which of its
    // chunks are copied whole, which are fetched row by row -- is decided from what the batch LOOKS UP (below).
    std::vector<uint64_t> cnt_off(ix->parts.size() + 1, 0);
    // (a SMALL batch needs no counting: if every streamed chunk is fetched by rows even when every table entry is given a
    // slot -- E rows per page, the round-3 layout -- that bound is used as the pages' capacity and the pass has no host
    // synchronisation before its scans: a single query against the 18.4 GB file)
    std::vector<bool> by_bound(ix->parts.size(), false);
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        const Part& p = ix->parts[f];
        bool counted = nq && ix->tune.row_fetch != 0 && p.streamed && p.file_dev && !p.synthetic && p.ncounters;
        if (counted) {
            const uint64_t E = (b->work[f].h_blk_off[nq] + nq) * 8ull * p.meta.num_hashes;
            auto fits = [&](const Chunk& c) {
                const uint64_t rows = E * c.vp.size();
                return c.fetch_ok && !c.resident && E < 0xFFFFFFF0ull && (rows + c.vp.size()) * (uint64_t)c.pitch <= sbufs.sbuf[0].cap &&
                       rows * (uint64_t)c.pitch * ix->tune.row_fetch_alpha <= c.bytes;
            };
            bool all = true;
            for (const Chunk& c : p.chunks) all = all && (c.resident || fits(c));
            for (const Chunk& g : p.fetch_groups) all = all && fits(g);
            if (all) { by_bound[f] = true; counted = false; }
        }
        cnt_off[f + 1] = cnt_off[f] + (counted ? p.ncounters : 0);
    }
    if (cnt_off.back()) {
        HIP_TRY(sbufs.d_counts.reserve((size_t)cnt_off.back()));
        HIP_TRY(sbufs.h_counts.reserve((size_t)cnt_off.back()));
        HIP_TRY(hipMemsetAsync(sbufs.d_counts.p, 0, (size_t)cnt_off.back() * sizeof(unsigned long long), hs));
    }
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        Part& p = ix->parts[f];
        if (nq == 0 || p.chunks.empty()) continue;
        HashArgs ha;
        ha.text = b->d_text;
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
        // (the kernel bounds itself by span_off[nq] on the device; the grid is rounded up so that a
        // captured launch serves every batch of its shape class)
        HIP_TRY(launch_hash(ha, round_up(b->span_off[nq], 1024), hs));
        if (cnt_off[f + 1] > cnt_off[f]) {
            // how many rows the batch looks up in every streamed piece of this file (one counter per whole slice / row range)
            CountArgs ca;
            ca.table = b->work[f].table.p;
            ca.blk_off = b->work[f].blk_off;
            ca.tpages = p.d_tpages;
            ca.cpages = p.d_cpages;
            ca.counts = sbufs.d_counts.p + cnt_off[f];
            ca.nq = (uint32_t)nq;
            ca.table_npages = p.num_tpages();
            ca.num_hashes = (uint32_t)p.meta.num_hashes;
            ca.ncounters = p.ncounters;
            const uint64_t total = (b->work[f].h_blk_off[nq] + nq) * 8ull * p.meta.num_hashes * p.num_tpages();
            HIP_TRY(launch_count_rows(ca, total, p.idx64, hs));
        }
    }
    HIP_TRY(hipEventRecord(ev[1], hs));          // K1 / K2 split of the timing events
    if (cnt_off.back()) {
        HIP_TRY(hipMemcpyAsync(sbufs.h_counts.p, sbufs.d_counts.p, (size_t)cnt_off.back() * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, hs));
        HIP_TRY(hipStreamSynchronize(hs));       // (an out-of-core pass waits for its stream buffers anyway)
    }
    bool scan_marked = false;
    if (split && nq) {       // the K2 launches wait for the tables (and for the flags fill)
        HIP_TRY(hipEventRecord(b->hashed, hs));
        HIP_TRY(hipStreamWaitEvent(st, b->hashed, 0));
        HIP_TRY(hipEventRecord(ev[3], st));
        scan_marked = true;
    }
    if (b->scan_after) {
        // a pass of the host-buffer API: its upload and K1 ran beside the scan of the pass before it; the scans follow
        // each other (host_api.cpp: host_pass_begin)
        HIP_TRY(hipStreamWaitEvent(st, b->scan_after, 0));
        b->scan_after = nullptr;
        if (!scan_marked) {
            HIP_TRY(hipEventRecord(ev[3], st));
            b->ev_split[b->run_seq % cobs_gpu_batch::kRing] = true;
            scan_marked = true;
        }
    }
    // scan geometry of every (file, chunk); with tile-level top-k also the files' places in the candidate pool
    std::vector<std::vector<ScanGeom>> geoms(ix->parts.size());
    std::vector<uint64_t> cand_off(ix->parts.size() + 1, 0);
    std::vector<uint32_t> cand_tiles(ix->parts.size(), 0), cand_stride(ix->parts.size(), 0);
    // the units of a file's pass: its chunks -- or, for a streamed file of which this batch fetches EVERY streamed chunk
    // by rows, the runs of chunks of equal pitch merged (one gather + one scan per run: a single query against a file
    // of fifty chunks is a handful of launch pairs)
    std::vector<std::vector<const Chunk*>> units(ix->parts.size());
    // rows a row-selective fetch of unit `c` gathers (EXACT: counted on the device right after K1) and what they occupy
    auto looked_up_rows = [&](size_t f, const Chunk& c) {
        if (by_bound[f]) return (uint64_t)((b->work[f].h_blk_off[nq] + nq) * 8ull * ix->parts[f].meta.num_hashes * c.vp.size());
        uint64_t n = 0;
        for (const auto& cp : c.cp)
            for (uint32_t k = 0; k < cp.second; ++k) n += sbufs.h_counts.p[cnt_off[f] + cp.first + k];
        return n;
    };
    for (size_t f = 0; f < ix->parts.size() && nq; ++f) {
        const Part& p = ix->parts[f];
        const uint64_t E = (b->work[f].h_blk_off[nq] + nq) * 8ull * p.meta.num_hashes;
        // Whole chunk, or only the rows this batch looks up?  Fetching them moves (looked-up rows) x pitch bytes over PCIe
        // at the rate random rows come in, copying the chunk moves all of its rows at the slab rate (row_fetch_alpha
        // prices the difference); the gathered rows must fit a stream buffer -- they always do when they are fewer
        // bytes than the chunk, whatever the buffers' size: the gather packs exactly the looked-up rows (round 5;
        // until then the buffer had a place for every table entry, which 256 MiB buffers do not hold for 256 queries).
        // The reference's mmap / AIO back-ends always take the first form (compact_index/mmap_search_file.cpp:34-67,
        // aio_search_file.cpp:58-97).
        auto fetchable = [&](const Chunk& c) {
            if (!(cnt_off[f + 1] > cnt_off[f] || by_bound[f]) || !c.fetch_ok || c.resident || E >= 0xFFFFFFF0ull) return false;
            const uint64_t rows = looked_up_rows(f, c);
            const uint64_t gathered = (rows + c.vp.size()) * (uint64_t)c.pitch;
            return gathered <= sbufs.sbuf[0].cap && rows * (uint64_t)c.pitch * ix->tune.row_fetch_alpha <= c.bytes &&
                   rows < 0xFFFFFFF0ull;
        };
        bool all = !p.fetch_groups.empty();
        for (const Chunk& c : p.chunks) all = all && (c.resident || fetchable(c));
        for (const Chunk& g : p.fetch_groups) all = all && fetchable(g);
        if (!all) {
            // Consecutive ROW RANGES of one sub-index that are all fetched by rows share a gather and a scan while their
            // looked-up rows fit a stream buffer together: a range's scan walks every term of every query whatever the
            // range holds (terms outside it read the zero row), so 234 ranges of a 62 GB sub-index would be 234 full scans
            // for 15.7 GB of looked-up rows -- merged, 59.
            for (size_t ci = 0; ci < p.chunks.size();) {
                const Chunk& c = p.chunks[ci];
                uint32_t m = 1;
                if (c.row_range && fetchable(c)) {
                    uint64_t rows = looked_up_rows(f, c);
                    while (ci + m < p.chunks.size()) {
                        const Chunk& d = p.chunks[ci + m];
                        if (!d.row_range || d.vp[0].fp != c.vp[0].fp || d.range_no != c.range_no + m || !fetchable(d)) break;
                        const uint64_t more = looked_up_rows(f, d);
                        if ((rows + more + 1) * (uint64_t)c.pitch > sbufs.sbuf[0].cap || rows + more >= 0xFFFFFFF0ull) break;
                        rows += more;
                        ++m;
                    }
                }
                units[f].push_back(&c);
                unit_merge[f].push_back(m);
                ci += m;
            }
        } else {
            // resident chunks where they lie, every run of streamed chunks as its fetch group -- in chunk (= document) order
            size_t gi = 0;
            for (size_t ci = 0; ci < p.chunks.size(); ++ci) {
                if (p.chunks[ci].resident) units[f].push_back(&p.chunks[ci]);
                else if (gi < p.fetch_groups.size() && p.fetch_groups[gi].first_chunk == ci) units[f].push_back(&p.fetch_groups[gi++]);
            }
        }
        unit_merge[f].resize(units[f].size(), 1u);
        fetch_unit[f].resize(units[f].size());
        for (size_t u = 0; u < units[f].size(); ++u) fetch_unit[f][u] = fetchable(*units[f][u]);
    }
    for (size_t f = 0; f < ix->parts.size() && nq; ++f) {
        const Part& p = ix->parts[f];
        for (const Chunk* cp : units[f]) {
            const Chunk& c = *cp;
            geoms[f].push_back(scan_geometry(c, b->work[f].h_blk_off[nq] / nq, (b->max_terms + 7) / 8, p.meta.num_hashes,
                                             ix->waves_per_group, b->planes, p.idx64, ix->tune));
            // (a sub-index counted in row ranges is ONE tile of the candidate pool: its k best come from the accumulated
            // scores after the last range, below)
            if (c.row_range) cand_tiles[f] += c.range_no == 0 ? 1u : 0u;
            else cand_tiles[f] += (c.total_chunks + geoms[f].back().tile_w - 1) / geoms[f].back().tile_w;
        }
        cand_stride[f] = (uint32_t)round_up((uint64_t)cand_tiles[f] * topk, 8);
        cand_off[f + 1] = cand_off[f] + (b->topk_direct ? (uint64_t)nq * cand_stride[f] : 0);
    }
    if (b->topk_direct) {
        // a pool larger than the score rows it replaces (a large k on a narrow score type), 16 GiB of candidates, or no
        // room for the pool: take the score rows instead
        if (cand_off.back() * sizeof(uint2) > (uint64_t)nq * ix->local_counts * b->elem_bytes || cand_off.back() > (1ull << 31) ||
            b->cand.reserve((size_t)cand_off.back()) != hipSuccess) {
            (void)hipGetLastError();
            b->topk_direct = false;
            b->have_counts = true;
            HIP_TRY(b->counts.reserve((size_t)(nq * ix->local_counts * b->elem_bytes)));
        }
    }
    uint64_t launches = 0;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        Part& p = ix->parts[f];
        if (nq == 0 || p.chunks.empty() || units[f].empty()) continue;
        uint32_t tile_base = 0;
        bool fetch_ready = false;
        if (p.streamed && ((p.file_dev && ix->tune.row_fetch != 0) || p.has_row_ranges)) {
            // a row-selective chunk (and a row-range chunk) gets its own row-index table (one per stream buffer); sized before the
            // chunk loop, when no scan of this handle is reading the old ones any more
            const size_t need = (size_t)b->work[f].table_entries * 4;
            for (int i = 0; i < 2; ++i) {
                if (sbufs.table2[i].cap >= need) continue;
                if (sbufs.used[i]) HIP_TRY(hipEventSynchronize(sbufs.scanned[i]));
                HIP_TRY(sbufs.table2[i].reserve(need));
            }
        }
        for (size_t ci = 0; ci < units[f].size(); ++ci) {
            const Chunk& c = *units[f][ci];
            const uint8_t* data = c.d_data;
            int buf = 0;
            // a later row range of a sub-index: its partial scores go to a scratch matrix and are added to the rows
            // (however the range's rows come in: streamed whole, or only the looked-up ones fetched)
            const bool partial = c.row_range && c.range_no > 0;
            // A pass that selects in the scan (hits into the pool, a tile's k best) and keeps no score rows: the ranges of
            // such a sub-index add up in a scratch matrix of its own width (first range: written there, later ones: added),
            // and the selection runs over that matrix after its last range -- the same hit pool / candidate pool the
            // resident path fills (reference: the filter is the same whatever back-end gathered the rows,
            // classic_search.cpp:127-145).  C3's largest sub-index, 10k queries: 250 MB of scratch instead of 2 GB of rows.
            const bool acc_mode = c.row_range && !b->have_counts && (b->selected || b->topk_direct);
            const PageDev* pages_dev = (partial || acc_mode) ? c.d_pages_acc : c.d_pages;
            const void* table_dev = b->work[f].table.p;
            bool unit_fetched = false;
            uint64_t unit_rows = 0;
            uint32_t unit_zero = 0;
            const bool stream_this = p.streamed && !c.resident;      // (a streamed file may keep some of its slices in HBM)
            if (stream_this) {
                // double buffer shared by all streamed files: the next chunk goes to the buffer
                // whose last scan is done
                buf = (int)(sbufs.seq++ & 1);
                if (sbufs.used[buf]) HIP_TRY(hipEventSynchronize(sbufs.scanned[buf]));
                const uint64_t E = (b->work[f].h_blk_off[nq] + nq) * 8ull * p.meta.num_hashes;
                const bool fetch = fetch_unit[f][ci];
                if (fetch) {
                    if (!fetch_ready) {          // the gather reads K1's table: once per file and pass
                        HIP_TRY(hipEventRecord(sbufs.hashed, st));
                        HIP_TRY(hipStreamWaitEvent(sbufs.copy_stream, sbufs.hashed, 0));
                        HIP_TRY(hipStreamWaitEvent(sbufs.prep_stream, sbufs.hashed, 0));
                        fetch_ready = true;
                    }
                    // the unit's pages as the gather sees them: exactly the looked-up rows of each, packed, a zero row behind
                    const size_t np = c.vp.size();
                    HIP_TRY(sbufs.h_gpages[buf].reserve(np));
                    HIP_TRY(sbufs.gpages[buf].reserve(np));
                    HIP_TRY(sbufs.cursor[buf].reserve(np));
                    GatherPage* gp = sbufs.h_gpages[buf].p;
                    uint64_t slot = 0, bm_words = 0;
                    uint32_t bm_max = 0;
                    for (size_t k = 0; k < np; ++k) {
                        uint64_t cnt = by_bound[f] ? E : 0;       // (no count: a place for every table entry)
                        for (uint32_t j = 0; j < c.cp[k].second && !by_bound[f]; ++j) cnt += sbufs.h_counts.p[cnt_off[f] + c.cp[k].first + j];
                        gp[k].src = c.src[k];
                        gp[k].row0 = c.pages[k].row0;
                        gp[k].nrows = c.pages[k].sig;
                        for (uint32_t m = 1; m < unit_merge[f][ci]; ++m) {      // (merged row ranges: one page, k == 0)
                            const Chunk& d = (&c)[m];
                            gp[k].nrows += d.pages[0].sig;
                            cnt += sbufs.h_counts.p[cnt_off[f] + d.cp[0].first];
                        }
                        gp[k].slot0 = slot;
                        gp[k].count = (uint32_t)cnt;
                        gp[k].tpage = c.pages[k].tpage;
                        gp[k].leader = (uint32_t)k;
                        for (size_t j = 0; j < k; ++j)       // column slices of one sub-index share the row list of the first
                            if (gp[j].tpage == gp[k].tpage && gp[j].row0 == gp[k].row0) { gp[k].leader = (uint32_t)j; break; }
                        gp[k].valid_bytes = c.pages[k].valid_bytes;
                        // the row bitmap of a leader page (one bit per row of its range): the gather hands a slot to every
                        // DISTINCT looked-up row -- `cnt`, the look-ups, bounds them and stays the page's capacity
                        gp[k].bm_off = bm_words;
                        gp[k].bm_words = 0;
                        gp[k].reserved = 0;
                        if (gp[k].leader == (uint32_t)k) {
                            const uint64_t w = (gp[k].nrows + 31u) / 32u;
                            if (w > 0xFFFFFFF0ull) return fail(COBS_GPU_ERR_UNSUPPORTED, "sub-index too large for a row-selective fetch");
                            gp[k].bm_words = (uint32_t)w;
                            bm_words += w;
                            bm_max = std::max(bm_max, (uint32_t)w);
                        }
                        slot += cnt + 1;
                    }
                    HIP_TRY(sbufs.rowlist[buf].reserve((size_t)slot));
                    HIP_TRY(sbufs.bitmap[buf].reserve((size_t)std::max<uint64_t>(bm_words, 1)));
                    HIP_TRY(sbufs.bprefix[buf].reserve((size_t)std::max<uint64_t>(bm_words, 1)));
                    if (!sbufs.d_fetched.p) {
                        HIP_TRY(sbufs.d_fetched.reserve(1));
                        HIP_TRY(hipMemset(sbufs.d_fetched.p, 0, sizeof(unsigned long long)));
                    }
                    // (the slot assignment on its own stream: it overlaps the copy of the previous unit -- 1.1 ms of table walking per
                    // unit beside 3.6-4.2 ms of waiting for PCIe, profiles/r05_out_of_core_kernel_stats.csv; the last scan that read
                    // this buffer's second table has finished: the host waited for scanned[buf] above)
                    // (... in a pass of several units -- 256 queries, 6 units: 24.1 against 25.1 ms; a pass of ONE has no previous copy
                    // to hide behind and keeps the assignment on the copy stream, without the event hop)
                    size_t fetched_units = 0;
                    for (size_t u = 0; u < fetch_unit[f].size(); ++u) fetched_units += fetch_unit[f][u] ? 1 : 0;
                    const bool pipelined = fetched_units > 16;
                    const bool own_prep = fetched_units > 1;
                    hipStream_t prep = own_prep ? sbufs.prep_stream : sbufs.copy_stream;
                    HIP_TRY(hipMemcpyAsync(sbufs.gpages[buf].p, gp, np * sizeof(GatherPage), hipMemcpyHostToDevice, prep));
                    HIP_TRY(hipMemsetAsync(sbufs.cursor[buf].p, 0, np * sizeof(unsigned long long), prep));
                    HIP_TRY(hipMemsetAsync(sbufs.bitmap[buf].p, 0, (size_t)bm_words * sizeof(uint32_t), prep));
                    GatherArgs ga;
                    ga.file = p.file_dev;
                    ga.table = b->work[f].table.p;
                    ga.table2 = sbufs.table2[buf].p;
                    ga.blk_off = b->work[f].blk_off;
                    ga.pages = sbufs.gpages[buf].p;
                    ga.pages_in = pages_dev;            // (a later row range: the page with slot0 = 0, see `partial`)
                    ga.pages2 = c.d_pages2[buf];
                    ga.dst = sbufs.sbuf[buf].p;
                    ga.rowlist = sbufs.rowlist[buf].p;
                    ga.cursor = sbufs.cursor[buf].p;
                    ga.bitmap = sbufs.bitmap[buf].p;
                    ga.bprefix = sbufs.bprefix[buf].p;
                    ga.fetched_bytes = sbufs.d_fetched.p;
                    ga.entries = E;
                    ga.total_rows = slot;
                    ga.src_pitch = p.meta.page_row_bytes();
                    ga.nq = (uint32_t)nq;
                    ga.npages = (uint32_t)np;
                    ga.table_npages = p.num_tpages();
                    ga.num_hashes = (uint32_t)p.meta.num_hashes;
                    ga.pitch = c.pitch;
                    // (two pieces per thread in flight over PCIe: 801 / 798 ms against 817 / 818 per 184 GB pass on one box, with
                    // non-temporal loads 822: profiles/r06_gather_blocks_ab.txt; COBS_GPU_GATHER_EXP is the A/B switch)
                    static const uint32_t gather_exp = getenv("COBS_GPU_GATHER_EXP") ? (uint32_t)std::strtoul(getenv("COBS_GPU_GATHER_EXP"), nullptr, 0) : 2u;
                    ga.exp = gather_exp;
                    HIP_TRY(launch_gather_assign(ga, p.idx64, bm_max, prep));
                    if (own_prep) {
                        HIP_TRY(hipEventRecord(sbufs.assigned[buf], sbufs.prep_stream));
                        HIP_TRY(hipStreamWaitEvent(sbufs.copy_stream, sbufs.assigned[buf], 0));
                    }
                    // A pass of MANY fetched units is a pipeline -- gather(i + 1) | compact + scan + add(i) --: the copy, which only
                    // waits for PCIe, then runs on 128 work-groups so that the kernels beside it find CUs (the 184 GB file, 253
                    // units: 1.20 s per pass with a grid of one thread per piece, 1.11 s with 256 work-groups, 1.03 s with 128);
                    // a pass of a few units (a small batch) has nothing to overlap and takes the wide grid (256 queries: 24.1
                    // against 25.4 ms).
                    HIP_TRY(launch_gather_copy(ga, pipelined ? 128u : 1024u, sbufs.copy_stream));
                    pages_dev = c.d_pages2[buf];
                    table_dev = sbufs.table2[buf].p;
                    unit_fetched = true;
                    unit_rows = slot - np;
                    unit_zero = gp[0].count;
                    ++sbufs.fetched_chunks;
                    sbufs.lookup_bytes += (slot - np) * (uint64_t)c.pitch;       // (what crosses PCIe: the DISTINCT rows, counted on the device)
                } else {
                    cobs_gpu_status cs = stream_chunk_in(ix, p, c, buf);
                    if (cs != COBS_GPU_OK) return cs;
                    ++sbufs.streamed_chunks;
                    sbufs.streamed_bytes += c.stage_bytes;
                }
                HIP_TRY(hipEventRecord(sbufs.copied[buf], sbufs.copy_stream));
                HIP_TRY(hipStreamWaitEvent(st, sbufs.copied[buf], 0));
                data = sbufs.sbuf[buf].p;
                if (c.row_range && !fetch) {
                    // the buffer holds rows [row0, row0 + n) of the sub-index: this chunk's scan reads K1's indices
                    // shifted into the range, every row outside it as the buffer's zero row
                    RemapArgs ra;
                    ra.table = b->work[f].table.p;
                    ra.table2 = sbufs.table2[buf].p;
                    ra.blk_off = b->work[f].blk_off;
                    ra.row0 = c.pages[0].row0;
                    ra.nrows = c.pages[0].sig;
                    ra.nq = (uint32_t)nq;
                    ra.tpage = c.pages[0].tpage;
                    ra.table_npages = p.num_tpages();
                    ra.num_hashes = (uint32_t)p.meta.num_hashes;
                    HIP_TRY(launch_remap_rows(ra, (b->work[f].h_blk_off[nq] + nq) * 8ull * p.meta.num_hashes, p.idx64, st));
                    table_dev = sbufs.table2[buf].p;
                }
            }
            // A row-range unit: only the terms whose row it HOLDS count, every other entry of its table names the zero row --
            // 99 % of them for a 256 MiB range of a 62 GB sub-index, and walking them all made the scan of such a unit
            // issue-bound (4.4 ms for 10k queries; 254 units per pass of the 184 GB file: 1.2 s of scans beside 0.9 s of PCIe).
            // Where few terms are in range the scan gets a compact table of them (compact_*_kernel) and its own block offsets.
            const uint64_t* blk_dev = b->work[f].blk_off;
            ScanGeom geom = geoms[f][ci];
            if (stream_this && c.row_range && c.vp.size() == 1 && p.meta.num_hashes == 1 && !p.idx64 && ix->tune.compact_terms != 0 &&
                !ix->tune.lds_staged && table_dev == (const void*)sbufs.table2[buf].p) {
                const uint64_t E = (b->work[f].h_blk_off[nq] + nq) * 8ull;
                uint64_t rows_in = unit_rows;                      // (a fetched unit: exact)
                uint32_t zero_idx = unit_zero;
                bool exact = unit_fetched;
                if (!unit_fetched) {
                    exact = cnt_off[f + 1] > cnt_off[f];
                    zero_idx = (uint32_t)c.pages[0].sig;
                    const uint64_t sig = p.meta.signature_sizes[c.vp[0].fp];
                    rows_in = cnt_off[f + 1] > cnt_off[f] ? sbufs.h_counts.p[cnt_off[f] + c.cp[0].first]
                                                          : (uint64_t)((long double)E * (long double)c.pages[0].sig / (long double)std::max<uint64_t>(sig, 1));
                }
                if (rows_in * 4 < E) {
                    // blocks of the compact table: sum over queries of ceil(n / 8) + the padding blocks -- from the exact count, or
                    // (no count: the mapping is not registered, row_fetch is off) no more than the table it is made from
                    const uint64_t blocks_bound = exact ? rows_in / 8 + 2 * nq : b->work[f].h_blk_off[nq] + nq;
                    HIP_TRY(sbufs.table3[buf].reserve((size_t)(blocks_bound * p.num_tpages() * 8)));
                    HIP_TRY(sbufs.blk2[buf].reserve(nq + 1));
                    HIP_TRY(sbufs.blkcnt[buf].reserve(nq + 1));
                    CompactArgs ca;
                    ca.table2 = reinterpret_cast<const uint32_t*>(sbufs.table2[buf].p);
                    ca.table3 = sbufs.table3[buf].p;
                    ca.blk_off = b->work[f].blk_off;
                    ca.blk2 = sbufs.blk2[buf].p;
                    ca.cnt = sbufs.blkcnt[buf].p;
                    ca.nq = (uint32_t)nq;
                    ca.tpage = c.pages[0].tpage;
                    ca.table_npages = p.num_tpages();
                    ca.zero_idx = zero_idx;
                    HIP_TRY(launch_compact_terms(ca, st));
                    table_dev = sbufs.table3[buf].p;
                    blk_dev = sbufs.blk2[buf].p;
                    const uint64_t mean2 = std::max<uint64_t>(1, (rows_in / std::max<size_t>(nq, 1) + 7) / 8);
                    geom = scan_geometry(c, mean2, mean2 * 4 + 4, 1, ix->waves_per_group, b->planes, false, ix->tune);
                }
            }
            uint32_t part_slots = 0;
            if (partial || acc_mode) {
                part_slots = (uint32_t)(c.vp[0].ncols * 8);
                if (partial) HIP_TRY(b->counts_part.reserve((size_t)nq * part_slots * b->elem_bytes));
                if (acc_mode) HIP_TRY(b->counts_acc.reserve((size_t)nq * part_slots * b->elem_bytes));
            }
            ScanArgs sa;
            sa.blob = data;
            sa.pages = pages_dev;
            sa.table = table_dev;
            sa.blk_off = blk_dev;
            sa.counts = partial ? b->counts_part.p : acc_mode ? b->counts_acc.p : b->counts.p;
            sa.thresholds = (b->selected && !acc_mode) ? b->work[f].thr.p : nullptr;
            sa.hits = b->hits.p;
            sa.hit_count = reinterpret_cast<unsigned long long*>(b->flags.p + 2);
            sa.counts_stride = (partial || acc_mode) ? part_slots : ix->local_counts;
            sa.counts_offset = (partial || acc_mode) ? 0 : p.local_offset;
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
            sa.write_counts = (b->have_counts || acc_mode) ? 1 : 0;
            sa.idx64 = p.idx64 ? 1u : 0u;
            const int nwaves = geom.nwaves;
            sa.tile_w = geom.tile_w;
            const bool tile_select = b->topk_direct && !acc_mode;
            sa.cand = tile_select ? b->cand.p + cand_off[f] : nullptr;
            sa.topk_k = tile_select ? (uint32_t)topk : 0u;
            sa.cand_stride = cand_stride[f];
            sa.tile_base = tile_base;
            // K2 filters by threshold only when it selects: into the hit pool, or the tile's k best
            if (tile_select) sa.thresholds = need_thr ? b->work[f].thr.p : nullptr;
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
            sa.exp = ix->tune.exp;
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
            if (!acc_mode) tile_base += ntiles;
            ++launches;
            if (partial) {
                AddScoresArgs aa;
                aa.dst = acc_mode ? b->counts_acc.p : b->counts.p;
                aa.src = b->counts_part.p;
                aa.dst_stride = acc_mode ? part_slots : ix->local_counts;
                aa.dst_offset = acc_mode ? 0 : p.local_offset + c.pages[0].slot0;
                aa.nslots = part_slots;
                aa.nq = (uint32_t)nq;
                aa.elem_bytes = b->elem_bytes;
                HIP_TRY(launch_add_scores(aa, st));
            }
            // the last range of the sub-index in this pass (its ranges are consecutive units): select from what they added up
            const bool last_range = acc_mode && (ci + 1 == units[f].size() || !units[f][ci + 1]->row_range ||
                                                 units[f][ci + 1]->vp[0].fp != c.vp[0].fp);
            if (last_range) {
                const uint32_t nvalid = std::min<uint32_t>(part_slots, c.pages[0].valid_bytes * 8u);
                if (b->selected) {
                    SelectRowsArgs sr;
                    sr.scores = b->counts_acc.p;
                    sr.thresholds = b->work[f].thr.p;
                    sr.hits = b->hits.p;
                    sr.hit_count = reinterpret_cast<unsigned long long*>(b->flags.p + 2);
                    sr.stride = part_slots;
                    sr.nslots = nvalid;
                    sr.nq = (uint32_t)nq;
                    sr.elem_bytes = b->elem_bytes;
                    sr.doc0 = c.pages[0].doc0;
                    sr.num_docs = (uint32_t)p.meta.doc_names.size();
                    sr.part = (uint32_t)f;
                    sr.hit_cap = b->hit_cap;
                    HIP_TRY(launch_select_rows(sr, st));
                } else {
                    // the sub-index's k best under (score desc, document asc) as ONE tile of the candidate pool: K3 over its
                    // accumulated rows, ordered -- K3's merge only needs equal scores in document order inside a tile
                    TopkArgs ta{};
                    ta.counts = b->counts_acc.p;
                    ta.score_bytes = b->elem_bytes;
                    ta.thresholds = need_thr ? b->work[f].thr.p : nullptr;
                    ta.out = b->cand.p + cand_off[f] + (uint64_t)tile_base * topk;
                    ta.out_count = nullptr;
                    ta.out_stride = cand_stride[f];
                    ta.pad_out = 1;
                    ta.counts_stride = part_slots;
                    ta.counts_offset = 0;
                    ta.nslots = nvalid;
                    ta.doc_base = c.pages[0].doc0;
                    ta.num_docs = (uint32_t)p.meta.doc_names.size();
                    ta.k = (uint32_t)topk;
                    ta.nq = (uint32_t)nq;
                    ta.score_bits = (uint32_t)b->planes;
                    ta.levels = ((uint32_t)b->planes + 11u) / 12u;
                    ta.level_bits = ((uint32_t)b->planes + ta.levels - 1u) / ta.levels;
                    ta.sort_limit = (uint32_t)topk;
                    HIP_TRY(launch_topk(ta, st));
                }
                tile_base += 1;
            }
            if (stream_this) {
                HIP_TRY(hipEventRecord(sbufs.scanned[buf], st));
                sbufs.used[buf] = true;
            }
        }
    }
    if (split && !scan_marked) {         // (no file launched anything: keep the flags fill ordered before the caller's stream)
        HIP_TRY(hipEventRecord(b->hashed, hs));
        HIP_TRY(hipStreamWaitEvent(st, b->hashed, 0));
        HIP_TRY(hipEventRecord(ev[3], st));
    }
    HIP_TRY(hipEventRecord(ev[2], st));
    b->scan_end = ev[2];                 // (host-buffer passes chain their SCANS: the next one may start under this run's K3)
    if (use_topk && nq) {
        for (size_t f = 0; f < ix->parts.size(); ++f) {
            const Part& p = ix->parts[f];
            TopkArgs ta{};
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
        HIP_TRY(hipEventElapsedTime(&c, b->ev_split[r % cobs_gpu_batch::kRing] ? ev[3] : ev[1], ev[2]));
        h += a;
        s += c;
    }
    const double n = (double)(b->run_seq - first);
    b->read_seq = b->run_seq;
    if (hash_ms) *hash_ms = (float)(h / n);
    if (scan_ms) *scan_ms = (float)(s / n);
    return COBS_GPU_OK;
}


}  // extern "C"
