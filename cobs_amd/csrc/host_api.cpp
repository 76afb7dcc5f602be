// cobs_amd/csrc/host_api.cpp -- the host-buffer search API (cobs_gpu_search / _search_batch / _counts: the
// reference's operator call, cobs/query/search.hpp:39-42, for one query or many): calls are cut into device
// passes that are pipelined over three scratch batches (upload of pass i+1 and ranking of pass i-1 under the scan
// of pass i), small passes are captured into hipGraphs and replayed, results are collected per pass.
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

extern "C" {

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
    b->flags_landing = false;
    ix->host_passes++;
    double t0 = now_s();
    size_t bad_local = 0;
    cobs_gpu_status st = set_queries_on(b, queries, lens, nq, b->own_stream, false, &bad_local, index_base);
    if (st != COBS_GPU_OK && bad_at) *bad_at = bad_local;
    if (st != COBS_GPU_OK) return st;
    ix->timers[1] += now_s() - t0;
    // the previous pass: a large pass only keeps its SCAN behind it (run_impl waits between K1 and K2: upload and hashing
    // of this pass run beside the scan of that one -- 0.12 ms of idle device per pass otherwise); small passes (a
    // captured graph) and out-of-core passes (shared stream buffers) wait here
    bool any_streamed_ = false;
    for (const auto& p : ix->parts) any_streamed_ = any_streamed_ || p.streamed;
    const bool scan_only_after = after && nq > 16 && !any_streamed_;
    if (after && !scan_only_after) HIP_TRY(hipStreamWaitEvent(b->own_stream, after, 0));
    b->scan_after = scan_only_after ? after : nullptr;
    // with a threshold and no limit only the selected hits travel back: skip the score rows,
    // unless the hit pool overflows (then the pass is repeated with them and ranked on the host)
    // ... and with a limit K2 / K3 select on the device (tile-level top-k where it applies): no score rows either
    const bool hits_only = (threshold > 0.0 && topk == 0) || topk > 0;
    // Small calls (a single query is the reference's own entry point, search.hpp:39-42) are
    // launch-bound: fill + K1 + K2 (+ K3) are four launches for ~15 us of work.  The second time
    // a pass of the same shape class comes along (pass_shape_class: query count, score planes, launch geometry --
    // not the exact lengths --, same parameters and buffers) it is captured
    // into a hipGraph and from then on replayed with one launch.
    bool any_streamed = false;
    for (const auto& p : ix->parts) any_streamed = any_streamed || p.streamed;
    if (nq > 0 && nq <= 16 && ix->tune.graph != 0 && !any_streamed && !ix->tune.phase_slots) {
        // the shape class of the pass and every address the captured nodes hold
        auto make_key = [&]() {
            uint64_t key = 1469598103934665603ull;
            auto mixin = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
            mixin(nq);
            mixin(pass_shape_class(b));       // not the exact lengths: one graph per shape class
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
            if (ix->tune.trace)
                std::fprintf(stderr, "[cobs_gpu] slot %d: graph %016llx replayed (t %g, k %zu, h_res %p)\n", slot,
                             (unsigned long long)key, threshold, topk, (void*)b->h_res.p);
            if (!b->graph_t0) {
                HIP_TRY(hipEventCreate(&b->graph_t0));
                HIP_TRY(hipEventCreate(&b->graph_t1));
            }
            HIP_TRY(hipEventRecord(b->graph_t0, b->own_stream));
            HIP_TRY(hipGraphLaunch(b->graph_exec, b->own_stream));
            HIP_TRY(hipEventRecord(b->graph_t1, b->own_stream));
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
                    if (ix->tune.trace)
                        std::fprintf(stderr, "[cobs_gpu] slot %d: graph %016llx captured (was candidate %016llx; t %g, k %zu, h_res %p + %zu)\n",
                                     slot, (unsigned long long)b->graph_key, (unsigned long long)key, threshold, topk,
                                     (void*)b->h_res.p, b->h_res.cap);
                    b->res_topk = res_topk;
                    b->res_pool = res_pool;
                    b->res_pool_n = pool_n;
                    b->res_rows = row_bytes_all != 0;
                    (void)hipGraphDestroy(graph);
                    if (!b->graph_t0) {
                        HIP_TRY(hipEventCreate(&b->graph_t0));
                        HIP_TRY(hipEventCreate(&b->graph_t1));
                    }
                    HIP_TRY(hipEventRecord(b->graph_t0, b->own_stream));
                    HIP_TRY(hipGraphLaunch(b->graph_exec, b->own_stream));
                    HIP_TRY(hipEventRecord(b->graph_t1, b->own_stream));
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
    b->scan_after = nullptr;
    if (st != COBS_GPU_OK) return st;
    // the pass's flag words (first invalid query, hit-pool fill) land in pinned memory right behind its kernels, `done`
    // after them: collecting the pass waits for THAT event, not for the stream -- with early_rank the ordering kernels of
    // the pass are queued on the same stream next, and cobs_gpu_batch_sync used to wait for them too before the first
    // piece could be expanded (the sharded call, which waits for the scan's event only, answered the default call of 256
    // queries in 2.88 ms against 3.17 here: profiles/r06_latency.txt)
    HIP_TRY(b->h_flags_pin.reserve(4));
    HIP_TRY(hipMemcpyAsync(b->h_flags_pin.p, b->flags.p, 16, hipMemcpyDeviceToHost, b->own_stream));
    b->flags_landing = true;
    HIP_TRY(hipEventRecord(b->done, b->own_stream));
    return COBS_GPU_OK;
}

static cobs_gpu_status host_pass_end(cobs_gpu_index* ix, int slot, double threshold, size_t topk, size_t* bad_query) {
    cobs_gpu_batch* b = ix->scratch[slot];
    cobs_gpu_status st;
    if (b->flags_landing) {
        b->flags_landing = false;
        HIP_TRY(hipEventSynchronize(b->done));
        std::memcpy(b->h_flags, b->h_flags_pin.p, sizeof b->h_flags);
        b->synced = true;
        st = COBS_GPU_OK;
        if (b->h_flags[0] != 0u) {           // K1 keeps 2^32-1 - (first query with a non-ACGT character)
            if (bad_query) *bad_query = 0xFFFFFFFFu - b->h_flags[0];
            st = fail(COBS_GPU_ERR_INVALID_BASE, "Invalid DNA base pair in query string. Only ACGT are allowed. (query " +
                                                 std::to_string(0xFFFFFFFFu - b->h_flags[0]) + ")");
        }
    } else {
        st = cobs_gpu_batch_sync(b, b->own_stream, bad_query);
    }
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
        } else if (b->graph_run && b->graph_t0 && hipEventElapsedTime(&sm, b->graph_t0, b->graph_t1) == hipSuccess) {
            ix->timers[2] += sm * 1e-3;          // the replayed pass as a whole: hashing, scan, selection, its copies home
        } else {
            (void)hipGetLastError();
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
                                         size_t* bad_query, cobs_amd::ResultArena* grow = nullptr) {
    // grow: `hits` is that arena (cobs_gpu_search_batch_view): a thresholded call whose hits outgrow it makes it larger in
    // place of reporting ERR_CAPACITY -- the caller would have to run the whole search a second time for the size
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
    // (the first quarter goes as two passes, 1/16 and 3/16 of the call: what the call waits for before its first scan
    // starts -- staging, upload and K1 of pass 0 -- is a sixteenth of the call's, and every later pass is staged under a
    // scan at least a third of its own size)
    const bool piped = !any_streamed && pipe_chars && total_chars >= pipe_chars && nq >= 64;
    static const bool split_first = !(getenv("COBS_GPU_SPLIT_FIRST") && getenv("COBS_GPU_SPLIT_FIRST")[0] == '0');     // (A/B switch)
    auto pass_cap = [&](size_t pass_index) -> size_t {
        if (!piped) return std::max<size_t>(nq, 1);
        const size_t quarter = (nq + 3) / 4;
        if (!split_first) return quarter;
        if (pass_index == 0) return std::max<size_t>(16, quarter / 4);
        if (pass_index == 1) return std::max<size_t>(16, quarter - quarter / 4);
        return quarter;
    };
    const size_t topk = num_results < ix->total_counts ? num_results : 0;   // bounded: K3 selects on the device
    // Every document of every query (the reference's default call): the ordering kernels of a pass are queued right behind
    // its scan (rank.cpp: rank_launch) -- no host round trip in between, and in a call of several passes they run, and the
    // first pieces cross PCIe, while the host still expands the pass before.  (Cutting a one-pass call in two or four so
    // that its later scans hide behind the first ordering was measured slower, EXPERIMENTS.md: the passes' kernels share
    // the device and the smaller pieces expand less efficiently.)
    const bool early_rank = ix->tune.device_rank != 0 && !(threshold > 0.0) && topk == 0 && !any_streamed;
    struct Pass { size_t g0, g1; int slot; };
    std::vector<Pass> inflight;                        // FIFO, at most `depth` entries
    auto drain = [&]() {                               // error paths: nothing may still use the scratch batches
        for (const Pass& ps : inflight) {
            (void)hipStreamSynchronize(ix->scratch[ps.slot]->own_stream);
            rank_cancel(ix->scratch[ps.slot]);
        }
        inflight.clear();
    };
    // a pass that failed while it was collected has left `inflight` already: nothing of IT may stay in flight either
    // (with early_rank its ordering kernels and D2H pieces were queued when it began) [ADVICE r5]
    auto settle = [&](const Pass& ps) {
        (void)hipStreamSynchronize(ix->scratch[ps.slot]->own_stream);
        rank_cancel(ix->scratch[ps.slot]);
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
            st = rank_on_device(sb, 0, ps.g1 - ps.g0, num_results, hits, cap, &used, hit_offsets + ps.g0, &overflow);
            ix->timers[4] += now_s() - t0;
            if (st != COBS_GPU_ERR_UNSUPPORTED) return st;          // (no room for its workspace: the host paths below)
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
        // a thresholded pass whose pool holds every hit: ordered on the device (results.cpp: order_pool), the finished
        // lists of ALL queries of the pass are copied in one sweep (round 4: one guarded call + one partial_sort per query)
        // (not with a limit: a limit too large for K3 -- k > 65536 -- runs as a thresholded pass whose lists are cut per query
        // below; until round 6 the sweep handed such a call every hit)
        if (!overflow && sb->selected && sb->topk_k == 0 && topk == 0 && !sb->graph_run && sb->h_nhits() <= sb->hit_cap) {
            const double t0 = now_s();
            if (!sb->pool_fetched) {
                st = order_pool(sb, sb->hits.p, sb->h_nhits(), sb->own_stream);
                if (st != COBS_GPU_OK) return st;
            }
            cobs_gpu_status hs = COBS_GPU_OK;
            const size_t n = sb->h_hits.size();
            if (hand_over_pool(sb, ps.g0, ps.g1, &hits, &cap, &used, hit_offsets, grow, nq, &hs)) {
                ix->timers[4] += now_s() - t0;
                if (ix->tune.trace) std::fprintf(stderr, "[cobs_gpu] pass of queries %zu..%zu: %zu hits ordered and handed over in %.3f ms\n", ps.g0, ps.g1, n, (now_s() - t0) * 1e3);
                return COBS_GPU_OK;
            }
            if (hs != COBS_GPU_OK) return hs;
            ix->timers[4] += now_s() - t0;
        }
        // a limited pass over one file whose lists K3 ordered: one sweep (results.cpp: hand_over_topk)
        if (!overflow && sb->topk_k != 0) {
            const double t0 = now_s();
            cobs_gpu_status hs = COBS_GPU_OK;
            const bool done = hand_over_topk(sb, ps.g0, ps.g1, num_results, hits, cap, &used, hit_offsets, &hs);
            ix->timers[4] += now_s() - t0;
            if (done) return COBS_GPU_OK;
            if (hs != COBS_GPU_OK) return hs;
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
        while (g1 < nq && g1 - g0 < pass_cap(pass_no)) {
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
            if (st != COBS_GPU_OK) { settle(oldest); drain(); return st; }
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
                const std::string keep = last_error_text();
                earlier = collect(ps);
                if (earlier == COBS_GPU_OK) last_error_text() = keep;
                else settle(ps);
            }
            drain();
            if (earlier != COBS_GPU_OK) return earlier;
            if (bad_query) *bad_query = first_bad;
            return st;
        }
        if (early_rank && g1 - g0 > 16 && rank_on_device_applies(ix->scratch[slot], g1 - g0))
            (void)rank_launch(ix->scratch[slot], 0, g1 - g0, num_results);      // (a failure shows when the pass is collected)
        // the next pass's scan follows this pass's SCAN (its K3 / selection kernels may run beside it: other buffers);
        // out-of-core passes share their stream buffers and follow the whole pass
        prev_done = (!any_streamed && g1 - g0 > 16 && ix->scratch[slot]->scan_end) ? ix->scratch[slot]->scan_end
                                                                                   : ix->scratch[slot]->done;
        inflight.push_back(Pass{g0, g1, slot});
        ++pass_no;
        if (nq == 0) break;
        g0 = g1;
    }
    while (!inflight.empty()) {
        const Pass ps = inflight.front();
        inflight.erase(inflight.begin());
        cobs_gpu_status st = collect(ps);
        if (st != COBS_GPU_OK) { settle(ps); drain(); return st; }
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


// The search whose results stay in memory the LIBRARY owns: a result arena kept on the handle, grown on demand, its pages
// faulted in once (by the first call that fills it) and reused by every later call -- a caller that allocates a fresh
// result array per call pays 75 000 first-touch page faults for the default call of 256 queries (5.6 ms against 3.4 ms
// into a kept array).  *hits / *hit_offsets are valid until the next search call on this handle.
cobs_gpu_status cobs_gpu_search_batch_view(cobs_gpu_index* ix, const char* const* queries, const size_t* lens, size_t nq,
                                           double threshold, size_t num_results, const cobs_gpu_hit** hits,
                                           const size_t** hit_offsets, size_t* bad_query) {
    if (!ix || !hits || !hit_offsets) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    *hits = nullptr;
    *hit_offsets = nullptr;
    return guarded([&]() -> cobs_gpu_status {
        ResultArena& ar = ix->arena;
        ar.offs.assign(nq + 1, 0);
        size_t per_query = 0;
        for (const Part& p : ix->parts) per_query += p.meta.doc_names.size();
        size_t cap;
        if (num_results > 0 && num_results < ix->total_counts) cap = nq * std::min<size_t>(num_results, per_query);
        else if (threshold <= 0.0) cap = nq * per_query;                // every document is a result
        else cap = std::max<size_t>(ar.cap, 16 * nq + 1024);            // grown on demand
        for (;;) {
            if (cobs_gpu_status rs = ar.reserve(std::max<size_t>(cap, 1)); rs != COBS_GPU_OK) return rs;
            const cobs_gpu_status st = search_batch_impl(ix, queries, lens, nq, threshold, num_results, ar.p, ar.cap,
                                                         ar.offs.data(), bad_query, &ar);
            if (st == COBS_GPU_ERR_CAPACITY && ar.offs[nq] > ar.cap) {   // hit_offsets[nq] holds the needed size
                cap = ar.offs[nq];
                continue;
            }
            if (st != COBS_GPU_OK) return st;
            break;
        }
        *hits = ar.p;
        *hit_offsets = ar.offs.data();
        return COBS_GPU_OK;
    });
}

}  // extern "C"
