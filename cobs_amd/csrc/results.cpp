// cobs_amd/csrc/results.cpp -- results of a finished pass on the host side: score rows through a pinned window,
// counts widened to u32, and counts_to_result (reference cobs/query/classic_search.cpp:109-202) from whatever
// the pass left -- K3's ordered lists, the hit pool, or whole score rows ranked by a stable counting sort
// (the device-side ranking of whole rows is rank.cpp / rank_kernels.hip).
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
cobs_gpu_status fetch_counts(cobs_gpu_batch* b, size_t q, uint32_t* counts) {
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
cobs_gpu_status rank_window(cobs_gpu_batch* b, size_t q0, size_t q1, size_t per_query, cobs_gpu_hit* hits) {
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

// The hit pool in result order, on the device (xchg_kernels.hip: count per query, scan, scatter, one wave per query
// ranks its bucket), then ONE copy home.  [Round 4 copied the pool to pageable memory, bucketed it with a counting
// scatter on one host thread and ran std::partial_sort per query: 15.7 ms for the 264 000 hits of a 10k-query pass
// (1.6 us per query, most of it per-call overhead) next to 19 ms of scan.]
// Two halves, so that a pass's pool is ordered and crosses PCIe under the scan of the next pass (host_api.cpp,
// sharded.cpp): order_pool_launch queues everything on `st` and returns; order_pool_collect waits for `st` and
// publishes the host lists.  The batch says "pool fetched / sorted" only after the collect half went through
// [ADVICE r5: the flags were set before the first fallible step, a failed ordering left an empty pool that later
// calls answered 0 hits from].
static bool order_less(const HitDev& x, const HitDev& y, bool single) {
    if (!single && x.score != y.score) return x.score > y.score;
    if (x.part != y.part) return x.part < y.part;
    return x.doc < y.doc;
}

// (a pool beyond the 32-bit indices of the device ordering: bucketed and ordered on the host, as round 4 did)
static cobs_gpu_status order_pool_host(cobs_gpu_batch* b, const HitDev* d_pool, uint64_t n, hipStream_t st) {
    const size_t nq = b->nq;
    std::vector<HitDev> raw((size_t)n);
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(raw.data(), d_pool, (size_t)n * sizeof(HitDev), hipMemcpyDeviceToHost));
    std::vector<size_t> off(nq + 1, 0);
    for (const HitDev& h : raw) if (h.query < nq) off[h.query + 1]++;
    for (size_t q = 0; q < nq; ++q) off[q + 1] += off[q];
    std::vector<HitDev> out(off[nq]);
    std::vector<size_t> cur(off.begin(), off.end() - 1);
    for (const HitDev& h : raw) if (h.query < nq) out[cur[h.query]++] = h;
    for (size_t q = 0; q < nq; ++q) {
        const bool single = total_hashes(b, q) <= 1;
        std::sort(out.begin() + off[q], out.begin() + off[q + 1], [single](const HitDev& x, const HitDev& y) { return order_less(x, y, single); });
    }
    b->h_hits.swap(out);
    b->h_hit_off.swap(off);
    b->pool_pending = false;
    b->pool_fetched = b->pool_sorted = true;
    return COBS_GPU_OK;
}

cobs_gpu_status order_pool_launch(cobs_gpu_batch* b, const HitDev* d_pool, uint64_t n, hipStream_t st) {
    const size_t nq = b->nq;
    b->pool_fetched = b->pool_sorted = b->pool_pending = false;
    b->pool_n = 0;
    if (nq == 0 || n == 0) { b->pool_pending = true; return COBS_GPU_OK; }
    if (n > 0x7FFFFFFFull) return COBS_GPU_ERR_UNSUPPORTED;        // (order_pool: the host path)
    HIP_TRY(hipSetDevice(b->ix->device));
    HIP_TRY(b->pool_idx.reserve(3 * (nq + 1)));
    HIP_TRY(b->pool_tmp.reserve((size_t)n));
    HIP_TRY(b->pool_out.reserve((size_t)n));
    const size_t off_bytes = round_up((nq + 1) * sizeof(uint32_t), 16);
    HIP_TRY(b->h_pool.reserve(off_bytes + (size_t)n * sizeof(HitDev)));
    bool any_single = false;
    for (size_t q = 0; q < nq && !any_single; ++q) any_single = total_hashes(b, q) <= 1;
    PoolArgs a{};
    a.in = d_pool;
    a.tmp = b->pool_tmp.p;
    a.out = b->pool_out.p;
    a.cnt = b->pool_idx.p;
    a.off = b->pool_idx.p + (nq + 1);
    a.cur = b->pool_idx.p + 2 * (nq + 1);
    a.single = nullptr;
    a.n = (uint32_t)n;
    a.nq = (uint32_t)nq;
    a.seg_max = kPoolSegMax;
    if (any_single) {
        HIP_TRY(b->pool_single.reserve(nq));
        HIP_TRY(b->h_single.reserve(nq));               // (pinned: the copy leaves when the stream gets to it)
        for (size_t q = 0; q < nq; ++q) b->h_single.p[q] = total_hashes(b, q) <= 1 ? 1 : 0;
        HIP_TRY(hipMemcpyAsync(b->pool_single.p, b->h_single.p, nq, hipMemcpyHostToDevice, st));
        a.single = b->pool_single.p;
    }
    HIP_TRY(hipMemsetAsync(b->pool_idx.p, 0, 3 * (nq + 1) * sizeof(uint32_t), st));
    HIP_TRY(launch_order_pool(a, st));
    HIP_TRY(hipMemcpyAsync(b->h_pool.p, a.off, (nq + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b->h_pool.p + off_bytes, a.out, (size_t)n * sizeof(HitDev), hipMemcpyDeviceToHost, st));
    b->pool_n = n;
    b->pool_pending = true;
    return COBS_GPU_OK;
}

cobs_gpu_status order_pool_collect(cobs_gpu_batch* b, hipStream_t st, bool waited) {
    if (!b->pool_pending) return fail(COBS_GPU_ERR_ARG, "no hit pool is being ordered");
    b->pool_pending = false;
    const size_t nq = b->nq;
    const uint64_t n = b->pool_n;
    if (nq == 0 || n == 0) {
        b->h_hit_off.assign(nq + 1, 0);
        b->h_hits.clear();
        b->pool_fetched = b->pool_sorted = true;
        return COBS_GPU_OK;
    }
    HIP_TRY(hipSetDevice(b->ix->device));
    if (!waited) HIP_TRY(hipStreamSynchronize(st));
    const size_t off_bytes = round_up((nq + 1) * sizeof(uint32_t), 16);
    const uint32_t* off = reinterpret_cast<const uint32_t*>(b->h_pool.p);
    const HitDev* rec = reinterpret_cast<const HitDev*>(b->h_pool.p + off_bytes);
    b->h_hit_off.resize(nq + 1);
    for (size_t q = 0; q <= nq; ++q) b->h_hit_off[q] = off[q];
    const size_t kept = off[nq];                    // (records whose query number is out of range are dropped, as before)
    b->h_hits.assign(rec, rec + kept);
    for (size_t q = 0; q < nq; ++q) {               // buckets too large for the device's one-wave ordering
        const size_t s0 = off[q], s1 = off[q + 1];
        if (s1 - s0 <= kPoolSegMax) continue;
        const bool single = total_hashes(b, q) <= 1;
        std::sort(b->h_hits.begin() + s0, b->h_hits.begin() + s1, [single](const HitDev& x, const HitDev& y) { return order_less(x, y, single); });
    }
    b->pool_fetched = b->pool_sorted = true;
    return COBS_GPU_OK;
}

cobs_gpu_status order_pool(cobs_gpu_batch* b, const HitDev* d_pool, uint64_t n, hipStream_t st) {
    const bool trace = b->ix->tune.trace;
    const double tr0 = trace ? now_s() : 0.0;
    cobs_gpu_status s = order_pool_launch(b, d_pool, n, st);
    if (s == COBS_GPU_ERR_UNSUPPORTED) return order_pool_host(b, d_pool, n, st);
    if (s == COBS_GPU_OK) s = order_pool_collect(b, st);
    if (s != COBS_GPU_OK) b->pool_fetched = b->pool_sorted = b->pool_pending = false;
    if (trace)
        std::fprintf(stderr, "[cobs_gpu] hit pool of %llu records ordered in %.3f ms\n", (unsigned long long)n, (now_s() - tr0) * 1e3);
    return s;
}

// The finished lists of ALL queries of a pass whose pool is in result order (order_pool), handed over in one sweep:
// query `i` of the pass is query g0 + i of the call; its records go behind *used, hit_offsets[g0 + i + 1] = the new
// *used.  grow: `hits` is that arena (cobs_gpu_search_batch_view) -- made larger in place of reporting an overflow,
// for this pass and, at the rate so far, for the passes to come (nq_call = queries of the whole call).
// Returns false (nothing written) when the lists do not fit: the caller counts them query by query.
bool hand_over_pool(cobs_gpu_batch* sb, size_t g0, size_t g1, cobs_gpu_hit** hits, size_t* cap, size_t* used, size_t* hit_offsets,
                    ResultArena* grow, size_t nq_call, cobs_gpu_status* status) {
    *status = COBS_GPU_OK;
    if (!sb->pool_sorted) return false;
    if (grow && sb->h_hits.size() > *cap - *used) {
        const size_t have = *used + sb->h_hits.size();
        const size_t want = std::max<size_t>(have + have / 4 + 1024, (size_t)((double)have * (double)nq_call / (double)std::max<size_t>(g1, 1) * 1.125));
        if (cobs_gpu_status gs = grow->grow_keep(want, *used); gs != COBS_GPU_OK) { *status = gs; return false; }
        *hits = grow->p;
        *cap = grow->cap;
    }
    if (sb->h_hits.size() > *cap - *used || (!sb->h_hits.empty() && !*hits)) return false;
    const HitDev* rec = sb->h_hits.data();
    const size_t n = sb->h_hits.size();
    cobs_gpu_hit* dst = *hits + *used;
    for (size_t i = 0; i < n; ++i) dst[i] = cobs_gpu_hit{rec[i].part, rec[i].doc, rec[i].score};
    for (size_t q = g0; q < g1; ++q) hit_offsets[q + 1] = *used + sb->h_hit_off[q - g0 + 1];
    *used += n;
    return true;
}

// K3's survivors of every (file, query) of the last run on the host, once per run (a replayed graph brought them home
// itself; an exchange of the shards' lists has left them already: topk_fetched)
static cobs_gpu_status fetch_topk(cobs_gpu_batch* b) {
    if (b->topk_fetched) return COBS_GPU_OK;
    const size_t k = b->topk_k, nparts = b->ix->parts.size();
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
    return COBS_GPU_OK;
}

// A limited pass over ONE index file whose lists K3 left in result order: all queries handed over in one sweep (the
// per-query call costs ~75 ns of checks and copies per query: 0.7-0.9 ms for 10 000 queries, a quarter of what a rank of
// an 8-GPU node scans for).  false (nothing written): not that case -- several files, a query with a single hash in
// total (index order: from rows), lists beyond the device's sort limit, a buffer that is too small -- the caller goes
// query by query.
bool hand_over_topk(cobs_gpu_batch* b, size_t g0, size_t g1, size_t num_results, cobs_gpu_hit* hits, size_t cap, size_t* used,
                    size_t* hit_offsets, cobs_gpu_status* status) {
    *status = COBS_GPU_OK;
    const cobs_gpu_index* ix = b->ix;
    const size_t k = b->topk_k, nq = g1 - g0;
    if (ix->parts.size() != 1 || k == 0 || !b->topk_sorted || num_results == 0 || num_results > k || nq != b->nq) return false;
    for (size_t q = 0; q < nq; ++q)
        if (total_hashes(b, q) <= 1) return false;
    if ((*status = fetch_topk(b)) != COBS_GPU_OK) return false;
    if (b->topk_stride) return false;                   // (the shards' lists side by side: merged per query)
    const size_t lim = std::min<size_t>(num_results, (size_t)ix->total_counts);
    size_t total = 0;
    for (size_t q = 0; q < nq; ++q) total += std::min<size_t>(lim, b->h_topk_cnt[q]);
    if (total > cap - *used || (total && !hits)) return false;
    cobs_gpu_hit* dst = hits + *used;
    for (size_t q = 0; q < nq; ++q) {
        const uint2* e = b->h_topk.data() + q * k;
        const size_t n = std::min<size_t>(lim, b->h_topk_cnt[q]);
        for (size_t i = 0; i < n; ++i) dst[i] = cobs_gpu_hit{0u, e[i].x, e[i].y};
        dst += n;
        *used += n;
        hit_offsets[g0 + q + 1] = *used;
    }
    return true;
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
        if (cobs_gpu_status fs = fetch_topk(b); fs != COBS_GPU_OK) return fs;
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
        if (b->pool_owned && (q < b->own_q0 || q >= b->own_q0 + b->own_qn))
            return fail(COBS_GPU_ERR_ARG, "this rank does not hold the exchanged hits of that query");
        if (!b->pool_fetched && !(b->graph_run && b->h_res.p && b->h_nhits() <= b->res_pool_n)) {
            // the pool is in the order the scan's atomics appended it: put into result order on the device, one copy home
            cobs_gpu_status os = order_pool(b, b->hits.p, b->h_nhits(), b->own_stream);
            if (os != COBS_GPU_OK) return os;
        }
        if (b->pool_sorted) {
            // finished lists: a limit is a prefix
            const size_t s0 = b->h_hit_off[q], s1 = b->h_hit_off[q + 1];
            size_t want = num_results == 0 ? (size_t)ix->total_counts : std::min<size_t>(num_results, (size_t)ix->total_counts);
            want = std::min(want, s1 - s0);
            *n_hits = want;
            if (want > cap) return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small");
            if (want && !hits) return fail(COBS_GPU_ERR_ARG, "NULL hit buffer");
            for (size_t i = 0; i < want; ++i) hits[i] = cobs_gpu_hit{b->h_hits[s0 + i].part, b->h_hits[s0 + i].doc, b->h_hits[s0 + i].score};
            return COBS_GPU_OK;
        }
        if (!b->pool_fetched) {
            // (a replayed graph brought the pool home by itself: a few records, bucketed here)
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

}  // namespace cobs_amd

extern "C" {

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

cobs_gpu_status cobs_gpu_batch_counts_host(cobs_gpu_batch* b, size_t q, uint32_t* counts, size_t cap) {
    if (!b || !counts) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (!b->ran || !b->synced) return fail(COBS_GPU_ERR_ARG, "run and sync the batch first");
    if (q >= b->nq) return fail(COBS_GPU_ERR_ARG, "query number out of range");
    if (cap < b->ix->total_counts) return fail(COBS_GPU_ERR_CAPACITY, "counts buffer too small");
    HIP_TRY(hipSetDevice(b->ix->device));
    return fetch_counts(b, q, counts);
}

// `cobs benchmark-fpr --dist` (reference src/cobs.cpp:627-632, 664-670): the distribution of the scores of all results.
// The reference walks every result of every query on the host (10 000 queries x 100 000 documents = 10^9 map updates);
// here the score rows of the last run are tallied where they lie.
cobs_gpu_status cobs_gpu_batch_score_histogram(cobs_gpu_batch* b, uint64_t* hist, size_t nbins) {
    if (!b || !hist || nbins == 0 || nbins > (1u << 26)) return fail(COBS_GPU_ERR_ARG, "bad argument");
    if (!b->ran || !b->have_counts || b->view_global) return fail(COBS_GPU_ERR_ARG, "run the batch with score rows first");
    return guarded([&]() -> cobs_gpu_status {
        const cobs_gpu_index* ix = b->ix;
        HIP_TRY(hipSetDevice(ix->device));
        std::vector<HistRange> ranges;
        for (const Part& p : ix->parts) {
            // the slots this shard holds that belong to real documents (padding documents of the last sub-index never score)
            const uint64_t docs = p.meta.doc_names.size();
            const uint64_t real = docs > p.slot_begin ? std::min<uint64_t>(docs - p.slot_begin, p.slot_count) : 0;
            if (real) ranges.push_back(HistRange{(uint32_t)p.local_offset, (uint32_t)(p.local_offset + real)});
        }
        DevBuf<HistRange> d_ranges;
        DevBuf<unsigned long long> d_hist;
        HIP_TRY(d_ranges.reserve(std::max<size_t>(ranges.size(), 1)));
        HIP_TRY(d_hist.reserve(nbins));
        HIP_TRY(hipMemcpy(d_ranges.p, ranges.data(), ranges.size() * sizeof(HistRange), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_hist.p, 0, nbins * sizeof(unsigned long long)));
        HIP_TRY(hipEventSynchronize(b->run_done));
        HistArgs a{};
        a.rows = b->counts.p;
        a.row_stride = ix->local_counts;
        a.ranges = d_ranges.p;
        a.hist = d_hist.p;
        a.nranges = (uint32_t)ranges.size();
        a.nq = (uint32_t)b->nq;
        a.nbins = (uint32_t)nbins;
        a.score_bytes = b->elem_bytes;
        HIP_TRY(launch_score_hist(a, nullptr));
        std::vector<unsigned long long> h(nbins);
        HIP_TRY(hipMemcpy(h.data(), d_hist.p, nbins * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nbins; ++i) hist[i] += h[i];
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_batch_hits_host(cobs_gpu_batch* b, size_t q, size_t num_results,
                                         cobs_gpu_hit* hits, size_t cap, size_t* n_hits) {
    return guarded([&]() { return hits_host_impl(b, q, num_results, hits, cap, n_hits); });
}

}  // extern "C"
