// cobs_amd/csrc/sharded.cpp -- the search over an index sharded across GPUs, as ONE call and as one pipeline:
//   * cobs_gpu_sharded_search_batch[_split]: ClassicSearch::search (reference cobs/query/classic_search.cpp:403-505)
//     for many queries over the shards of a communicator.  The reference parallelises INSIDE search() -- parallel_for
//     over document batches, classic_search.cpp:355-400 -- so every caller gets it; here the call is cut into passes
//     that overlap on three kinds of streams: upload + K1 of pass i+1 | K2 of pass i | the ranks' agreement, the
//     exchange over RCCL / xGMI and the ordering of the results of pass i-1, tied by events only.  What the ranks have
//     to agree on after a scan -- did it go through everywhere, the first invalid query, every shard's hit-pool fill
//     -- travels as ONE all-gathered record per pass that the device writes (no host round trip before it), read once
//     per pass while the next pass scans.  [Until round 6 a pass was set_queries -> scan -> host wait -> up to three
//     host-synchronous all-reduces -> exchange -> host wait -> ranking on ONE stream; the overlapped form lived in
//     bench.py only: VERDICT r5.]
//   * cobs_gpu_sharded_batch: the device-resident form (queries uploaded once, count rows left on the device in global
//     document order): sub-batches whose hashing, scan and exchange overlap -- what bench.py --gpus N times.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "comm.hpp"

using namespace cobs_amd;

namespace {

constexpr size_t kMetaWords = 4;     // status | first invalid query word | hit-pool fill | carried status of earlier passes

// This rank's record of a pass -> every rank's, on the exchange stream: the device packs it from the batch's flag words
// (xchg_kernels.hip: pass_meta_kernel), one ncclAllGather of 32 bytes per rank, the records land in pinned memory, and
// x.pass_ev says when.  flags_valid = false: the pass never launched on this rank (its host-side status says why).
cobs_gpu_status meta_launch(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t xs, bool flags_valid, uint32_t status, uint32_t carried) {
    if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
    if (!b->xchg) b->xchg = new Exchange;
    Exchange& x = *b->xchg;
    const size_t N = (size_t)c->nranks;
    HIP_TRY(x.d_pass.reserve(kMetaWords * (N + 1)));
    HIP_TRY(x.h_pass.reserve(kMetaWords * N));
    if (!x.pass_ev) HIP_TRY(hipEventCreateWithFlags(&x.pass_ev, hipEventDisableTiming));
    if (!x.x_done) HIP_TRY(hipEventCreateWithFlags(&x.x_done, hipEventDisableTiming));
    HIP_TRY(launch_pass_meta(flags_valid ? b->flags.p : nullptr, x.d_pass.p, status, carried, xs));
    NCCL_C(c, xs, ncclAllGather(x.d_pass.p, x.d_pass.p + kMetaWords, kMetaWords * 8, ncclUint8, c->comm, xs));
    HIP_TRY(hipMemcpyAsync(x.h_pass.p, x.d_pass.p + kMetaWords, kMetaWords * 8 * N, hipMemcpyDeviceToHost, xs));
    HIP_TRY(hipEventRecord(x.pass_ev, xs));
    return COBS_GPU_OK;
}

struct Agreed {
    uint32_t worst = 0;              // max over ranks of the host-side status and of what earlier passes carried
    uint32_t bad_word = 0;           // max over ranks of K1's word: 2^32-1 - (first invalid query of the pass), 0 = none
    std::vector<uint64_t> fills;     // [N] hit-pool fills
};

cobs_gpu_status meta_wait(cobs_gpu_batch* b, cobs_gpu_comm* c, Agreed* a) {
    Exchange& x = *b->xchg;
    if (cobs_gpu_status ws = event_bounded(c, x.pass_ev, "the ranks' agreement on a pass (all-gather of the status records)"); ws != COBS_GPU_OK)
        return ws;
    const size_t N = (size_t)c->nranks, me = (size_t)c->rank;
    a->fills.assign(N, 0);
    a->worst = 0;
    a->bad_word = 0;
    for (size_t r = 0; r < N; ++r) {
        const uint64_t* rec = x.h_pass.p + r * kMetaWords;
        a->worst = std::max(a->worst, std::max((uint32_t)rec[0], (uint32_t)rec[3]));
        a->bad_word = std::max(a->bad_word, (uint32_t)rec[1]);
        a->fills[r] = rec[2];
    }
    const uint64_t* mine = x.h_pass.p + me * kMetaWords;
    b->h_flags[0] = (uint32_t)mine[1];
    b->h_flags[1] = 0;
    b->h_flags[2] = (uint32_t)mine[2];
    b->h_flags[3] = (uint32_t)(mine[2] >> 32);
    return COBS_GPU_OK;
}

cobs_gpu_status make_scratch(cobs_gpu_index* ix, int slot) {
    if (ix->scratch[slot]) return COBS_GPU_OK;
    cobs_gpu_status cs = cobs_gpu_batch_create(ix, 0, 0, &ix->scratch[slot]);
    if (cs != COBS_GPU_OK) return cs;
    HIP_TRY(hipStreamCreateWithFlags(&ix->scratch[slot]->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ix->scratch[slot]->done, hipEventDisableTiming));
    return COBS_GPU_OK;
}

// ClassicSearch::search over the sharded index: every rank calls this with the same queries and
// gets the same, global, result (hits ordered as cobs_gpu_search_batch orders them).
// split (cobs_gpu_sharded_search_batch_split): for the all-documents call (threshold <= 0, no limit) the ranks SHARE
// the ranking instead of repeating it -- the count rows go all-to-all to query owners, rank j orders the queries
// [n*j/N, n*(j+1)/N) of every pass and writes their results (and offsets) at their final places of the caller's arrays.
cobs_gpu_status sharded_search_impl(cobs_gpu_index* ix, cobs_gpu_comm* c, const char* const* queries, const size_t* lens,
                                    size_t nq, double threshold, size_t num_results, cobs_gpu_hit* hits, size_t cap,
                                    size_t* hit_offsets, size_t* bad_query, bool split) {
    if (!ix || !c || !hit_offsets) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (nq && (!queries || !lens)) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        HIP_TRY(hipSetDevice(ix->device));
        if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
        if (!ix->xchg_stream) HIP_TRY(hipStreamCreateWithFlags(&ix->xchg_stream, hipStreamNonBlocking));
        hipStream_t xs = ix->xchg_stream;
        const size_t N = (size_t)c->nranks;
        // Passes bounded like the single-GPU API (score rows / tables below the workspace limit; a call with a lot of
        // query text in at least four, so that there is something to overlap) -- from quantities that are the SAME on
        // every rank (whole-file geometry, not this shard's; the tuning keys, which every rank must set alike): all
        // ranks must cut the batch at the same places, every pass is a set of collectives.
        uint32_t min_term = 0xFFFFFFFFu;
        uint64_t table_per_char = 0;
        bool any_streamed = false;
        for (const auto& p : ix->parts) {
            min_term = std::min(min_term, p.meta.term_size);
            table_per_char += 4ull * p.meta.num_hashes * std::max<uint32_t>(p.meta.num_pages(), 1) * (p.idx64 ? 2 : 1);
            any_streamed = any_streamed || p.streamed;
        }
        const size_t topk = num_results < ix->total_counts ? num_results : 0;
        const bool all_docs = threshold <= 0.0 && topk == 0;
        const bool hits_only = threshold > 0.0 && topk == 0;
        // shared ranking: every query yields one result per real document, so every result's place is known up front
        const bool shared = split && all_docs;
        if (!shared || c->rank == 0) hit_offsets[0] = 0;          // (shared arrays: every entry has exactly one writer)
        size_t per_query = 0;
        for (const auto& p : ix->parts) per_query += p.meta.doc_names.size();
        if (shared) {
            if (cap < nq * per_query || (nq * per_query && !hits)) {      // the same on every rank: nobody enters a collective
                if (c->rank == 0)           // (ranks of one process share the array: one writer)
                    for (size_t q = 0; q < nq; ++q) hit_offsets[q + 1] = (q + 1) * per_query;
                return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small; hit_offsets[nq] holds the needed size");
            }
        }
        uint64_t total_chars = 0;
        for (size_t q = 0; q < nq; ++q) total_chars += lens[q];
        // (the first quarter as two passes, 1 / 16 and 3 / 16 of the call, as the one-GPU call cuts: host_api.cpp)
        const bool piped = ix->tune.pipe_chars && total_chars >= ix->tune.pipe_chars && nq >= 64;
        static const bool split_first = !(getenv("COBS_GPU_SPLIT_FIRST") && getenv("COBS_GPU_SPLIT_FIRST")[0] == '0');     // (A/B switch)
        auto pass_cap = [&](size_t pass_index) -> size_t {
            if (!piped) return std::max<size_t>(nq, 1);
            const size_t quarter = (nq + 3) / 4;
            if (!split_first) return quarter;
            if (pass_index == 0) return std::max<size_t>(16, quarter / 4);
            if (pass_index == 1) return std::max<size_t>(16, quarter - quarter / 4);
            return quarter;
        };
        struct Pass {
            size_t g0 = 0, g1 = 0;
            int slot = 0;
            bool launched = false;           // K1 / K2 of the pass are queued on this rank
            cobs_gpu_status local = COBS_GPU_OK;
            std::string local_msg;
            size_t bad = 0;
            bool need_rows = false;          // stage A decided: the results come from exchanged score rows
            bool pool = false;               // ... from the gathered hit pools (ordered on the device)
            bool pool_host = false;          // ... too many records for the device ordering: ordered when collected
            bool ranked = false;             // ... rank_launch queued the ordering of the rows
        };
        std::vector<Pass> passes;
        for (size_t g0 = 0; g0 < nq || passes.empty();) {
            size_t g1 = g0;
            uint64_t tb = 0, max_terms = 1;
            while (g1 < nq && g1 - g0 < pass_cap(passes.size())) {
                const uint64_t terms = lens[g1] >= min_term ? lens[g1] - min_term + 1 : 1;
                const uint64_t mt = std::max(max_terms, terms);
                const int planes = scan_planes_for(mt);
                const uint64_t eb = planes > 0 ? scan_score_bytes(planes) : 4u;
                // local rows (bounded by the whole vector), plus the assembled global rows where the pass exchanges rows
                const uint64_t sb = (uint64_t)(g1 - g0 + 1) * (ix->total_counts + (all_docs ? ix->total_counts : 0)) * eb;
                const uint64_t t = (uint64_t)(lens[g1] + 16) * table_per_char;
                if (g1 > g0 && (sb > ix->tune.pass_bytes || tb + t > ix->tune.pass_bytes)) break;
                max_terms = mt;
                tb += t;
                ++g1;
            }
            Pass ps;
            ps.g0 = g0;
            ps.g1 = g1;
            ps.slot = (int)(passes.size() % cobs_gpu_index::kScratch);
            passes.push_back(ps);
            if (nq == 0) break;
            g0 = g1;
        }
        const size_t np = passes.size();
        size_t used = 0;
        bool overflow = false;
        uint32_t carried = 0;                // the first failure this rank met AFTER a pass's agreement (exchange, results): travels with the next record
        std::string carried_msg;
        hipEvent_t prev = nullptr;           // what the next pass's scan follows
        auto carry = [&](cobs_gpu_status s) {
            if (s != COBS_GPU_OK && carried == 0) { carried = (uint32_t)s; carried_msg = last_error_text(); }
        };
        // error paths: nothing of this call may still be in flight when it returns
        auto drain = [&]() {
            (void)hipStreamSynchronize(xs);
            for (auto* b : ix->scratch)
                if (b) { (void)hipStreamSynchronize(b->own_stream); rank_cancel(b); b->pool_pending = false; }
        };

        // ---- upload, K1, K2 (K3) of pass i on its slot's stream
        auto launch = [&](Pass& ps) {
            ps.local = make_scratch(ix, ps.slot);
            if (ps.local != COBS_GPU_OK) { ps.local_msg = last_error_text(); return; }
            cobs_gpu_batch* b = ix->scratch[ps.slot];
            const size_t n = ps.g1 - ps.g0;
            ix->host_passes++;
            const double t0 = now_s();
            cobs_gpu_status s = set_queries_on(b, queries + ps.g0, lens + ps.g0, n, b->own_stream, false, &ps.bad, ps.g0);
            ix->timers[1] += now_s() - t0;
            if (s == COBS_GPU_OK) {
                // the previous pass: a large pass of a resident shard only keeps its SCAN behind it (upload and K1 of this
                // pass run beside that scan); an out-of-core shard shares its stream buffers and follows the whole pass
                const bool scan_only = prev && n > 16 && !any_streamed;
                hipError_t e = hipSuccess;
                if (prev && !scan_only) e = hipStreamWaitEvent(b->own_stream, prev, 0);
                b->scan_after = scan_only ? prev : nullptr;
                // no score rows for a pass that selects on the device: hits into the pool, or -- with a limit -- the k best
                // of every tile (run_impl keeps the rows anyway where that does not apply, e.g. a query with a single hash
                // in total; b->have_counts says which)
                s = e != hipSuccess ? hip_fail(e, "hipStreamWaitEvent") : run_impl(b, threshold, topk, b->own_stream, !(hits_only || topk > 0));
                b->scan_after = nullptr;
                if (s == COBS_GPU_OK && (e = hipEventRecord(b->done, b->own_stream)) != hipSuccess) s = hip_fail(e, "hipEventRecord");
                if (s == COBS_GPU_OK) prev = (!any_streamed && n > 16 && b->scan_end) ? b->scan_end : b->done;
            }
            ps.local = s;
            ps.launched = s == COBS_GPU_OK;
            if (s != COBS_GPU_OK) ps.local_msg = last_error_text();
        };
        // ---- this rank's record of pass i into the all-gather, behind its scan
        auto agree_launch = [&](Pass& ps) -> cobs_gpu_status {
            cobs_gpu_batch* b = ix->scratch[ps.slot];
            if (!b) return fail(ps.local != COBS_GPU_OK ? ps.local : COBS_GPU_ERR_HIP, ps.local_msg);     // (no workspace at all: nothing to send from)
            if (ps.launched) HIP_TRY(hipStreamWaitEvent(xs, b->done, 0));
            return meta_launch(b, c, xs, ps.launched, (uint32_t)ps.local, carried);
        };
        // ---- stage A of pass i: read the agreement, queue the exchange and the ordering of the results behind it
        auto exchange_rows = [&](Pass& ps, cobs_gpu_batch* b) -> cobs_gpu_status {
            Exchange& x = *b->xchg;
            // each count row to the rank that owns its query (shared ranking), or every slice to every rank (the contract of
            // the plain call: every rank returns every query)
            cobs_gpu_status s = cobs_gpu_batch_exchange_counts(b, c, shared ? COBS_GPU_XCHG_ALLTOALL : COBS_GPU_XCHG_ALLGATHER, xs);
            if (s != COBS_GPU_OK) return s;
            HIP_TRY(hipEventRecord(x.x_done, xs));
            HIP_TRY(hipStreamWaitEvent(b->own_stream, x.x_done, 0));        // (the ranking runs on the batch's own stream)
            const size_t q0 = shared ? (size_t)b->g_q0 : 0, qn = shared ? (size_t)b->g_qn : ps.g1 - ps.g0;
            if (b->topk_k == 0 && qn && ix->tune.device_rank != 0 && rank_on_device_applies(b, qn))
                ps.ranked = rank_launch(b, q0, qn, shared ? 0 : num_results) == COBS_GPU_OK;       // (a failure: ranked when collected)
            return COBS_GPU_OK;
        };
        auto stage_a = [&](Pass& ps) -> cobs_gpu_status {
            cobs_gpu_batch* b = ix->scratch[ps.slot];
            Agreed ag;
            cobs_gpu_status s = meta_wait(b, c, &ag);
            if (s != COBS_GPU_OK) return s;
            // Every rank knows now whether the scan went through EVERYWHERE, before anybody enters the exchange: a rank that
            // failed alone (out of memory, ...) would leave the others waiting in a collective.  Bad input fails identically on
            // every rank that HASHES the queries -- a rank whose shard is empty (more ranks than sub-index blocks) does not, so
            // the records also carry K1's first invalid query and all ranks report the lowest.
            if (ag.worst != COBS_GPU_OK) {
                if (ps.local != COBS_GPU_OK) {
                    if (bad_query) *bad_query = ps.g0 + ps.bad;
                    return fail(ps.local, ps.local_msg);
                }
                if (carried) return fail((cobs_gpu_status)carried, carried_msg);
                return fail(COBS_GPU_ERR_RCCL, "the pass failed on another rank (status " + std::to_string(ag.worst) + ")");
            }
            if (ag.bad_word != 0u) {
                const size_t bad = ps.g0 + (size_t)(0xFFFFFFFFu - ag.bad_word);
                if (bad_query) *bad_query = bad;
                return fail(COBS_GPU_ERR_INVALID_BASE, "Invalid DNA base pair in query string. Only ACGT are allowed. (query " +
                                                       std::to_string(bad) + ")");
            }
            b->synced = true;                    // (the flag words are home: what cobs_gpu_batch_sync would have fetched)
            Exchange& x = *b->xchg;
            if (b->topk_k) {
                s = xchg_topk_launch(b, c, xs);
                if (s != COBS_GPU_OK) return s;
                HIP_TRY(hipEventRecord(x.x_done, xs));
                // a query with a single hash in total is NOT ordered by score (max_counts <= 1, classic_search.cpp:134,177):
                // its result is the first documents in index order, which K3's per-shard best-of lists do not determine --
                // such a pass also needs the rows (run_impl kept them: the same decision, from the lengths, on every rank)
                for (size_t q = 0; q < ps.g1 - ps.g0 && !ps.need_rows; ++q) ps.need_rows = total_hashes(b, q) <= 1;
                if (ps.need_rows) return exchange_rows(ps, b);
                return COBS_GPU_OK;
            }
            if (b->selected) {
                bool over = false;
                for (size_t r = 0; r < N; ++r) over = over || ag.fills[r] > b->hit_cap;     // the capacity is a function of the batch: equal on all ranks
                if (!over) {
                    const HitDev* pool = nullptr;
                    uint64_t total = 0;
                    s = xchg_hits_launch(b, c, xs, ag.fills.data(), &pool, &total);
                    if (s != COBS_GPU_OK) return s;
                    s = order_pool_launch(b, pool, total, xs);
                    if (s == COBS_GPU_ERR_UNSUPPORTED) { ps.pool_host = true; b->pool_n = total; s = COBS_GPU_OK; }
                    if (s != COBS_GPU_OK) return s;
                    HIP_TRY(hipEventRecord(x.x_done, xs));
                    ps.pool = true;
                    return COBS_GPU_OK;
                }
                // some shard selected more hits than its pool holds: the pass again, with score rows, on every rank.  The
                // repeat may fail on one rank alone (its score rows did not fit, ...): agree once more before the row
                // exchange, or the others wait in that collective for ever.  (Rare and not pipelined.)
                s = run_impl(b, threshold, topk, b->own_stream, true);
                if (s == COBS_GPU_OK) HIP_TRY(hipEventRecord(b->done, b->own_stream));
                const std::string keep = s != COBS_GPU_OK ? last_error_text() : std::string();
                if (s == COBS_GPU_OK) HIP_TRY(hipStreamWaitEvent(xs, b->done, 0));
                cobs_gpu_status ms = meta_launch(b, c, xs, s == COBS_GPU_OK, (uint32_t)s, carried);
                if (ms == COBS_GPU_OK) ms = meta_wait(b, c, &ag);
                if (ms != COBS_GPU_OK) return ms;
                if (s != COBS_GPU_OK) return fail(s, keep);
                if (ag.worst != COBS_GPU_OK)
                    return fail(COBS_GPU_ERR_RCCL, "the pass failed on another rank (status " + std::to_string(ag.worst) + ")");
                b->selected = false;
                ps.need_rows = true;
                return exchange_rows(ps, b);
            }
            ps.need_rows = true;
            return exchange_rows(ps, b);
        };
        // ---- stage B of pass i: wait for its results, hand them to the caller
        auto stage_b = [&](Pass& ps) -> cobs_gpu_status {
            cobs_gpu_batch* b = ix->scratch[ps.slot];
            Exchange& x = *b->xchg;
            const size_t n = ps.g1 - ps.g0;
            cobs_gpu_status s = COBS_GPU_OK;
            {   // (the pass's device times for cobs_gpu_timers; its events are long through)
                float sm = 0, hm = 0;
                if (event_bounded(c, b->done, "the scan of a pass") == COBS_GPU_OK && cobs_gpu_batch_kernel_ms(b, &sm, &hm) == COBS_GPU_OK) {
                    ix->timers[0] += hm * 1e-3;
                    ix->timers[2] += sm * 1e-3;
                }
            }
            const double t0 = now_s();
            struct Book { cobs_gpu_index* ix; double t0; ~Book() { ix->timers[4] += now_s() - t0; } } book{ix, t0};
            if (ps.pool) {
                if ((s = event_bounded(c, x.x_done, "the exchange of the hit records")) != COBS_GPU_OK) return s;
                if (ps.pool_host) {
                    s = order_pool(b, N == 1 ? b->hits.p : x.hits_all.p, b->pool_n, xs);
                } else {
                    s = order_pool_collect(b, nullptr, true);
                }
                if (s != COBS_GPU_OK) return s;
                b->pool_global = true;
                cobs_gpu_status hs = COBS_GPU_OK;
                // (one sweep for the whole pass -- unless the call has a limit, too large for K3, that cuts every list: below)
                if (!overflow && topk == 0 && hand_over_pool(b, ps.g0, ps.g1, &hits, &cap, &used, hit_offsets, nullptr, nq, &hs)) return COBS_GPU_OK;
                if (hs != COBS_GPU_OK) return hs;
            } else if (b->topk_k) {
                if ((s = event_bounded(c, x.x_done, "the all-gather of the shards' best-of lists")) != COBS_GPU_OK) return s;
                if ((s = xchg_topk_collect(b, c, xs, true)) != COBS_GPU_OK) return s;
                cobs_gpu_status hs = COBS_GPU_OK;
                if (!overflow && !ps.need_rows && hand_over_topk(b, ps.g0, ps.g1, num_results, hits, cap, &used, hit_offsets, &hs)) return COBS_GPU_OK;
                if (hs != COBS_GPU_OK) return hs;
            }
            if (ps.need_rows && shared) {
                // the rank that owns a query orders it and writes the results where they belong: the ranking and its PCIe
                // traffic are divided by the number of GPUs
                const size_t q0 = (size_t)b->g_q0, qn = (size_t)b->g_qn;         // owned queries of this pass
                size_t u = (ps.g0 + q0) * per_query;
                bool ovf = false;
                cobs_gpu_status rs = COBS_GPU_ERR_UNSUPPORTED;
                if (qn && ix->tune.device_rank != 0 && rank_on_device_applies(b, qn))
                    rs = rank_on_device(b, q0, qn, 0, hits, cap, &u, hit_offsets + ps.g0 + q0, &ovf);
                if (rs == COBS_GPU_ERR_UNSUPPORTED) {
                    rs = event_bounded(c, x.x_done, "the all-to-all of the count rows");
                    u = (ps.g0 + q0) * per_query;
                    for (size_t q = q0; q < q0 + qn && rs == COBS_GPU_OK; ++q) {
                        size_t m = 0;
                        rs = cobs_gpu_batch_hits_host(b, q, 0, hits + u, cap - u, &m);
                        u += m;
                        hit_offsets[ps.g0 + q + 1] = u;
                    }
                }
                if (rs == COBS_GPU_OK && (ovf || u != (ps.g0 + q0 + qn) * per_query))
                    rs = fail(COBS_GPU_ERR_ARG, "a query did not yield one result per document");
                return rs;
            }
            if (ps.need_rows && b->topk_k == 0 && ix->tune.device_rank != 0 && rank_on_device_applies(b, n)) {
                // whole (assembled, global) rows: ordered on the device, the records cross PCIe (rank.cpp) -- on a
                // host thread this loop ranks ~90 queries x 100 000 documents per second
                s = rank_on_device(b, 0, n, num_results, hits, cap, &used, hit_offsets + ps.g0, &overflow);
                if (s != COBS_GPU_ERR_UNSUPPORTED) return s;       // (no room for its workspace: the host loop below)
            }
            if (ps.need_rows && (s = event_bounded(c, x.x_done, "the all-gather of the count rows")) != COBS_GPU_OK) return s;
            for (size_t q = ps.g0; q < ps.g1; ++q) {
                size_t m = 0;
                s = cobs_gpu_batch_hits_host(b, q - ps.g0, num_results, overflow ? nullptr : hits + used, overflow ? 0 : cap - used, &m);
                if (s == COBS_GPU_ERR_CAPACITY || (overflow && s == COBS_GPU_ERR_ARG)) overflow = true;
                else if (s != COBS_GPU_OK) return s;
                used += m;
                hit_offsets[q + 1] = used;
            }
            return COBS_GPU_OK;
        };

        // The pipeline.  Iteration i: launch(i) | stage A(i-1) | record of pass i into the all-gather | stage B(i-2).
        // On the exchange stream the order is agree(0), [exchange(0), agree(1)], [exchange(1), agree(2)], ... -- the same on
        // every rank; exchange(i-1) runs while K2(i) does, agree(i) waits for K2(i).  An error that every rank reads from
        // an agreement (stage A) stops all of them at the same place: nothing else is outstanding on the communicator then.
        // An error a rank meets alone later (exchange, results) is carried by its next record, and by one more agreement at
        // the end where the ranks share one result (split): nobody returns success beside a rank that failed.
        for (size_t i = 0; i < np + 2; ++i) {
            if (i < np) launch(passes[i]);
            if (i >= 1 && i - 1 < np) {
                cobs_gpu_status s = stage_a(passes[i - 1]);
                if (s != COBS_GPU_OK) {
                    const std::string keep = last_error_text();
                    drain();
                    return fail(s, keep);
                }
            }
            if (i < np) {
                cobs_gpu_status s = agree_launch(passes[i]);
                if (s != COBS_GPU_OK) {
                    const std::string keep = last_error_text();
                    drain();
                    return fail(s, keep);
                }
            }
            if (i >= 2 && i - 2 < np && carried == 0) carry(stage_b(passes[i - 2]));
        }
        if (shared) {
            // one result, written by all ranks together: they end with the same status
            cobs_gpu_batch* b = ix->scratch[passes.back().slot];
            Agreed ag;
            cobs_gpu_status s = meta_launch(b, c, xs, false, COBS_GPU_OK, carried);
            if (s == COBS_GPU_OK) s = meta_wait(b, c, &ag);
            if (s != COBS_GPU_OK) { const std::string keep = last_error_text(); drain(); return fail(s, keep); }
            if (carried == 0 && ag.worst != COBS_GPU_OK)
                return fail(COBS_GPU_ERR_RCCL, "the ranking failed on another rank (status " + std::to_string(ag.worst) + ")");
        }
        if (carried) {
            drain();
            return fail((cobs_gpu_status)carried, carried_msg);
        }
        if (overflow) return fail(COBS_GPU_ERR_CAPACITY, "hit buffer too small; hit_offsets[nq] holds the needed size");
        return COBS_GPU_OK;
    });
}

}  // namespace

// ---------------------------------------------------------------------------
// The device-resident form: ONE batch of queries, uploaded once, scanned by every rank against its shard and left as
// count rows in global document order on the device -- the step bench.py --gpus N times.  The batch is cut into
// sub-batches (the reference's own loop is per batch of documents, classic_search.cpp:355-400; here the cut is over
// queries), each a cobs_gpu_batch: K1 of a sub-batch on that batch's own stream (tuning key hash_stream), K2 on the
// scan stream, the exchange on the exchange stream, tied by events only -- hash(i+1) | scan(i) | exchange(i-1) overlap,
// also across steps (scan i of step s+1 waits for exchange i of step s, whose source rows it overwrites, and for
// nothing else).
struct cobs_gpu_sharded_batch {
    cobs_gpu_index* ix = nullptr;
    cobs_gpu_comm* c = nullptr;
    std::vector<cobs_gpu_batch*> sub;
    std::vector<size_t> q0;                       // [sub + 1] first query of every sub-batch
    hipStream_t scan = nullptr, xchg = nullptr;
    std::vector<hipEvent_t> scanned, x_done;      // per sub-batch
    std::vector<bool> x_valid;
    static constexpr int kRing = 64;
    std::vector<hipEvent_t> xt0, xt1;             // [kRing * sub] timing events around the exchanges
    uint64_t steps = 0, read_steps = 0;
    uint32_t mode = COBS_GPU_XCHG_ALLTOALL;
    ~cobs_gpu_sharded_batch() {
        if (scan) (void)hipStreamSynchronize(scan);
        if (xchg) (void)hipStreamSynchronize(xchg);
        for (auto* b : sub) cobs_gpu_batch_destroy(b);
        for (auto e : scanned) if (e) (void)hipEventDestroy(e);
        for (auto e : x_done) if (e) (void)hipEventDestroy(e);
        for (auto e : xt0) if (e) (void)hipEventDestroy(e);
        for (auto e : xt1) if (e) (void)hipEventDestroy(e);
        if (scan) (void)hipStreamDestroy(scan);
        if (xchg) (void)hipStreamDestroy(xchg);
    }
};

extern "C" {

cobs_gpu_status cobs_gpu_sharded_search_batch(cobs_gpu_index* ix, cobs_gpu_comm* c, const char* const* queries,
                                              const size_t* lens, size_t nq, double threshold, size_t num_results,
                                              cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets, size_t* bad_query) {
    return sharded_search_impl(ix, c, queries, lens, nq, threshold, num_results, hits, cap, hit_offsets, bad_query, false);
}

cobs_gpu_status cobs_gpu_sharded_search_batch_split(cobs_gpu_index* ix, cobs_gpu_comm* c, const char* const* queries,
                                                    const size_t* lens, size_t nq, double threshold, size_t num_results,
                                                    cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets, size_t* bad_query) {
    return sharded_search_impl(ix, c, queries, lens, nq, threshold, num_results, hits, cap, hit_offsets, bad_query, true);
}

cobs_gpu_status cobs_gpu_sharded_batch_create(cobs_gpu_index* ix, cobs_gpu_comm* c, uint32_t sub_batches,
                                              cobs_gpu_sharded_batch** out) {
    if (!ix || !c || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (sub_batches == 0 || sub_batches > 64) return fail(COBS_GPU_ERR_ARG, "1 to 64 sub-batches");
    return guarded([&]() -> cobs_gpu_status {
        HIP_TRY(hipSetDevice(ix->device));
        std::unique_ptr<cobs_gpu_sharded_batch> sb(new cobs_gpu_sharded_batch);
        sb->ix = ix;
        sb->c = c;
        HIP_TRY(hipStreamCreateWithFlags(&sb->scan, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&sb->xchg, hipStreamNonBlocking));
        sb->sub.assign(sub_batches, nullptr);
        sb->scanned.assign(sub_batches, nullptr);
        sb->x_done.assign(sub_batches, nullptr);
        sb->x_valid.assign(sub_batches, false);
        sb->xt0.assign((size_t)cobs_gpu_sharded_batch::kRing * sub_batches, nullptr);
        sb->xt1.assign((size_t)cobs_gpu_sharded_batch::kRing * sub_batches, nullptr);
        sb->q0.assign(sub_batches + 1, 0);
        for (uint32_t i = 0; i < sub_batches; ++i) {
            cobs_gpu_status cs = cobs_gpu_batch_create(ix, 0, 0, &sb->sub[i]);
            if (cs != COBS_GPU_OK) return cs;
            HIP_TRY(hipEventCreateWithFlags(&sb->scanned[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sb->x_done[i], hipEventDisableTiming));
        }
        for (auto& e : sb->xt0) HIP_TRY(hipEventCreate(&e));
        for (auto& e : sb->xt1) HIP_TRY(hipEventCreate(&e));
        // K1 of a sub-batch on that batch's own stream: it runs under the scan / exchange of the others
        if (sub_batches > 1) ix->tune.hash_stream = 1;
        *out = sb.release();
        return COBS_GPU_OK;
    });
}

void cobs_gpu_sharded_batch_destroy(cobs_gpu_sharded_batch* sb) { delete sb; }

cobs_gpu_status cobs_gpu_sharded_batch_set_queries(cobs_gpu_sharded_batch* sb, const char* const* queries, const size_t* lens,
                                                   size_t nq) {
    if (!sb || (nq && (!queries || !lens))) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        HIP_TRY(hipSetDevice(sb->ix->device));
        HIP_TRY(hipStreamSynchronize(sb->scan));
        HIP_TRY(hipStreamSynchronize(sb->xchg));
        const size_t S = sb->sub.size();
        for (size_t i = 0; i <= S; ++i) sb->q0[i] = nq * i / S;
        for (size_t i = 0; i < S; ++i) {
            const size_t a = sb->q0[i], n = sb->q0[i + 1] - a;
            size_t bad = 0;
            cobs_gpu_status s = set_queries_on(sb->sub[i], queries + a, lens + a, n, nullptr, true, &bad, a);
            if (s != COBS_GPU_OK) return s;
            sb->x_valid[i] = false;
        }
        sb->steps = sb->read_steps = 0;
        return COBS_GPU_OK;
    });
}

// One step: every sub-batch hashed, scanned and its count rows exchanged (mode: cobs_gpu_exchange_mode).  Asynchronous.
cobs_gpu_status cobs_gpu_sharded_batch_step(cobs_gpu_sharded_batch* sb, double threshold, uint32_t mode) {
    if (!sb) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (mode > COBS_GPU_XCHG_REDUCE) return fail(COBS_GPU_ERR_ARG, "unknown exchange mode");
    return guarded([&]() -> cobs_gpu_status {
        HIP_TRY(hipSetDevice(sb->ix->device));
        const size_t S = sb->sub.size();
        const size_t ring = (size_t)(sb->steps % cobs_gpu_sharded_batch::kRing) * S;
        sb->mode = mode;
        for (size_t i = 0; i < S; ++i) {
            cobs_gpu_batch* b = sb->sub[i];
            // the previous step's exchange read the count rows this scan overwrites
            if (sb->x_valid[i]) HIP_TRY(hipStreamWaitEvent(sb->scan, sb->x_done[i], 0));
            cobs_gpu_status s = run_impl(b, threshold, 0, sb->scan, true);
            if (s != COBS_GPU_OK) return s;
            HIP_TRY(hipEventRecord(sb->scanned[i], sb->scan));
            HIP_TRY(hipStreamWaitEvent(sb->xchg, sb->scanned[i], 0));
            HIP_TRY(hipEventRecord(sb->xt0[ring + i], sb->xchg));
            s = cobs_gpu_batch_exchange_counts(b, sb->c, mode, sb->xchg);
            if (s != COBS_GPU_OK) return s;
            HIP_TRY(hipEventRecord(sb->xt1[ring + i], sb->xchg));
            HIP_TRY(hipEventRecord(sb->x_done[i], sb->xchg));
            sb->x_valid[i] = true;
        }
        sb->steps++;
        return COBS_GPU_OK;
    });
}

// Wait for everything queued so far (the exchange under the communicator's time limit); reports an invalid query.
cobs_gpu_status cobs_gpu_sharded_batch_sync(cobs_gpu_sharded_batch* sb, size_t* bad_query) {
    if (!sb) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        HIP_TRY(hipSetDevice(sb->ix->device));
        cobs_gpu_status first = COBS_GPU_OK;
        std::string keep;
        for (size_t i = 0; i < sb->sub.size(); ++i) {
            cobs_gpu_batch* b = sb->sub[i];
            if (!b->ran) continue;
            size_t bad = 0;
            const cobs_gpu_status s = cobs_gpu_batch_sync(b, sb->scan, &bad);
            if (s != COBS_GPU_OK && first == COBS_GPU_OK) {
                first = s;
                keep = s == COBS_GPU_ERR_INVALID_BASE ? "Invalid DNA base pair in query string. Only ACGT are allowed. (query " +
                                                        std::to_string(sb->q0[i] + bad) + ")"
                                                      : last_error_text();
                if (bad_query) *bad_query = sb->q0[i] + bad;
            }
        }
        if (cobs_gpu_status ws = sync_bounded(sb->c, sb->xchg, "the exchange of the count rows"); ws != COBS_GPU_OK) return ws;
        if (first != COBS_GPU_OK) return fail(first, keep);
        return COBS_GPU_OK;
    });
}

size_t cobs_gpu_sharded_batch_subs(const cobs_gpu_sharded_batch* sb) { return sb ? sb->sub.size() : 0; }

cobs_gpu_batch* cobs_gpu_sharded_batch_sub(cobs_gpu_sharded_batch* sb, size_t i, size_t* q_begin, size_t* q_count) {
    if (!sb || i >= sb->sub.size()) return nullptr;
    if (q_begin) *q_begin = sb->q0[i];
    if (q_count) *q_count = sb->q0[i + 1] - sb->q0[i];
    return sb->sub[i];
}

// Per step, summed over the sub-batches and averaged over the steps since the previous call (at most the last 64), after a
// sync: out = scan ms | hash ms | exchange ms | algorithmic bytes | bytes received over the fabric | scan launches
// | steps averaged over | 0.  Events on the streams the kernels and collectives actually ran on.
cobs_gpu_status cobs_gpu_sharded_batch_times(cobs_gpu_sharded_batch* sb, double out[8]) {
    if (!sb || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        for (int i = 0; i < 8; ++i) out[i] = 0;
        HIP_TRY(hipSetDevice(sb->ix->device));
        const size_t S = sb->sub.size();
        uint64_t first = sb->read_steps;
        if (sb->steps - first > (uint64_t)cobs_gpu_sharded_batch::kRing) first = sb->steps - cobs_gpu_sharded_batch::kRing;
        if (first == sb->steps && sb->steps) first = sb->steps - 1;
        const double n = (double)(sb->steps - first);
        for (size_t i = 0; i < S; ++i) {
            cobs_gpu_batch* b = sb->sub[i];
            if (!b->ran) continue;
            float sm = 0, hm = 0;
            cobs_gpu_status s = cobs_gpu_batch_kernel_ms(b, &sm, &hm);
            if (s != COBS_GPU_OK) return s;
            out[0] += sm;
            out[1] += hm;
            out[3] += (double)b->stats[0];
            out[4] += (double)cobs_gpu_batch_exchange_bytes(b);
            out[5] += (double)b->stats[1];
            for (uint64_t st = first; st < sb->steps; ++st) {
                float xm = 0;
                const size_t ring = (size_t)(st % cobs_gpu_sharded_batch::kRing) * S;
                HIP_TRY(hipEventElapsedTime(&xm, sb->xt0[ring + i], sb->xt1[ring + i]));
                out[2] += xm / (n > 0 ? n : 1.0);
            }
        }
        out[6] = n;
        sb->read_steps = sb->steps;
        return COBS_GPU_OK;
    });
}

}  // extern "C"
