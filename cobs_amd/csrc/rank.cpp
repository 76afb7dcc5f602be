// cobs_amd/csrc/rank.cpp -- counts_to_result over whole score rows on the device: the host side of
// rank_kernels.hip.  The reference's default call (threshold 0, no limit; also its own benchmark,
// src/cobs.cpp:618-626) returns EVERY document of every query in rank order
// (cobs/query/classic_search.cpp:109-202).  The score rows stay in HBM; a window of queries is
// ordered there (one work-group per query), the ordered records cross PCIe once as the
// cobs_gpu_hit array the caller gets -- 12 bytes per document instead of a score row that host
// threads then sort -- and while window w crosses, window w+1 is being ordered and the host copies
// window w-1 from the pinned landing buffer into the caller's (pageable) memory.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "engine.hpp"

namespace cobs_amd {

struct RankWork {
    DevBuf<cobs_gpu_hit> out[2];
    DevBuf<uint2> pairs[2];
    DevBuf<uint32_t> cnt[2];            // [window]: results per query; npass of multi-pass sorts behind it
    DevBuf<RankPart> parts;
    DevBuf<uint8_t> by_score;
    PinnedBuf<uint8_t> land[2];         // records of a window, then its counts
    hipStream_t copy_stream = nullptr;
    hipEvent_t ranked[2] = {nullptr, nullptr}, landed[2] = {nullptr, nullptr};
    ~RankWork() {
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto e : ranked) if (e) (void)hipEventDestroy(e);
        for (auto e : landed) if (e) (void)hipEventDestroy(e);
    }
};

void destroy_rank_work(RankWork* w) { delete w; }

namespace {

constexpr size_t kWindowBytes = 32u << 20;     // per window: short head (first ordering) and tail (last host copy) of the pipeline

// copy `bytes` from the pinned landing buffer into caller memory with a few threads
void spread_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
    const size_t kPiece = 4u << 20;
    const unsigned nthr = (unsigned)std::min<size_t>(std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())),
                                                      (bytes + kPiece - 1) / kPiece);
    if (nthr <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> pool;
    const size_t per = (bytes + nthr - 1) / nthr;
    for (unsigned t = 0; t < nthr; ++t) {
        const size_t a = std::min(bytes, (size_t)t * per), e = std::min(bytes, a + per);
        if (e > a) pool.emplace_back([=]() { std::memcpy(dst + a, src + a, e - a); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

bool rank_on_device_applies(const cobs_gpu_batch* b, size_t nq) {
    // the result has to come from whole score rows (no K3 list, no complete hit pool): the batch's own rows, or --
    // after an exchange (comm.cpp) -- the assembled global rows of the queries this rank holds
    if (!b->have_counts || b->topk_k != 0 || nq < 4) return false;
    if (b->selected && (b->pool_global || b->h_nhits() <= b->hit_cap)) return false;
    if (b->ix->local_counts >= 0xFFFFFFF0ull || b->ix->local_counts == 0 || b->ix->total_counts >= 0xFFFFFFF0ull) return false;
    if (b->view_global && (b->g_rows == nullptr || b->g_qn == 0)) return false;
    return true;
}

// Queries [q_first, q_first + nq) of the last (synced) run of `b`, ranked on the device.  Query q_first + i's results -- at most
// `limit` (0 = all) -- are appended at hits + *used, hit_offsets[i + 1] = the new *used.  When the
// caller's buffer is too small the offsets keep counting (the caller reports the needed capacity)
// and *overflow is set; hits are then not valid, as in the host path.
cobs_gpu_status rank_on_device(cobs_gpu_batch* b, size_t q_first, size_t nq, size_t limit, cobs_gpu_hit* hits, size_t cap,
                               size_t* used, size_t* hit_offsets, bool* overflow) {
    cobs_gpu_index* ix = b->ix;
    HIP_TRY(hipSetDevice(ix->device));
    if (!b->rank) b->rank = new RankWork;
    RankWork& w = *b->rank;
    hipStream_t st = b->own_stream;             // the stream the pass ran on (host-buffer API)
    if (!w.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&w.copy_stream, hipStreamNonBlocking));
        for (auto& e : w.ranked) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : w.landed) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // the files' slices as the ranked rows hold them: this shard's slots back to back (local rows), or every
    // file whole at its document offset (global rows assembled by an exchange)
    const bool glob = b->view_global;
    const uint64_t row_elems = glob ? ix->total_counts : ix->local_counts;
    std::vector<RankPart> parts;
    size_t per_query = 0;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        const Part& p = ix->parts[f];
        RankPart rp;
        rp.thr = b->threshold > 0.0 ? b->work[f].thr.p : nullptr;
        rp.file_no = (uint32_t)f;
        if (glob) {
            rp.slot0 = (uint32_t)p.doc_offset;
            rp.doc_first = 0;
            rp.ndocs = (uint32_t)p.meta.doc_names.size();
        } else {
            if (p.slot_count == 0) continue;
            rp.slot0 = (uint32_t)p.local_offset;
            rp.doc_first = (uint32_t)p.slot_begin;
            const uint64_t d1 = std::min<uint64_t>(p.slot_begin + p.slot_count, p.meta.doc_names.size());
            rp.ndocs = d1 > p.slot_begin ? (uint32_t)(d1 - p.slot_begin) : 0u;
        }
        per_query += rp.ndocs;
        parts.push_back(rp);
    }
    if (parts.empty() || per_query == 0) {
        for (size_t i = 0; i < nq; ++i) hit_offsets[i + 1] = *used;
        return COBS_GPU_OK;
    }
    const size_t stride = limit == 0 ? per_query : std::min(limit, per_query);
    HIP_TRY(w.parts.reserve(parts.size()));
    HIP_TRY(hipMemcpyAsync(w.parts.p, parts.data(), parts.size() * sizeof(RankPart), hipMemcpyHostToDevice, st));
    if (b->view_global && (q_first < b->g_q0 || q_first + nq > b->g_q0 + b->g_qn))
        return fail(COBS_GPU_ERR_ARG, "this rank does not hold the exchanged rows of those queries");
    std::vector<uint8_t> by_score(b->nq);
    for (size_t q = 0; q < b->nq; ++q) by_score[q] = total_hashes(b, q) > 1 ? 1 : 0;   // max_counts <= 1: index order
    HIP_TRY(w.by_score.reserve(b->nq));
    HIP_TRY(hipMemcpyAsync(w.by_score.p, by_score.data(), b->nq, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));          // the two host vectors may go; everything below is asynchronous

    // radix passes of at most 12 bits over the score bits the scan produced
    const uint32_t planes = (uint32_t)b->planes;
    const uint32_t npasses = (planes + 11u) / 12u;
    const uint32_t pbits = (planes + npasses - 1u) / npasses;
    const size_t wq = std::max<size_t>(1, std::min<size_t>(nq, kWindowBytes / (stride * sizeof(cobs_gpu_hit))));
    const size_t land_bytes = wq * stride * sizeof(cobs_gpu_hit);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(w.out[i].reserve(wq * stride));
        HIP_TRY(w.cnt[i].reserve(2 * wq));
        HIP_TRY(w.land[i].reserve(land_bytes + 4 * wq));
        if (npasses > 1) HIP_TRY(w.pairs[i].reserve(wq * row_elems));
    }
    struct Win { size_t q0, n; };
    std::vector<Win> wins;
    for (size_t q0 = 0; q0 < nq; q0 += wq) wins.push_back(Win{q0, std::min(wq, nq - q0)});
    auto launch = [&](size_t wi) -> cobs_gpu_status {
        const Win& wn = wins[wi];
        const int s = (int)(wi & 1);
        if (wi >= 2) HIP_TRY(hipStreamWaitEvent(st, w.landed[s], 0));     // the window that used these buffers has left them
        RankArgs a{};
        a.rows = glob ? (const void*)b->g_rows : (const void*)b->counts.p;
        a.row_stride = row_elems;
        a.row_q0 = glob ? (uint32_t)b->g_q0 : 0u;
        a.parts = w.parts.p;
        a.by_score = w.by_score.p;
        a.npass = w.cnt[s].p + wq;
        a.out = w.out[s].p;
        a.out_count = w.cnt[s].p;
        a.pair_stride = row_elems;
        a.out_stride = stride;
        a.nparts = (uint32_t)parts.size();
        a.nslots = (uint32_t)row_elems;
        a.q0 = (uint32_t)(q_first + wn.q0);
        a.nq = (uint32_t)wn.n;
        a.limit = (uint32_t)std::min<size_t>(stride, 0xFFFFFFFFu);
        a.score_bytes = b->elem_bytes;
        for (uint32_t ps = 0; ps < npasses; ++ps) {
            a.shift = ps * pbits;
            a.bits = std::min(pbits, planes - a.shift);
            a.src = w.pairs[(ps + 1) & 1].p;
            a.dst = w.pairs[ps & 1].p;
            HIP_TRY(launch_rank(a, ps == 0, ps + 1 == npasses, st));
        }
        HIP_TRY(hipEventRecord(w.ranked[s], st));
        HIP_TRY(hipStreamWaitEvent(w.copy_stream, w.ranked[s], 0));
        HIP_TRY(hipMemcpyAsync(w.land[s].p, w.out[s].p, wn.n * stride * sizeof(cobs_gpu_hit), hipMemcpyDeviceToHost, w.copy_stream));
        HIP_TRY(hipMemcpyAsync(w.land[s].p + land_bytes, w.cnt[s].p, 4 * wn.n, hipMemcpyDeviceToHost, w.copy_stream));
        HIP_TRY(hipEventRecord(w.landed[s], w.copy_stream));
        return COBS_GPU_OK;
    };
    const bool trace = ix->tune.trace;
    double t_wait = 0, t_copy = 0, t_prep = now_s();
    cobs_gpu_status rs = launch(0);
    if (rs != COBS_GPU_OK) return rs;
    t_prep = now_s() - t_prep;
    for (size_t wi = 0; wi < wins.size(); ++wi) {
        // window wi+1 is ordered and crosses PCIe while the host takes window wi out of its landing
        // buffer -- which window wi+2 will reuse, so wi+2 is launched only after this copy
        if (wi + 1 < wins.size() && (rs = launch(wi + 1)) != COBS_GPU_OK) break;
        const int s = (int)(wi & 1);
        double t0 = now_s();
        if (hipEventSynchronize(w.landed[s]) != hipSuccess) { rs = hip_fail(hipGetLastError(), "rank window"); break; }
        t_wait += now_s() - t0;
        t0 = now_s();
        const Win& wn = wins[wi];
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(w.land[s].p + land_bytes);
        const uint8_t* rec = w.land[s].p;
        bool full = true;
        for (size_t i = 0; i < wn.n; ++i) full = full && cnt[i] == stride;
        if (full && !*overflow && wn.n * stride <= cap - *used) {
            // every query of the window yields `stride` results (the default call): one block copy
            spread_copy(reinterpret_cast<uint8_t*>(hits + *used), rec, wn.n * stride * sizeof(cobs_gpu_hit));
            for (size_t i = 0; i < wn.n; ++i) {
                *used += stride;
                hit_offsets[wn.q0 + i + 1] = *used;
            }
            t_copy += now_s() - t0;
            continue;
        }
        for (size_t i = 0; i < wn.n; ++i) {
            const size_t n = cnt[i];
            if (!*overflow && n <= cap - *used)
                std::memcpy(hits + *used, rec + i * stride * sizeof(cobs_gpu_hit), n * sizeof(cobs_gpu_hit));
            else
                *overflow = true;
            *used += n;
            hit_offsets[wn.q0 + i + 1] = *used;
        }
    }
    if (trace)
        std::fprintf(stderr, "[cobs_gpu] device ranking: %zu queries x %zu records in %zu windows; first launch %.3f ms, "
                     "waiting for windows %.3f ms, copying out of the landing buffers %.3f ms\n",
                     nq, stride, wins.size(), t_prep * 1e3, t_wait * 1e3, t_copy * 1e3);
    if (rs != COBS_GPU_OK) {                    // nothing of this batch may still be in flight when the caller sees the error
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(w.copy_stream);
    }
    return rs;
}

}  // namespace cobs_amd
