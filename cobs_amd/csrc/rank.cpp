// cobs_amd/csrc/rank.cpp -- counts_to_result over whole score rows on the device: the host side of
// rank_kernels.hip.  The reference's default call (threshold 0, no limit; also its own benchmark,
// src/cobs.cpp:618-626) returns EVERY document of every query in rank order
// (cobs/query/classic_search.cpp:109-202).  The score rows stay in HBM; a window of queries is
// ordered there (one work-group per query), the ordered (slot, score) records cross PCIe once -- 4 bytes per
// result where slot and score fit one word together (8 otherwise), already in rank order, instead of a score row
// that host threads then sort -- and while window w
// crosses, window w+1 is being ordered and the host expands window w-1 from the pinned landing buffer into the
// cobs_gpu_hit records (file, document, score) of the caller's (pageable) array.
#include <sys/mman.h>

#if defined(__SSE2__)
#include <emmintrin.h>      // non-temporal 16-byte stores of the host expansion (x86 hosts; a scalar loop elsewhere)
#endif
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "engine.hpp"

namespace cobs_amd {

// Host threads that stay around for the life of the process (one pool of up to 31, shared by every handle; a
// window's expansion holds it for a fraction of a millisecond): expanding a window of records is ~1 ms of work,
// spawning the threads per window would cost as much again.
// CPUs of the NUMA node the device's PCIe root sits on (sysfs), empty if unknown: the pool's threads read the pinned
// landing buffers the DMA engine filled -- they live next to the GPU -- and stream the results out
static std::vector<int> cpus_near_device(int device) {
    std::vector<int> cpus;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return cpus; }
    for (char& ch : bdf) ch = (char)std::tolower((unsigned char)ch);
    int node = -1;
    if (FILE* f = std::fopen((std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return cpus;
    char list[4096] = {0};
    if (FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
        if (!std::fgets(list, sizeof list, f)) list[0] = 0;
        std::fclose(f);
    }
    for (const char* p = list; *p && *p != '\n';) {
        char* e = nullptr;
        const long a = std::strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        p = e;
        if (*p == '-') { b = std::strtol(p + 1, &e, 10); p = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) cpus.push_back((int)c);
        if (*p == ',') ++p;
    }
    return cpus;
}

class ExpandPool {
public:
    explicit ExpandPool(unsigned n, const std::vector<int>& cpus) {
        for (unsigned i = 0; i < n; ++i) threads_.emplace_back([this]() { loop(); });
        if (!cpus.empty()) {             // (the process's own affinity mask still applies: the intersection, if any)
            cpu_set_t allowed, want;
            CPU_ZERO(&want);
            if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
                int n_ok = 0;
                for (int c : cpus) if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); ++n_ok; }
                if (n_ok >= 2)
                    for (auto& t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof want, &want);
            }
        }
    }
    ~ExpandPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    unsigned size() const { return (unsigned)threads_.size(); }
    // run fn(i) for i in [0, n) on the pool (and the caller), return when all are done.  Jobs are handed out by an atomic
    // counter (a mutex per job cost ~1 us under 32 contending threads: a piece of 16 Ki-record jobs took longer to hand out
    // than to expand); the mutex only guards the start of a run (generation, function) and the two waits.
    void run(size_t n, const std::function<void(size_t)>& fn) {
        std::lock_guard<std::mutex> one_job(run_mu_);      // callers of different handles take turns
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn;
            total_ = n;
            next_.store(0, std::memory_order_relaxed);
            pending_.store(n, std::memory_order_relaxed);
            ++gen_;
        }
        cv_.notify_all();
        work(fn);                                   // the caller works too
        std::unique_lock<std::mutex> g(mu_);
        // (no worker may still be inside work() when fn goes out of scope or the counters are set for the next run)
        done_.wait(g, [this]() { return pending_.load(std::memory_order_acquire) == 0 && active_ == 0; });
        fn_ = nullptr;
    }

private:
    void work(const std::function<void(size_t)>& fn) {
        for (;;) {
            const size_t i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= total_) break;
            fn(i);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> g(mu_);
                done_.notify_all();
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* fn;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&]() { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
                if (!fn) continue;                  // (woke up after the run was over)
                ++active_;
            }
            work(*fn);
            std::lock_guard<std::mutex> g(mu_);
            if (--active_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    std::atomic<size_t> next_{0}, pending_{0};
    size_t total_ = 0, active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

ExpandPool* expand_pool(int device) {
    // (one pool per process, placed by the first device that ranks: a process drives one GPU, or the GPUs of one node)
    static const unsigned want = []() {
        const char* e = std::getenv("COBS_GPU_EXPAND_THREADS");            // (A/B of the pool size)
        return e ? (unsigned)std::max(1, std::atoi(e)) : 31u;
    }();
    static ExpandPool pool(std::min(want, std::max(2u, std::thread::hardware_concurrency()) - 1u), cpus_near_device(device));
    return &pool;
}

// One ranking between its two halves: rank_launch (workspace, the ordering kernels of the first spans, the first two
// pieces on their way over PCIe -- everything asynchronous) and the drain inside rank_on_device (pieces expanded into the
// caller's array as they land).  The host-buffer API launches the ranking of a pass right behind its scan, so that it
// runs while the pass BEFORE it is expanded (host_api.cpp).
struct RankJob {
    bool active = false;
    size_t q_first = 0, nq = 0, limit = 0;
    uint64_t run_seq = 0;               // the run of the batch whose rows it orders
    std::vector<RankPart> parts;
    std::vector<uint8_t> by_score;
    size_t per_query = 0, stride = 0;
    bool glob = false, slim = false;
    uint64_t row_elems = 0;
    uint32_t planes = 0, npasses = 0, pbits = 0, pack_bits = 0, nbins = 0, nseg = 1;
    size_t rec = 0, words = 0, qbytes = 0, sq = 0, pq = 0, land_bytes = 0, nspans = 0;
    struct Piece { size_t span, q0, n; bool last_of_span; };      // q0 relative to q_first
    std::vector<Piece> pieces;
    size_t spans_launched = 0, pieces_issued = 0;
    double t_prep = 0;
};

struct RankWork {
    static constexpr int kDepth = 3;    // landing buffers: one being filled over PCIe, one queued, one being expanded
    RankJob job;
    DevBuf<uint2> out[2];               // (slot, score) records of a span of queries, in rank order
    DevBuf<uint2> pairs[2];
    DevBuf<uint32_t> cnt[2];            // [span]: results per query; npass of multi-pass sorts behind it
    DevBuf<uint32_t> packed[2];         // slim records: the slots of a span as a bit stream (SlotPackArgs)
    DevBuf<uint32_t> bins[2];           // ... and its records per score (RankArgs::bin_count)
    DevBuf<uint32_t> seghist;           // single-pass sorts of long rows: histograms / positions of the segments (RankArgs::seg_hist)
    DevBuf<RankPart> parts;
    DevBuf<uint8_t> by_score;
    PinnedBuf<uint8_t> land[kDepth];    // records of a piece, then its counts
    PinnedBuf<uint8_t> h_stage;         // parts | by_score on their way to the device: pinned, so that the launch half never waits
                                        // for the stream (a copy from pageable memory does, inside the runtime) [ADVICE r5]
    hipStream_t copy_stream = nullptr;
    hipEvent_t ranked[2] = {nullptr, nullptr}, drained[2] = {nullptr, nullptr}, landed[kDepth] = {nullptr, nullptr, nullptr};
    ~RankWork() {
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);      // (a job that was launched and never drained)
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto e : ranked) if (e) (void)hipEventDestroy(e);
        for (auto e : drained) if (e) (void)hipEventDestroy(e);
        for (auto e : landed) if (e) (void)hipEventDestroy(e);
    }
};

void destroy_rank_work(RankWork* w) { delete w; }

namespace {

constexpr size_t kSpanBytes = 512u << 20;      // records one kernel launch orders (two such device buffers)

// records [first, first + cnt) of a source -> d[0 .. cnt).  A source reads ONE record, `one(i, slot, score)`, or -- x86 --
// FOUR, `four(i, slots, scores)` as two vectors, always with ascending i.  Four 12-byte results are three 16-byte
// non-temporal stores built with shuffles (building them from scalars cost more than the stores: the expansion of the
// default call ran at 170 GB/s where 8 threads of plain stores reach 300, scripts/probes/nt_store_probe.cpp).
#if defined(__SSE2__)
static inline void store4(cobs_gpu_hit* d, __m128i docs, __m128i scores, __m128 files) {
    const __m128 A = _mm_castsi128_ps(_mm_unpacklo_epi32(docs, scores));    // d0 c0 d1 c1
    const __m128 B = _mm_castsi128_ps(_mm_unpackhi_epi32(docs, scores));    // d2 c2 d3 c3
    const __m128 t0 = _mm_shuffle_ps(files, A, _MM_SHUFFLE(1, 0, 0, 0));    // f  f  d0 c0
    const __m128 u = _mm_shuffle_ps(files, B, _MM_SHUFFLE(0, 0, 0, 0));     // f  f  d2 d2
    const __m128 v = _mm_shuffle_ps(B, files, _MM_SHUFFLE(0, 0, 1, 1));     // c2 c2 f  f
    __m128i* o = reinterpret_cast<__m128i*>(d);
    _mm_stream_si128(o + 0, _mm_shuffle_epi32(_mm_castps_si128(t0), _MM_SHUFFLE(0, 3, 2, 0)));   // f  d0 c0 f
    _mm_stream_si128(o + 1, _mm_castps_si128(_mm_shuffle_ps(A, u, _MM_SHUFFLE(2, 0, 3, 2))));    // d1 c1 f  d2
    _mm_stream_si128(o + 2, _mm_castps_si128(_mm_shuffle_ps(v, B, _MM_SHUFFLE(3, 2, 2, 0))));    // c2 f  d3 c3
}
#endif

template <class Source>
void emit_hits(cobs_gpu_hit* d, size_t first, size_t cnt, const std::vector<RankPart>& parts, Source src) {
    static_assert(sizeof(cobs_gpu_hit) == 12, "three u32 per result");
    if (parts.size() == 1) {
        const RankPart pt = parts[0];
        const uint32_t bias = pt.doc_first - pt.slot0, f = pt.file_no;
        size_t i = 0;
#if defined(__SSE2__)
        if ((reinterpret_cast<uintptr_t>(d) & 3u) == 0) {
            for (; i < cnt && (reinterpret_cast<uintptr_t>(d + i) & 15u) != 0; ++i) {     // at most 3: 12 i mod 16
                uint32_t slot, score;
                src.one(first + i, slot, score);
                d[i] = cobs_gpu_hit{f, slot + bias, score};
            }
            const __m128i vbias = _mm_set1_epi32((int)bias);
            const __m128 vf = _mm_castsi128_ps(_mm_set1_epi32((int)f));
            for (; i + 4 <= cnt; i += 4) {
                __m128i sl, sc;
                src.four(first + i, sl, sc);
                store4(d + i, _mm_add_epi32(sl, vbias), sc, vf);
            }
            _mm_sfence();
        }
#endif
        for (; i < cnt; ++i) {
            uint32_t slot, score;
            src.one(first + i, slot, score);
            d[i] = cobs_gpu_hit{f, slot + bias, score};
        }
        return;
    }
    for (size_t i = 0; i < cnt; ++i) {
        uint32_t slot, score;
        src.one(first + i, slot, score);
        size_t p = 0;
        while (p + 1 < parts.size() && slot >= parts[p + 1].slot0) ++p;
        d[i] = cobs_gpu_hit{parts[p].file_no, parts[p].doc_first + (slot - parts[p].slot0), score};
    }
}

// one u32 per record, score << pack_bits | slot
struct PackedSource {
    const uint32_t* s;
    uint32_t smask, pack_bits;
    void one(size_t i, uint32_t& slot, uint32_t& score) const { slot = s[i] & smask; score = s[i] >> pack_bits; }
#if defined(__SSE2__)
    void four(size_t i, __m128i& slots, __m128i& scores) const {
        const __m128i r = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + i));
        slots = _mm_and_si128(r, _mm_set1_epi32((int)smask));
        scores = _mm_srl_epi32(r, _mm_cvtsi32_si128((int)pack_bits));
    }
#endif
};

// (slot, score) pairs
struct PairSource {
    const uint2* s;
    void one(size_t i, uint32_t& slot, uint32_t& score) const { slot = s[i].x; score = s[i].y; }
#if defined(__SSE2__)
    void four(size_t i, __m128i& slots, __m128i& scores) const {
        const __m128 a = _mm_loadu_ps(reinterpret_cast<const float*>(s + i)), b = _mm_loadu_ps(reinterpret_cast<const float*>(s + i + 2));
        slots = _mm_castps_si128(_mm_shuffle_ps(a, b, _MM_SHUFFLE(2, 0, 2, 0)));
        scores = _mm_castps_si128(_mm_shuffle_ps(a, b, _MM_SHUFFLE(3, 1, 3, 1)));
    }
#endif
};

// the slim form of one query: a bit stream of slots in score order + the number of records per bin (bin = nbins - 1 - score)
struct SlimSource {
    const uint32_t* bq;        // [nbins]
    const uint8_t* pk;         // the slot stream
    uint32_t slot_bits, nbins;
    uint64_t smask;
    uint32_t bin = 0;
    size_t upto = 0;           // records in bins [0, bin]
    SlimSource(const uint32_t* bins, const uint32_t* packed, uint32_t sb, uint32_t nb)
        : bq(bins), pk(reinterpret_cast<const uint8_t*>(packed)), slot_bits(sb), nbins(nb), smask((1ull << sb) - 1ull), upto(bins[0]) {}
    uint32_t slot_at(size_t p) const {
        const uint64_t bit = (uint64_t)p * slot_bits;
        uint64_t v;
        std::memcpy(&v, pk + (bit >> 5) * 4u, 8);
        return (uint32_t)((v >> (bit & 31u)) & smask);
    }
    void seek(size_t p) { while (upto <= p && bin + 1u < nbins) upto += bq[++bin]; }
    void one(size_t p, uint32_t& slot, uint32_t& score) {
        seek(p);
        score = nbins - 1u - bin;
        slot = slot_at(p);
    }
#if defined(__SSE2__)
    void four(size_t p, __m128i& slots, __m128i& scores) {
        slots = _mm_set_epi32((int)slot_at(p + 3), (int)slot_at(p + 2), (int)slot_at(p + 1), (int)slot_at(p));
        seek(p);
        if (upto >= p + 4 || bin + 1u >= nbins) {            // the four share a score (runs are ~1000 records long)
            scores = _mm_set1_epi32((int)(nbins - 1u - bin));
            return;
        }
        uint32_t c[4];
        for (int j = 0; j < 4; ++j) { seek(p + j); c[j] = nbins - 1u - bin; }
        scores = _mm_set_epi32((int)c[3], (int)c[2], (int)c[1], (int)c[0]);
    }
#endif
};

// The slim form of FULL lists (every real document of every query in score order: the reference's default call).  Per
// query `words` dwords of slots -- record p = bits [p * slot_bits, (p + 1) * slot_bits) of that stream -- and `nbins`
// counts, bin = nbins - 1 - score: the records are ordered by score, so the counts say where every score begins
// (rank_kernels.hip: RankArgs::bin_count, pack_slots_kernel).  `nq` queries of `stride` records each -> dst.
// (the stream of a query is read 8 bytes at a time: `words` leaves two dwords behind the last record)
void expand_slim(ExpandPool* pool, cobs_gpu_hit* dst, const uint32_t* packed, const uint32_t* bins, size_t nq, size_t stride,
                 size_t words, uint32_t slot_bits, uint32_t nbins, const std::vector<RankPart>& parts) {
    const size_t kPiece = 64u << 10;                         // records per job, at most; a job stays inside one query
    const size_t per_q = std::max<size_t>(1, (stride + kPiece - 1) / kPiece);
    const size_t job_n = (stride + per_q - 1) / per_q;       // (equal jobs: 100 000 records = 2 x 50 000, not 65 536 + 34 464)
    auto work = [&](size_t j) {
        const size_t q = j / per_q, p0 = (j % per_q) * job_n, p1 = std::min(stride, p0 + job_n);
        if (p0 >= p1) return;
        emit_hits(dst + q * stride + p0, p0, p1 - p0, parts, SlimSource(bins + q * nbins, packed + q * words, slot_bits, nbins));
    };
    const size_t jobs = nq * per_q;
    if (jobs <= 1 || !pool) { for (size_t j = 0; j < jobs; ++j) work(j); return; }
    const std::function<void(size_t)> job = work;
    pool->run(jobs, job);
}

// `n` records of the pinned landing buffer -> cobs_gpu_hit records in caller memory, with the pool's threads: the slot
// of the ranked row becomes (file, document).  pack_bits == 0: (slot, score) pairs of 8 bytes; else one u32 per record,
// score << pack_bits | slot (the form the device writes whenever slot and score fit 32 bits together).
// With 4-byte records the link delivers a window faster than the host used to expand it (2.75 ms of expansion against
// 1.8 ms of PCIe for 256 queries x 100 000 documents): the 12-byte results are written four at a time as three
// 16-byte non-temporal stores -- no read-for-ownership of 307 MB the caller has not looked at yet.
void expand_records(ExpandPool* pool, cobs_gpu_hit* dst, const void* src, size_t n, uint32_t pack_bits,
                    const std::vector<RankPart>& parts) {
    const uint32_t smask = pack_bits ? (1u << pack_bits) - 1u : 0u;
    auto work = [&parts, pack_bits, smask](cobs_gpu_hit* d, const void* sv, size_t first, size_t cnt) {
        if (pack_bits) emit_hits(d, first, cnt, parts, PackedSource{static_cast<const uint32_t*>(sv), smask, pack_bits});
        else emit_hits(d, first, cnt, parts, PairSource{static_cast<const uint2*>(sv)});
    };
    const size_t kPiece = 64u << 10;         // records per job
    const size_t jobs = (n + kPiece - 1) / kPiece;
    if (jobs <= 1 || !pool) { work(dst, src, 0, n); return; }
    const std::function<void(size_t)> job = [&](size_t j) {
        const size_t a = j * kPiece, e = std::min(n, a + kPiece);
        work(dst + a, src, a, e - a);
    };
    pool->run(jobs, job);
}

}  // namespace

bool rank_on_device_applies(const cobs_gpu_batch* b, size_t nq) {
    // the result has to come from whole score rows (no K3 list, no complete hit pool): the batch's own rows, or --
    // after an exchange (comm.cpp) -- the assembled global rows of the queries this rank holds
    if (!b->have_counts || b->topk_k != 0 || nq < 4 || b->planes < 1) return false;
    if (b->selected && (b->pool_global || b->h_nhits() <= b->hit_cap)) return false;
    if (b->ix->local_counts >= 0xFFFFFFF0ull || b->ix->local_counts == 0 || b->ix->total_counts >= 0xFFFFFFF0ull) return false;
    if (b->view_global && (b->g_rows == nullptr || b->g_qn == 0)) return false;
    return true;
}

namespace {

cobs_gpu_status job_launch_span(cobs_gpu_batch* b, RankWork& w, RankJob& j, size_t sp) {
    hipStream_t st = b->own_stream;
    const int s = (int)(sp & 1);
    const size_t s0 = sp * j.sq, n = std::min(j.nq, s0 + j.sq) - s0;
    if (sp >= 2) HIP_TRY(hipStreamWaitEvent(st, w.drained[s], 0));       // the span that used this buffer has crossed PCIe
    RankArgs a{};
    a.rows = j.glob ? (const void*)b->g_rows : (const void*)b->counts.p;
    a.row_stride = j.row_elems;
    a.row_q0 = j.glob ? (uint32_t)b->g_q0 : 0u;
    a.parts = w.parts.p;
    a.by_score = w.by_score.p;
    a.npass = w.cnt[s].p + j.sq;
    a.out = w.out[s].p;
    a.out_count = w.cnt[s].p;
    a.pair_stride = j.row_elems;
    a.out_stride = j.stride;
    a.nparts = (uint32_t)j.parts.size();
    a.nslots = (uint32_t)j.row_elems;
    a.q0 = (uint32_t)(j.q_first + s0);
    a.nq = (uint32_t)n;
    a.limit = (uint32_t)std::min<size_t>(j.stride, 0xFFFFFFFFu);
    a.score_bytes = b->elem_bytes;
    a.pack_bits = j.pack_bits;
    a.bin_count = j.slim ? w.bins[s].p : nullptr;
    a.seg_hist = j.nseg > 1 ? w.seghist.p : nullptr;
    a.nseg = j.nseg;
    for (uint32_t ps = 0; ps < j.npasses; ++ps) {
        a.shift = ps * j.pbits;
        a.bits = std::min(j.pbits, j.planes - a.shift);
        a.src = w.pairs[(ps + 1) & 1].p;
        a.dst = w.pairs[ps & 1].p;
        HIP_TRY(launch_rank(a, ps == 0, ps + 1 == j.npasses, st));
    }
    if (j.slim) {
        SlotPackArgs pa{};
        pa.in = reinterpret_cast<const uint32_t*>(w.out[s].p);
        pa.out = w.packed[s].p;
        pa.in_stride = j.stride;
        pa.n = (uint32_t)j.stride;
        pa.words = (uint32_t)j.words;
        pa.slot_bits = j.pack_bits;
        pa.nq = (uint32_t)n;
        HIP_TRY(launch_pack_slots(pa, st));
    }
    HIP_TRY(hipEventRecord(w.ranked[s], st));
    return COBS_GPU_OK;
}

cobs_gpu_status job_issue_piece(cobs_gpu_batch* b, RankWork& w, RankJob& j, size_t pi) {
    constexpr size_t kDepth = RankWork::kDepth;
    const RankJob::Piece& pc = j.pieces[pi];
    while (j.spans_launched <= pc.span) {                      // (a span is launched when its first piece is wanted ...
        cobs_gpu_status ls = job_launch_span(b, w, j, j.spans_launched);
        if (ls != COBS_GPU_OK) return ls;
        ++j.spans_launched;
    }
    const int s = (int)(pc.span & 1), l = (int)(pi % kDepth);
    const size_t off = pc.q0 - pc.span * j.sq;                 // queries into the span
    HIP_TRY(hipStreamWaitEvent(w.copy_stream, w.ranked[s], 0));
    if (j.slim) {
        HIP_TRY(hipMemcpyAsync(w.land[l].p, w.packed[s].p + off * j.words, pc.n * j.words * sizeof(uint32_t), hipMemcpyDeviceToHost,
                               w.copy_stream));
        HIP_TRY(hipMemcpyAsync(w.land[l].p + j.pq * j.words * sizeof(uint32_t), w.bins[s].p + off * j.nbins,
                               pc.n * j.nbins * sizeof(uint32_t), hipMemcpyDeviceToHost, w.copy_stream));
    } else {
        HIP_TRY(hipMemcpyAsync(w.land[l].p, reinterpret_cast<const uint8_t*>(w.out[s].p) + off * j.stride * j.rec, pc.n * j.stride * j.rec,
                               hipMemcpyDeviceToHost, w.copy_stream));
    }
    HIP_TRY(hipMemcpyAsync(w.land[l].p + j.land_bytes, w.cnt[s].p + off, 4 * pc.n, hipMemcpyDeviceToHost, w.copy_stream));
    HIP_TRY(hipEventRecord(w.landed[l], w.copy_stream));
    if (pc.last_of_span) {
        HIP_TRY(hipEventRecord(w.drained[s], w.copy_stream));
        if (j.spans_launched < j.nspans && j.spans_launched == pc.span + 2) {   // ... or as soon as its buffer drains: it is
                                                                                // ordered while the span in between crosses)
            cobs_gpu_status ls = job_launch_span(b, w, j, j.spans_launched);
            if (ls != COBS_GPU_OK) return ls;
            ++j.spans_launched;
        }
    }
    ++j.pieces_issued;
    return COBS_GPU_OK;
}

}  // namespace

// a ranking that was launched and never drained (its call failed in between): nothing of it may still be in flight
void rank_cancel(cobs_gpu_batch* b) {
    if (!b || !b->rank || !b->rank->job.active) return;
    (void)hipStreamSynchronize(b->own_stream);
    if (b->rank->copy_stream) (void)hipStreamSynchronize(b->rank->copy_stream);
    b->rank->job.active = false;
}

// The launch half: queries [q_first, q_first + nq) of the last run of `b` (its kernels may still be running on the batch's
// stream: everything here is queued behind them), at most `limit` (0 = all) results per query.  COBS_GPU_ERR_UNSUPPORTED
// (nothing launched): the workspace could not be allocated, rank on the host.
cobs_gpu_status rank_launch(cobs_gpu_batch* b, size_t q_first, size_t nq, size_t limit) {
    cobs_gpu_index* ix = b->ix;
    HIP_TRY(hipSetDevice(ix->device));
    if (!b->rank) b->rank = new RankWork;
    RankWork& w = *b->rank;
    rank_cancel(b);
    RankJob& j = w.job;
    hipStream_t st = b->own_stream;             // the stream the pass ran on (host-buffer API)
    if (!w.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&w.copy_stream, hipStreamNonBlocking));
        for (auto& e : w.ranked) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : w.drained) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : w.landed) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    j.q_first = q_first;
    j.nq = nq;
    j.limit = limit;
    j.run_seq = b->run_seq;
    j.pieces.clear();
    j.spans_launched = j.pieces_issued = 0;
    // the files' slices as the ranked rows hold them: this shard's slots back to back (local rows), or every
    // file whole at its document offset (global rows assembled by an exchange)
    j.glob = b->view_global;
    j.row_elems = j.glob ? ix->total_counts : ix->local_counts;
    j.parts.clear();
    j.per_query = 0;
    for (size_t f = 0; f < ix->parts.size(); ++f) {
        const Part& p = ix->parts[f];
        RankPart rp;
        rp.thr = b->threshold > 0.0 ? b->work[f].thr.p : nullptr;
        rp.file_no = (uint32_t)f;
        if (j.glob) {
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
        j.per_query += rp.ndocs;
        j.parts.push_back(rp);
    }
    j.stride = limit == 0 ? j.per_query : std::min(limit, j.per_query);
    if (j.parts.empty() || j.per_query == 0) {          // nothing to rank: the drain half writes the empty lists
        j.active = true;
        return COBS_GPU_OK;
    }
    if (b->view_global && (q_first < b->g_q0 || q_first + nq > b->g_q0 + b->g_qn))
        return fail(COBS_GPU_ERR_ARG, "this rank does not hold the exchanged rows of those queries");
    j.by_score.resize(b->nq);
    for (size_t q = 0; q < b->nq; ++q) j.by_score[q] = total_hashes(b, q) > 1 ? 1 : 0;   // max_counts <= 1: index order
    const size_t stride = j.stride;

    // radix passes of at most 12 bits over the score bits the scan produced
    j.planes = (uint32_t)b->planes;
    j.npasses = (j.planes + 11u) / 12u;
    j.pbits = (j.planes + j.npasses - 1u) / j.npasses;
    // record width: slot and score in one word where they fit (C3: 17 + 10 bits), else a pair -- the default call moves
    // one record per (query, document) over PCIe and is bound by exactly those bytes
    uint32_t slot_bits = 1;
    while (slot_bits < 32u && (1ull << slot_bits) < j.row_elems) ++slot_bits;
    j.pack_bits = (ix->tune.rank_pack != 0 && slot_bits + j.planes <= 32u) ? slot_bits : 0u;
    j.rec = j.pack_bits ? sizeof(uint32_t) : sizeof(uint2);
    // FULL lists in score order (the reference's default call: threshold 0, no limit) cross PCIe slimmer still: the order
    // of the slots plus the number of records per score say everything -- C3: 17 bits per result instead of 32, 54 MB
    // instead of 102 MB for 256 queries (expand_slim above; tuning key rank_slim = 0: the 4-byte records, A/B)
    j.slim = ix->tune.rank_slim != 0 && j.pack_bits != 0 && j.npasses == 1 && limit == 0 && !(b->threshold > 0.0) && stride == j.per_query;
    for (size_t q = q_first; j.slim && q < q_first + nq; ++q) j.slim = j.by_score[q] != 0;
    j.nbins = 1u << j.pbits;
    j.words = j.slim ? (stride * j.pack_bits + 31) / 32 + 2 : 0;                         // dwords of one query's slot stream
    j.qbytes = j.slim ? (j.words + j.nbins) * sizeof(uint32_t) : stride * j.rec;         // what crosses PCIe per query
    // Two granularities.  A SPAN is what one kernel launch orders: up to 512 MiB of records -- a launch wants hundreds of
    // queries to fill the device (launching per 32 MiB window left 84 % of the CUs idle and made the call kernel-bound:
    // profiles/r03_rank_kernel_stats.csv).  A PIECE is what crosses PCIe at a time: 16 MiB of a span's records into one
    // of three pinned landing buffers, expanded by the host while the next piece crosses.
    j.sq = std::max<size_t>(1, std::min<size_t>(nq, kSpanBytes / (stride * j.rec)));     // queries per span
    // (a piece = what is in flight per stage of the PCIe | expansion pipeline: the head of a call is the first piece's
    // crossing, its tail the last piece's expansion, in between both overlap -- tuning key rank_window_kib)
    const size_t window = (size_t)std::max<uint32_t>(ix->tune.rank_window_kib, 256u) << 10;
    // A single-pass sort of a long row is cut into segments, a work-group each (rank_kernels.hip: SEG): one work-group per
    // query is alone on its CU and latency-bound -- 0.36 ms for 100 000 documents however few queries there are, the
    // head of every default call.  Tuning key rank_segments: 0 = by row length, 1 = off, else that many.
    j.nseg = 1;
    if (j.npasses == 1) {
        j.nseg = ix->tune.rank_segments > 0 ? (uint32_t)std::min(ix->tune.rank_segments, 64)
                                            : (j.row_elems >= 65536 ? 8u : j.row_elems >= 16384 ? 4u : j.row_elems >= 8192 ? 2u : 1u);
        if (j.sq > 65535u) j.nseg = 1;
    }
    j.pq = std::max<size_t>(1, std::min<size_t>(j.sq, window / j.qbytes));   // queries per piece
    j.land_bytes = (j.pq * j.qbytes + 15) / 16 * 16;
    constexpr size_t kDepth = RankWork::kDepth;
    {   // the workspace: if the device or the pinned pool cannot give it, the caller ranks on the host as before
        bool ok = w.parts.reserve(j.parts.size()) == hipSuccess && w.by_score.reserve(b->nq) == hipSuccess &&
                  w.h_stage.reserve(round_up(j.parts.size() * sizeof(RankPart), 16) + b->nq) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i)
            ok = w.out[i].reserve((j.sq * stride * j.rec + sizeof(uint2) - 1) / sizeof(uint2)) == hipSuccess && w.cnt[i].reserve(2 * j.sq) == hipSuccess;
        for (size_t i = 0; i < kDepth && ok; ++i) ok = w.land[i].reserve(j.land_bytes + 4 * j.pq) == hipSuccess;
        if (ok && j.nseg > 1 && w.seghist.reserve(j.sq * 4u * j.nseg * (size_t)j.nbins) != hipSuccess) {
            (void)hipGetLastError();
            j.nseg = 1;
        }
        for (int i = 0; i < 2 && ok && j.slim; ++i)
            ok = w.packed[i].reserve(j.sq * j.words) == hipSuccess && w.bins[i].reserve(j.sq * j.nbins) == hipSuccess;
        if (ok && j.npasses > 1)
            for (auto& pr : w.pairs) ok = ok && pr.reserve(j.sq * j.row_elems) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            return COBS_GPU_ERR_UNSUPPORTED;
        }
    }
    // (through pinned staging the job owns: rank_cancel above waited for whatever read it before)
    uint8_t* h_by_score = w.h_stage.p + round_up(j.parts.size() * sizeof(RankPart), 16);
    std::memcpy(w.h_stage.p, j.parts.data(), j.parts.size() * sizeof(RankPart));
    std::memcpy(h_by_score, j.by_score.data(), b->nq);
    HIP_TRY(hipMemcpyAsync(w.parts.p, w.h_stage.p, j.parts.size() * sizeof(RankPart), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w.by_score.p, h_by_score, b->nq, hipMemcpyHostToDevice, st));
    j.nspans = (nq + j.sq - 1) / j.sq;
    for (size_t sp = 0; sp < j.nspans; ++sp) {
        const size_t s0 = sp * j.sq, s1 = std::min(nq, s0 + j.sq);
        for (size_t q0 = s0; q0 < s1; q0 += j.pq) j.pieces.push_back(RankJob::Piece{sp, q0, std::min(j.pq, s1 - q0), q0 + j.pq >= s1});
    }
    j.t_prep = now_s();
    cobs_gpu_status rs = COBS_GPU_OK;
    if (j.nspans > 1) {                          // both span buffers are free at the start: order two spans right away
        rs = job_launch_span(b, w, j, 0);
        if (rs == COBS_GPU_OK) rs = job_launch_span(b, w, j, 1);
        j.spans_launched = rs == COBS_GPU_OK ? 2 : 0;
    }
    for (size_t pi = 0; rs == COBS_GPU_OK && pi < std::min<size_t>(2, j.pieces.size()); ++pi) rs = job_issue_piece(b, w, j, pi);
    if (rs != COBS_GPU_OK) {                    // nothing of this batch may still be in flight when the caller sees the error
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(w.copy_stream);
        return rs;
    }
    j.t_prep = now_s() - j.t_prep;
    j.active = true;
    return COBS_GPU_OK;
}

// Queries [q_first, q_first + nq) of the last run of `b`, ranked on the device (the launch half above, unless the caller
// has run it already for exactly these queries, and the drain half).  Query q_first + i's results -- at most
// `limit` (0 = all) -- are appended at hits + *used, hit_offsets[i + 1] = the new *used.  When the
// caller's buffer is too small the offsets keep counting (the caller reports the needed capacity)
// and *overflow is set; hits are then not valid, as in the host path.  COBS_GPU_ERR_UNSUPPORTED (nothing written
// yet): the workspace could not be allocated, rank on the host.
cobs_gpu_status rank_on_device(cobs_gpu_batch* b, size_t q_first, size_t nq, size_t limit, cobs_gpu_hit* hits, size_t cap,
                               size_t* used, size_t* hit_offsets, bool* overflow) {
    cobs_gpu_index* ix = b->ix;
    HIP_TRY(hipSetDevice(ix->device));
    if (!(b->rank && b->rank->job.active && b->rank->job.run_seq == b->run_seq && b->rank->job.q_first == q_first &&
          b->rank->job.nq == nq && b->rank->job.limit == limit)) {
        const cobs_gpu_status ls = rank_launch(b, q_first, nq, limit);
        if (ls != COBS_GPU_OK) return ls;
    }
    RankWork& w = *b->rank;
    RankJob& j = w.job;
    j.active = false;
    hipStream_t st = b->own_stream;
    if (j.parts.empty() || j.per_query == 0) {
        for (size_t i = 0; i < nq; ++i) hit_offsets[i + 1] = *used;
        return COBS_GPU_OK;
    }
    const size_t stride = j.stride;
    // A large result range the caller has not touched yet is written through first-touch page faults (75 000 of them
    // for 256 queries x 100 000 documents): transparent huge pages, where the host offers them on request, cut that
    // to 150 (probe: 5.6 -> 4.9 ms per call into a fresh array, scripts/probes/fresh_buffer_probe.py; a kept array: 3.4).
    // Advisory and harmless on a range that is already populated.
    if (hits && *used < cap) {
        const size_t span = std::min(cap - *used, nq * stride) * sizeof(cobs_gpu_hit);
        if (span >= (32u << 20)) {
            const uintptr_t a0 = (reinterpret_cast<uintptr_t>(hits + *used) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
            const uintptr_t a1 = (reinterpret_cast<uintptr_t>(hits + *used) + span) & ~(uintptr_t)((2u << 20) - 1);
            if (a1 > a0) (void)madvise(reinterpret_cast<void*>(a0), a1 - a0, MADV_HUGEPAGE);
        }
    }
    constexpr size_t kDepth = RankWork::kDepth;
    const bool trace = ix->tune.trace;
    double t_wait = 0, t_copy = 0;
    cobs_gpu_status rs = COBS_GPU_OK;
    for (size_t wi = 0; wi < j.pieces.size(); ++wi) {
        // two pieces ahead: while the host expands piece wi out of its landing buffer, piece wi+1 crosses PCIe and
        // piece wi+2 is queued behind it (into the landing buffer the host left at iteration wi-1)
        if (wi + 2 < j.pieces.size() && (rs = job_issue_piece(b, w, j, wi + 2)) != COBS_GPU_OK) break;
        const int s = (int)(wi % kDepth);
        double t0 = now_s();
        if (hipEventSynchronize(w.landed[s]) != hipSuccess) { rs = hip_fail(hipGetLastError(), "rank window"); break; }
        t_wait += now_s() - t0;
        t0 = now_s();
        const RankJob::Piece& wn = j.pieces[wi];
        const uint32_t* cnt = reinterpret_cast<const uint32_t*>(w.land[s].p + j.land_bytes);
        const uint8_t* recs = w.land[s].p;
        const uint32_t* bins_at = reinterpret_cast<const uint32_t*>(recs + j.pq * j.words * sizeof(uint32_t));
        bool full = true;
        for (size_t i = 0; i < wn.n; ++i) full = full && cnt[i] == stride;
        if (full && !*overflow && wn.n * stride <= cap - *used) {
            // every query of the piece yields `stride` results (the default call): one block
            if (j.slim)
                expand_slim(expand_pool(ix->device), hits + *used, reinterpret_cast<const uint32_t*>(recs), bins_at, wn.n, stride,
                            j.words, j.pack_bits, j.nbins, j.parts);
            else
                expand_records(expand_pool(ix->device), hits + *used, recs, wn.n * stride, j.pack_bits, j.parts);
            for (size_t i = 0; i < wn.n; ++i) {
                *used += stride;
                hit_offsets[wn.q0 + i + 1] = *used;
            }
            t_copy += now_s() - t0;
            continue;
        }
        for (size_t i = 0; i < wn.n; ++i) {
            const size_t n = cnt[i];
            if (!*overflow && n <= cap - *used) {
                if (j.slim)
                    expand_slim(expand_pool(ix->device), hits + *used, reinterpret_cast<const uint32_t*>(recs) + i * j.words,
                                bins_at + i * j.nbins, 1, n, j.words, j.pack_bits, j.nbins, j.parts);
                else
                    expand_records(expand_pool(ix->device), hits + *used, recs + i * stride * j.rec, n, j.pack_bits, j.parts);
            } else
                *overflow = true;
            *used += n;
            hit_offsets[wn.q0 + i + 1] = *used;
        }
    }
    if (trace)
        std::fprintf(stderr, "[cobs_gpu] device ranking: %zu queries x %zu records in %zu pieces (%s, %zu bytes per query); first "
                     "launches %.3f ms, waiting for windows %.3f ms, copying out of the landing buffers %.3f ms\n",
                     nq, stride, j.pieces.size(), j.slim ? "slot streams + score counts" : j.pack_bits ? "4-byte records" : "8-byte records",
                     j.qbytes, j.t_prep * 1e3, t_wait * 1e3, t_copy * 1e3);
    if (rs != COBS_GPU_OK) {                    // nothing of this batch may still be in flight when the caller sees the error
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(w.copy_stream);
    }
    return rs;
}

}  // namespace cobs_amd
