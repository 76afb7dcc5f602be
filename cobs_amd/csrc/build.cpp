// cobs_amd/csrc/build.cpp -- GPU index construction: the step in front
// of the query path.  Restates on the device what the reference does per document with
// process_term / set_bit (cobs/construction/classic_index.cpp:40-130) and writes files in
// the reference's formats (cobs/file/classic_index_header.cpp:26-37,
// cobs/file/compact_index_header.cpp:20-43), so that `cobs query`, the reference's tests
// and this engine can read them.  Documents come either already parsed (a document = its
// sequences joined by '\n') or as a document list whose files the library reads itself
// (documents.cpp): host threads parse the next window of documents into a pinned staging buffer
// while the GPU hashes the previous one.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "documents.hpp"
#include "engine.hpp"

using namespace cobs_amd;

__attribute__((visibility("hidden"))) cobs_gpu_status cobs_gpu_set_error(cobs_gpu_status st, const char* msg);   // engine.cpp

namespace {

#define BUILD_TRY(expr)                                                             \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            const bool nodev = _e == hipErrorNoDevice || _e == hipErrorInvalidDevice; \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(_e);      \
            (void)hipGetLastError();                                                \
            return cobs_gpu_set_error(nodev ? COBS_GPU_ERR_NO_DEVICE : COBS_GPU_ERR_HIP, m.c_str()); \
        }                                                                           \
    } while (0)

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
};

// calc_signature_size, cobs/util/calc_signature_size.cpp:17-33 (all arithmetic in double)
uint64_t signature_size_for(uint64_t num_elements, double num_hashes, double fpr) {
    const double ratio = -num_hashes / std::log(1.0 - std::pow(fpr, 1.0 / num_hashes));
    return (uint64_t)std::ceil((double)num_elements * ratio);
}

// number of k-grams of a document given as '\n'-separated sequences
uint64_t count_terms(const char* text, size_t len, uint32_t k) {
    uint64_t total = 0, run = 0;
    for (size_t i = 0; i <= len; ++i) {
        if (i == len || text[i] == '\n') {
            if (run >= k) total += run - k + 1;
            run = 0;
        } else {
            ++run;
        }
    }
    return total;
}

struct Params {
    uint32_t term_size = 31, canonicalize = 1, num_hashes = 1;
    double fpr = 0.3;
    uint64_t signature_size = 0, page_size = 0;
    int device = -1;
    const uint64_t* doc_terms = nullptr;
    uint64_t text_batch = 0;
    uint32_t set_bits_mode = 0;
};

cobs_gpu_status read_params(const cobs_gpu_build_params* p, Params& out) {
    if (p) {
        if (p->struct_size < offsetof(cobs_gpu_build_params, doc_terms))
            return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "cobs_gpu_build_params.struct_size is too small");
        if (p->struct_size >= offsetof(cobs_gpu_build_params, set_bits_mode)) out.doc_terms = p->doc_terms;
        if (p->struct_size >= sizeof(cobs_gpu_build_params)) out.set_bits_mode = p->set_bits_mode;
        if (out.set_bits_mode > 2) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "set_bits_mode: 0, 1 or 2");
        out.term_size = p->term_size;
        out.canonicalize = p->canonicalize;
        out.num_hashes = p->num_hashes;
        out.fpr = p->false_positive_rate;
        out.signature_size = p->signature_size;
        out.page_size = p->page_size;
        out.device = p->device;
        out.text_batch = p->text_batch_bytes;
    }
    if (out.term_size == 0 || out.num_hashes == 0 || out.num_hashes > 64 || out.canonicalize > 1)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad term_size / num_hashes / canonicalize");
    if (out.signature_size == 0 && !(out.fpr > 0.0 && out.fpr < 1.0))
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "false_positive_rate must be in (0, 1)");
    return COBS_GPU_OK;
}

cobs_gpu_status pick_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return cobs_gpu_set_error(COBS_GPU_ERR_NO_DEVICE, "no HIP device visible; libcobs_gpu has no CPU fallback");
    }
    if (device >= 0) {
        if (device >= n) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "device ordinal out of range");
        BUILD_TRY(hipSetDevice(device));
    }
    return COBS_GPU_OK;
}

// ---- where the documents come from ---------------------------------------------------------------
struct DocSource {
    virtual ~DocSource() = default;
    virtual size_t size() const = 0;
    virtual const char* name(size_t d) const = 0;
    virtual uint64_t terms(size_t d, uint32_t k) const = 0;          // what sizes a signature
    virtual uint64_t text_bound(size_t d, uint32_t k) const = 0;     // upper bound of the document's term text
    virtual bool parses() const = 0;                                 // loading reads and parses files
    // the document's term text into `out` (a span of the staging buffer), its stretches into `segs`
    virtual cobs_gpu_status load(size_t d, uint32_t k, TermSink& out, std::vector<TermSeg>& segs,
                                 std::string& scratch) const = 0;
};

// documents handed over as texts (cobs_gpu_build_classic / _compact / _index)
struct ArraySource final : DocSource {
    const char* const* names;
    const char* const* texts;
    const size_t* lens;
    size_t n;
    const uint64_t* doc_terms;
    ArraySource(const char* const* nm, const char* const* tx, const size_t* ln, size_t nd, const uint64_t* dt)
        : names(nm), texts(tx), lens(ln), n(nd), doc_terms(dt) {}
    size_t size() const override { return n; }
    const char* name(size_t d) const override { return names[d]; }
    uint64_t terms(size_t d, uint32_t k) const override { return doc_terms ? doc_terms[d] : count_terms(texts[d], lens[d], k); }
    uint64_t text_bound(size_t d, uint32_t) const override { return lens[d] + 1; }
    bool parses() const override { return false; }
    cobs_gpu_status load(size_t d, uint32_t, TermSink& out, std::vector<TermSeg>& segs, std::string&) const override {
        out.put(texts[d], lens[d]);
        out.put('\n');
        segs.push_back(TermSeg{0, (uint64_t)lens[d] + 1, false});
        return COBS_GPU_OK;
    }
};

// a document list (cobs_gpu_build_*_list)
struct ListSource final : DocSource {
    const std::vector<DocEntry>& list;
    explicit ListSource(const std::vector<DocEntry>& l) : list(l) {}
    size_t size() const override { return list.size(); }
    const char* name(size_t d) const override { return list[d].name.c_str(); }
    uint64_t terms(size_t d, uint32_t k) const override { return num_terms(list[d], k); }
    uint64_t text_bound(size_t d, uint32_t k) const override { return term_text_bound(list[d], k); }
    bool parses() const override { return true; }
    cobs_gpu_status load(size_t d, uint32_t k, TermSink& out, std::vector<TermSeg>& segs, std::string& scratch) const override {
        return load_terms(list[d], k, out, segs, scratch);
    }
};

// Documents reach the device in batches of at most this many bytes of term text (the reference
// batches documents by a memory budget too: classic_index.cpp:565-659 builds one small index per
// batch and interleaves them afterwards; here every batch sets its bits straight at the documents'
// final columns of the one matrix in HBM, so there is nothing to combine).
constexpr uint64_t kTextBatchBytes = 256ull << 20;
constexpr size_t kTextPad = 64;                 // readable bytes behind the text (build_kernel loads dwords)
// staging sets of a build: one being parsed into, one on its way over PCIe, one being hashed (with two,
// parsing waits for the kernel of the batch before last: 8.2 ms per 256 MiB batch instead of 6)
constexpr int kStages = 3;

// One of the staging sets of a build: pinned term text + stretch tables, their device copies,
// the event that tells when the GPU is done with them.  Host threads parse documents straight into
// `text` (every document of a batch owns a span sized by its text bound; what it leaves unused is
// a gap stretch the kernel skips), so a character is written once between the file and the H2D copy.
struct Stage {
    PinnedBuf<uint8_t> text;
    PinnedBuf<uint64_t> seg_off;
    PinnedBuf<uint32_t> seg_col;
    DevBuf<uint8_t> d_text;
    DevBuf<uint64_t> d_off;
    DevBuf<uint32_t> d_col;
    hipEvent_t done = nullptr, copied = nullptr;
    bool busy = false;
    ~Stage() {
        if (done) (void)hipEventDestroy(done);
        if (copied) (void)hipEventDestroy(copied);
    }
};

// Staging memory outlives a build: pinning 2 x 256 MiB costs more than hashing them.  The sets are
// checked out per build and handed back; never freed (a static destructor would run after the HIP
// runtime's own).
struct StagePool {
    std::mutex mu;
    std::vector<Stage*> idle;
    std::vector<DevBuf<uint8_t>*> idle_planes;     // byte-map planes (one buffer per build in flight)
    int device = -1;
    DevBuf<uint8_t>* take_planes(int dev) {
        std::lock_guard<std::mutex> g(mu);
        if (device != dev) {
            for (Stage* s : idle) delete s;
            idle.clear();
            for (auto* b : idle_planes) delete b;
            idle_planes.clear();
            device = dev;
        }
        if (idle_planes.empty()) return new DevBuf<uint8_t>;
        DevBuf<uint8_t>* b = idle_planes.back();
        idle_planes.pop_back();
        return b;
    }
    void give_planes(DevBuf<uint8_t>* b) {
        std::lock_guard<std::mutex> g(mu);
        idle_planes.push_back(b);
    }
    void release_idle() {
        std::lock_guard<std::mutex> g(mu);
        for (Stage* s : idle) delete s;
        idle.clear();
        for (auto* b : idle_planes) delete b;
        idle_planes.clear();
    }
    Stage* take(int dev) {
        std::lock_guard<std::mutex> g(mu);
        if (device != dev) {                    // buffers belong to the device they were made on
            for (Stage* s : idle) delete s;
            idle.clear();
            for (auto* b : idle_planes) delete b;
            idle_planes.clear();
            device = dev;
        }
        if (idle.empty()) return new Stage;
        Stage* s = idle.back();
        idle.pop_back();
        return s;
    }
    void give(Stage* s) {
        std::lock_guard<std::mutex> g(mu);
        idle.push_back(s);
    }
};
StagePool& stage_pool() {
    static StagePool* pool = new StagePool;
    return *pool;
}

// Host threads that stay up for a whole build: a batch hands them one job (parse the documents of
// the batch), run() returns when every worker has finished it.  Spawning 64-128 threads per batch
// cost 1-2 ms of the ~6 ms a batch has.
class WorkerPool {
public:
    explicit WorkerPool(size_t n) {
        for (size_t t = 0; t < n; ++t) threads_.emplace_back([this, t]() { loop(t); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    size_t size() const { return threads_.size(); }
    void run(const std::function<void(size_t)>& fn) {
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = &fn;
            done_ = 0;
            ++gen_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return done_ == threads_.size(); });
        job_ = nullptr;
    }

private:
    void loop(size_t tid) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (quit_) return;
                job = job_;
            }
            (*job)(tid);
            {
                std::lock_guard<std::mutex> g(mu_);
                ++done_;
            }
            cv_done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(size_t)>* job_ = nullptr;
    uint64_t gen_ = 0;
    size_t done_ = 0;
    bool quit_ = false;
};

struct Slot {                                   // one document of a batch
    size_t doc_col;                             // its column
    size_t src;                                 // its index in the source
    uint64_t begin, cap;                        // its span of the staging text
    uint64_t used = 0;
    std::vector<TermSeg> segs;
    cobs_gpu_status status = COBS_GPU_OK;
    std::string error;
};

// What one build call keeps across its matrices (a compact index is one build_into per
// sub-index): the two streams, the staging sets and byte planes checked out of the pool, the
// parser threads and their scratch buffers.
struct BuildContext {
    hipStream_t stream = nullptr, copy_stream = nullptr;   // kernels | uploads (batch i + 1 beside the kernel of batch i)
    Stage* st[kStages] = {};
    DevBuf<uint8_t>* planes = nullptr;
    std::unique_ptr<WorkerPool> workers;                   // created by the first batch with more than one document
    std::vector<std::string> scratch;                      // the file being parsed, one per worker, reused
    size_t max_threads = 1;
    bool ready = false;
    // writing an index file: the matrix of the current (sub-)index and the two pinned buffers its
    // rows leave through -- allocated once per build, not once per sub-index
    DevBuf<uint8_t> matrix;
    PinnedBuf<uint8_t> out_host[2];

    cobs_gpu_status init(bool parses) {
        if (ready) return COBS_GPU_OK;
        int dev = 0;
        BUILD_TRY(hipGetDevice(&dev));
        BUILD_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        BUILD_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        for (Stage*& s : st) {
            s = stage_pool().take(dev);
            if (!s->done) BUILD_TRY(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
            if (!s->copied) BUILD_TRY(hipEventCreateWithFlags(&s->copied, hipEventDisableTiming));
        }
        planes = stage_pool().take_planes(dev);
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        max_threads = parses ? std::min<size_t>(hw, 128) : std::min<size_t>(hw, 8);
        scratch.resize(max_threads);
        ready = true;
        return COBS_GPU_OK;
    }
    BuildContext() = default;
    BuildContext(const BuildContext&) = delete;
    BuildContext& operator=(const BuildContext&) = delete;
    ~BuildContext() {
        workers.reset();
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        // handed back last-to-first: the next build takes the set that was used first (and has its
        // buffers) as its first set again
        for (int i = kStages - 1; i >= 0; --i)
            if (st[i]) { st[i]->busy = false; stage_pool().give(st[i]); }
        if (planes) stage_pool().give_planes(planes);
    }
};

// Set the bits of documents docs[0..n) -- document docs[i] in column i -- in a zeroed device
// matrix of `sig` rows, `row_bytes` (multiple of 4) apart.
cobs_gpu_status build_into(BuildContext& ctx, uint32_t* d_matrix, uint64_t sig, uint64_t row_bytes, const DocSource& src,
                           const size_t* docs, size_t n, const Params& pr) {
    const uint64_t text_batch = pr.text_batch ? pr.text_batch : kTextBatchBytes;
    if (sig == 0 || sig > (1ull << 46)) return cobs_gpu_set_error(COBS_GPU_ERR_UNSUPPORTED, "signature_size must be in 1..2^46");
    if (n >= (kBuildRawStretch - 1)) return cobs_gpu_set_error(COBS_GPU_ERR_UNSUPPORTED, "too many documents in one matrix");
    cobs_gpu_status ist = ctx.init(src.parses());
    if (ist != COBS_GPU_OK) return ist;
    hipStream_t stream = ctx.stream, copy_stream = ctx.copy_stream;
    struct { DevBuf<uint8_t>* planes; } guard{ctx.planes};
    Stage* const* stage = ctx.st;
    // byte-map planes of a batch: one byte per (document of the batch, signature row)
    const uint64_t bm_stride = (sig + 255) / 256 * 256;
    constexpr uint64_t kPlaneBudget = 3ull << 30;

    // COBS_GPU_BUILD_TRACE=1: where the host side of a build spends its time (stderr, one line per build)
    static const bool trace = std::getenv("COBS_GPU_BUILD_TRACE") != nullptr;
    double t_wait = 0, t_parse = 0, t_table = 0, t_issue = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const size_t max_threads = ctx.max_threads;
    std::vector<Slot> slots;
    std::vector<std::string>& scratch = ctx.scratch;
    std::unique_ptr<WorkerPool>& workers = ctx.workers;
    int cur = 0;
    for (size_t b0 = 0; b0 < n;) {
        // documents [b0, b1): as many as fit the batch by their text bounds (at least one)
        slots.clear();
        uint64_t total = 0;
        size_t b1 = b0;
        while (b1 < n) {
            const uint64_t bound = src.text_bound(docs[b1], pr.term_size);
            if (b1 > b0 && total + bound > text_batch) break;
            Slot sl;
            sl.doc_col = b1;
            sl.src = docs[b1];
            sl.begin = total;
            sl.cap = bound;
            slots.push_back(std::move(sl));
            total += bound;
            ++b1;
        }
        Stage& s = *stage[cur];
        double t0 = now();
        if (s.busy) { BUILD_TRY(hipEventSynchronize(s.done)); s.busy = false; }
        BUILD_TRY(s.text.reserve((size_t)std::max<uint64_t>(total, text_batch) + kTextPad));
        t_wait += now() - t0;
        t0 = now();
        // parse: every worker takes the next document and writes its term text into its span
        std::atomic<size_t> next{0};
        auto work = [&](size_t tid) {
            for (size_t i; (i = next.fetch_add(1)) < slots.size();) {
                Slot& sl = slots[i];
                TermSink sink;
                sink.data = reinterpret_cast<char*>(s.text.p) + sl.begin;
                sink.cap = (size_t)sl.cap;
                sl.status = src.load(sl.src, pr.term_size, sink, sl.segs, scratch[tid]);
                if (sl.status == COBS_GPU_OK && sink.overflow) {
                    sl.status = COBS_GPU_ERR_FORMAT;
                    sl.error = "a document outgrew the size its list entry recorded";
                } else if (sl.status != COBS_GPU_OK) {
                    sl.error = cobs_gpu_last_error();
                }
                sl.used = sink.size;
            }
        };
        if (max_threads <= 1 || slots.size() <= 1) {
            work(0);
        } else {
            // as many workers as the largest batch so far has documents (a compact build may start
            // with a small sub-index and go on to large ones)
            const size_t want = std::min(max_threads, std::max<size_t>(slots.size(), 8));
            if (!workers || workers->size() < want) workers.reset(new WorkerPool(want));
            workers->run(work);
        }
        t_parse += now() - t0;
        t0 = now();
        // the stretch table: a document's stretches, the rest of its span as a gap
        size_t nsegs = 0;
        for (const Slot& sl : slots) {
            if (sl.status != COBS_GPU_OK) return cobs_gpu_set_error(sl.status, sl.error.c_str());
            nsegs += 2 * sl.segs.size() + 2;
        }
        BUILD_TRY(s.seg_off.reserve(nsegs + 1));
        BUILD_TRY(s.seg_col.reserve(nsegs + 1));
        size_t ns = 0;
        auto add = [&](uint64_t off, uint32_t col) {
            if (ns && s.seg_off.p[ns - 1] == off) { s.seg_col.p[ns - 1] = col; return; }   // the previous one was empty
            s.seg_off.p[ns] = off;
            s.seg_col.p[ns] = col;
            ++ns;
        };
        for (const Slot& sl : slots) {
            uint64_t at = sl.begin;                         // everything before `at` is described
            for (const TermSeg& g : sl.segs) {
                if (g.len == 0) continue;
                if (sl.begin + g.begin > at) add(at, kBuildGapStretch);
                add(sl.begin + g.begin, (uint32_t)sl.doc_col | (g.raw ? kBuildRawStretch : 0u));
                at = sl.begin + g.begin + g.len;
            }
            if (at < sl.begin + sl.cap) add(at, kBuildGapStretch);
        }
        s.seg_off.p[ns] = total;
        t_table += now() - t0;
        t0 = now();
        if (ns && total) {
            BUILD_TRY(s.d_text.reserve(s.text.cap));
            BUILD_TRY(s.d_off.reserve(s.seg_off.cap));
            BUILD_TRY(s.d_col.reserve(s.seg_col.cap));
            BUILD_TRY(hipMemcpyAsync(s.d_text.p, s.text.p, (size_t)total, hipMemcpyHostToDevice, copy_stream));
            BUILD_TRY(hipMemcpyAsync(s.d_off.p, s.seg_off.p, (ns + 1) * 8, hipMemcpyHostToDevice, copy_stream));
            BUILD_TRY(hipMemcpyAsync(s.d_col.p, s.seg_col.p, ns * 4, hipMemcpyHostToDevice, copy_stream));
            BUILD_TRY(hipEventRecord(s.copied, copy_stream));
            BUILD_TRY(hipStreamWaitEvent(stream, s.copied, 0));
            BuildArgs a;
            a.text = s.d_text.p;
            a.seg_off = s.d_off.p;
            a.seg_col = s.d_col.p;
            a.matrix = d_matrix;
            a.signature_size = sig;
            a.magic = ~0ull / sig;
            a.row_bytes = row_bytes;
            a.nsegs = (uint32_t)ns;
            a.term_size = pr.term_size;
            a.canonicalize = pr.canonicalize;
            a.num_hashes = pr.num_hashes;
            a.bytemap = nullptr;
            a.bm_stride = bm_stride;
            a.col_base = (uint32_t)b0;
            // byte stores into per-document planes + one packing pass beat the scattered atomics
            // whenever the planes of the batch are affordable and there is enough text to pay for
            // zeroing and packing them (mode 2 forces them, mode 1 the atomics)
            const uint64_t plane_bytes = (uint64_t)(b1 - b0) * bm_stride;
            bool planes = pr.set_bits_mode == 2 ||
                          (pr.set_bits_mode == 0 && plane_bytes <= kPlaneBudget && plane_bytes <= 16 * total);
            if (planes && pr.set_bits_mode == 0 && guard.planes->reserve((size_t)plane_bytes) != hipSuccess) {
                (void)hipGetLastError();            // no room for the planes next to the index: the atomics need none
                planes = false;
            }
            if (planes) {
                BUILD_TRY(guard.planes->reserve((size_t)plane_bytes));
                BUILD_TRY(hipMemsetAsync(guard.planes->p, 0, (size_t)plane_bytes, stream));
                a.bytemap = guard.planes->p;
            }
            BUILD_TRY(launch_build(a, total, stream));
            if (planes) {
                PackArgs pk;
                pk.bytemap = guard.planes->p;
                pk.bm_stride = bm_stride;
                pk.matrix = d_matrix;
                pk.row_bytes = row_bytes;
                pk.rows = sig;
                pk.col_base = (uint32_t)b0;
                pk.ndocs = (uint32_t)(b1 - b0);
                BUILD_TRY(launch_pack_bytemap(pk, stream));
            }
            BUILD_TRY(hipEventRecord(s.done, stream));
            s.busy = true;
        }
        t_issue += now() - t0;
        cur = (cur + 1) % kStages;                          // the next set is parsed into while this one is uploaded and hashed
        b0 = b1;
    }
    const double t0 = now();
    BUILD_TRY(hipStreamSynchronize(stream));
    if (trace)
        std::fprintf(stderr, "[cobs_gpu build] %zu documents: wait for a staging set %.3f s, parse %.3f s, stretch table %.3f s, "
                             "issue %.3f s, drain %.3f s\n", n, t_wait, t_parse, t_table, t_issue, now() - t0);
    return COBS_GPU_OK;
}

bool write_all(FILE* f, const void* p, size_t n);

// rows [0, rows) of a device matrix (pitch bytes apart) -> file, row_size bytes each, through two
// pinned buffers: the host writes chunk i while the device sends chunk i + 1
cobs_gpu_status stream_rows_to_file(FILE* f, const uint8_t* d_matrix, uint64_t pitch, uint64_t row_size, uint64_t rows,
                                    PinnedBuf<uint8_t> (*keep)[2] = nullptr) {
    if (rows == 0 || row_size == 0) return COBS_GPU_OK;
    // Rows travel as ONE contiguous copy per chunk, padding included (the device pitch is the row
    // size rounded up to 4 bytes): a 2-D copy of millions of 2..11-byte rows -- compact indexes
    // with a small, odd page size -- takes minutes in the runtime.  The padding is dropped on the
    // host, in place, before the chunk is written.
    const uint64_t rows_per = std::max<uint64_t>(1, (128ull << 20) / pitch);
    PinnedBuf<uint8_t> own[2];
    PinnedBuf<uint8_t>* host = keep ? *keep : own;
    for (int i = 0; i < 2; ++i) BUILD_TRY(host[i].reserve((size_t)(std::min(rows_per, rows) * pitch)));
    hipStream_t stream = nullptr;
    BUILD_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    // rows p[r * pitch .. + row_size) -> p[r * row_size ..): slices of rows are packed by a few threads
    // (each inside its own slice), then the packed slices are moved together
    auto squeeze = [&](uint8_t* p, uint64_t n) {
        if (pitch == row_size) return;
        const uint64_t nthr = std::min<uint64_t>(8, std::max<uint64_t>(1, n >> 20));
        auto pack = [&](uint64_t r0, uint64_t r1) {     // rows [r0, r1) packed at p + r0 * pitch
            uint8_t* base = p + r0 * pitch;
            for (uint64_t r = 1; r < r1 - r0; ++r) std::memmove(base + r * row_size, base + r * pitch, (size_t)row_size);
        };
        if (nthr == 1) { pack(0, n); return; }
        std::vector<std::thread> pool;
        for (uint64_t t = 0; t < nthr; ++t) pool.emplace_back(pack, n * t / nthr, n * (t + 1) / nthr);
        for (auto& th : pool) th.join();
        for (uint64_t t = 1; t < nthr; ++t) {
            const uint64_t r0 = n * t / nthr, r1 = n * (t + 1) / nthr;
            std::memmove(p + r0 * row_size, p + r0 * pitch, (size_t)((r1 - r0) * row_size));
        }
    };
    int cur = 0;
    uint64_t pending = 0;
    for (uint64_t r = 0; r < rows; r += rows_per) {
        const uint64_t n = std::min(rows_per, rows - r);
        BUILD_TRY(hipMemcpyAsync(host[cur].p, d_matrix + r * pitch, (size_t)(n * pitch), hipMemcpyDeviceToHost, stream));
        if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
            return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        BUILD_TRY(hipStreamSynchronize(stream));
        squeeze((uint8_t*)host[cur].p, n);
        pending = n * row_size;
        cur ^= 1;
    }
    if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
        return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

bool write_all(FILE* f, const void* p, size_t n) { return n == 0 || std::fwrite(p, 1, n, f) == n; }

template <typename T>
void put(std::string& s, T v) { s.append(reinterpret_cast<const char*>(&v), sizeof v); }

// ---- the layout of an index: which documents, in which order, in which sub-index -----------------
struct Group {
    std::vector<size_t> docs;       // source documents in column order
    uint64_t sig = 0;
};
struct Layout {
    bool compact = false;
    uint64_t page_size = 0;         // compact only
    std::vector<Group> groups;      // classic: one
};

uint64_t default_page_size(size_t ndocs) {      // compact_construct, compact_index.cpp:184-189
    const uint64_t v = (uint64_t)std::sqrt((double)(ndocs / 8));
    uint64_t p2 = 1;
    while (p2 < v) p2 <<= 1;
    return std::min<uint64_t>(std::max<uint64_t>(v == 0 ? 0 : p2, 8), 4096);
}

// documents already in their final order (the array entry points)
cobs_gpu_status layout_in_order(const DocSource& src, bool compact, const Params& pr, Layout& out) {
    const size_t n = src.size();
    out.compact = compact;
    if (!compact) {
        Group g;
        uint64_t max_terms = 0;
        for (size_t d = 0; d < n; ++d) {
            g.docs.push_back(d);
            if (pr.signature_size == 0) max_terms = std::max(max_terms, src.terms(d, pr.term_size));
        }
        // classic_construct, classic_index.cpp:571-575 (the caller's doc_terms name the largest document)
        g.sig = pr.signature_size ? pr.signature_size : signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
        if (g.sig == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
        out.groups.push_back(std::move(g));
        return COBS_GPU_OK;
    }
    out.page_size = pr.page_size ? pr.page_size : default_page_size(n);
    const size_t per = (size_t)(8 * out.page_size);
    for (size_t g0 = 0; g0 < n; g0 += per) {
        Group g;
        uint64_t max_terms = 0;
        for (size_t d = g0; d < std::min(n, g0 + per); ++d) {
            g.docs.push_back(d);
            max_terms = std::max(max_terms, src.terms(d, pr.term_size));
        }
        if (max_terms == 0) continue;                           // compact_index.cpp:285-286: empty group is dropped
        g.sig = pr.signature_size ? pr.signature_size : signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
        out.groups.push_back(std::move(g));
    }
    if (out.groups.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
    return COBS_GPU_OK;
}

// a document list, ordered and sized the way classic_construct / compact_construct do it
cobs_gpu_status layout_of_list(const std::vector<DocEntry>& list, bool compact, const Params& pr, Layout& out) {
    const size_t n = list.size();
    out.compact = compact;
    if (!compact) {
        Group g;
        for (size_t d = 0; d < n; ++d) g.docs.push_back(d);     // list order (process_batches_parallel, document_list.hpp:475-514)
        g.sig = pr.signature_size;
        if (g.sig == 0) {
            // get_max_file_size, classic_index.cpp:521-563: the num_terms of the largest document by
            // (size, path) -- std::max_element, the first of equals
            size_t big = 0;
            for (size_t d = 1; d < n; ++d)
                if (std::tie(list[big].size, list[big].path) < std::tie(list[d].size, list[d].path)) big = d;
            g.sig = signature_size_for(num_terms(list[big], pr.term_size), (double)pr.num_hashes, pr.fpr);
        }
        if (g.sig == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
        out.groups.push_back(std::move(g));
        return COBS_GPU_OK;
    }
    std::vector<size_t> order(n);
    for (size_t d = 0; d < n; ++d) order[d] = d;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {      // sort_by_size, compact_index.cpp:182
        return std::tie(list[a].size, list[a].path) < std::tie(list[b].size, list[b].path);
    });
    out.page_size = pr.page_size ? pr.page_size : default_page_size(n);
    const size_t per = (size_t)(8 * out.page_size);
    for (size_t g0 = 0; g0 < n; g0 += per) {
        Group g;
        g.docs.assign(order.begin() + g0, order.begin() + std::min(n, g0 + per));
        // DocumentList batch_list(files), compact_index.cpp:315: sorted by (path, sub-document)
        std::stable_sort(g.docs.begin(), g.docs.end(), [&](size_t a, size_t b) {
            return std::tie(list[a].path, list[a].subdoc_index) < std::tie(list[b].path, list[b].subdoc_index);
        });
        uint64_t max_terms = 0;
        for (size_t d : g.docs) max_terms = std::max(max_terms, num_terms(list[d], pr.term_size));
        if (max_terms == 0) continue;                           // :285-286
        g.sig = pr.signature_size ? pr.signature_size : signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
        out.groups.push_back(std::move(g));
    }
    if (out.groups.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
    return COBS_GPU_OK;
}

// ---- layout + source -> index file ---------------------------------------------------------------
cobs_gpu_status write_index_file(const DocSource& src, const Layout& lay, const Params& pr, const char* out_path) {
    std::string h;
    uint64_t row_size, row_bytes;
    if (!lay.compact) {
        const Group& g = lay.groups[0];
        h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, pr.term_size);
        put<uint8_t>(h, (uint8_t)pr.canonicalize);
        put<uint32_t>(h, (uint32_t)g.docs.size());
        put<uint64_t>(h, g.sig);
        put<uint64_t>(h, (uint64_t)pr.num_hashes);
        for (size_t d : g.docs) { h += src.name(d); h += '\n'; }
        h += "CLASSIC_INDEX";
        row_size = (g.docs.size() + 7) / 8;
    } else {
        const uint64_t ps = lay.page_size;
        size_t kept = 0;
        for (const Group& g : lay.groups) kept += g.docs.size();
        h = "COBS:COMPACT_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, pr.term_size);
        put<uint8_t>(h, (uint8_t)pr.canonicalize);
        put<uint32_t>(h, (uint32_t)lay.groups.size());
        put<uint32_t>(h, (uint32_t)kept);
        put<uint64_t>(h, ps);
        for (const Group& g : lay.groups) { put<uint64_t>(h, g.sig); put<uint64_t>(h, (uint64_t)pr.num_hashes); }
        for (const Group& g : lay.groups)
            for (size_t d : g.docs) { h += src.name(d); h += '\n'; }
        const uint64_t pad = (ps - ((h.size() + 13) % ps)) % ps;     // data starts page-aligned
        h.append((size_t)pad, '\0');
        h += "COMPACT_INDEX";
        row_size = ps;                                               // rows padded to page_size (:116-156)
    }
    row_bytes = (row_size + 3) / 4 * 4;
    FILE* f = std::fopen(out_path, "wb");
    if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
    struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
    if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    // one (sub-)index after the other: built in HBM (288 GB; never in host RAM), streamed out
    BuildContext ctx;
    for (const Group& g : lay.groups) {
        BUILD_TRY(ctx.matrix.reserve((size_t)(g.sig * row_bytes)));
        BUILD_TRY(hipMemset(ctx.matrix.p, 0, (size_t)(g.sig * row_bytes)));
        BUILD_TRY(hipStreamSynchronize(nullptr));       // the build runs on non-blocking streams: no implicit order with the null stream
        cobs_gpu_status st = build_into(ctx, (uint32_t*)ctx.matrix.p, g.sig, row_bytes, src, g.docs.data(), g.docs.size(), pr);
        if (st != COBS_GPU_OK) return st;
        st = stream_rows_to_file(f, ctx.matrix.p, row_bytes, row_size, g.sig, &ctx.out_host);
        if (st != COBS_GPU_OK) return st;
    }
    closer.f = nullptr;
    if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

// ---- layout + source -> resident query handle ------------------------------------------------------
cobs_gpu_status build_resident(const DocSource& src, const Layout& lay, const Params& pr, const cobs_gpu_options* opts,
                               cobs_gpu_index** out) {
    IndexMeta meta;
    meta.kind = lay.compact ? IndexKind::Compact : IndexKind::Classic;
    meta.term_size = pr.term_size;
    meta.canonicalize = (uint8_t)pr.canonicalize;
    meta.num_hashes = pr.num_hashes;
    if (lay.compact) meta.header_page_size = lay.page_size;
    for (const Group& g : lay.groups) {
        meta.signature_sizes.push_back(g.sig);
        for (size_t d : g.docs) meta.doc_names.emplace_back(src.name(d));
    }
    cobs_gpu_options o{};
    if (opts) std::memcpy(&o, opts, std::min<size_t>(opts->struct_size, sizeof o));
    o.struct_size = sizeof o;
    if (pr.device >= 0) o.device = pr.device;
    else if (!opts) o.device = -1;
    cobs_gpu_index* ix = nullptr;
    cobs_gpu_status st = open_zeroed(std::move(meta), &o, &ix);
    if (st != COBS_GPU_OK) return st;
    std::unique_ptr<cobs_gpu_index, void (*)(cobs_gpu_index*)> guard(ix, cobs_gpu_close);
    Part& pt = ix->parts[0];
    BuildContext ctx;
    BUILD_TRY(hipStreamSynchronize(nullptr));           // open_zeroed cleared the matrix on the null stream; the build's streams do not wait for it
    for (Chunk& c : pt.chunks)
        for (size_t i = 0; i < c.vp.size(); ++i) {
            const Group& g = lay.groups[c.vp[i].fp];
            st = build_into(ctx, reinterpret_cast<uint32_t*>(c.d_data + c.pages[i].base), c.pages[i].sig, c.pitch, src,
                            g.docs.data(), g.docs.size(), pr);
            if (st != COBS_GPU_OK) return st;
        }
    *out = guard.release();
    return COBS_GPU_OK;
}

cobs_gpu_status build_from_arrays(bool compact, const char* const* names, const char* const* texts, const size_t* lens,
                                  size_t ndocs, const cobs_gpu_build_params* params, const char* out_path,
                                  const cobs_gpu_options* opts, cobs_gpu_index** out) {
    return guarded([&]() -> cobs_gpu_status {
        Params pr;
        cobs_gpu_status st = read_params(params, pr);
        if (st != COBS_GPU_OK) return st;
        ArraySource src(names, texts, lens, ndocs, pr.doc_terms);
        Layout lay;
        if ((st = layout_in_order(src, compact, pr, lay)) != COBS_GPU_OK) return st;
        if (out) return build_resident(src, lay, pr, opts, out);
        if ((st = pick_device(pr.device)) != COBS_GPU_OK) return st;
        return write_index_file(src, lay, pr, out_path);
    });
}

cobs_gpu_status build_from_list(bool compact, const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                const char* out_path, const cobs_gpu_options* opts, cobs_gpu_index** out) {
    return guarded([&]() -> cobs_gpu_status {
        Params pr;
        cobs_gpu_status st = read_params(params, pr);
        if (st != COBS_GPU_OK) return st;
        ListSource src(dl->list);
        Layout lay;
        if ((st = layout_of_list(dl->list, compact, pr, lay)) != COBS_GPU_OK) return st;
        if (out) return build_resident(src, lay, pr, opts, out);
        if ((st = pick_device(pr.device)) != COBS_GPU_OK) return st;
        return write_index_file(src, lay, pr, out_path);
    });
}

}  // namespace

extern "C" {

void cobs_gpu_build_release_buffers(void) { stage_pool().release_idle(); }

cobs_gpu_status cobs_gpu_build_classic(const char* const* names, const char* const* texts, const size_t* lens,
                                       size_t ndocs, const cobs_gpu_build_params* params, const char* out_path) {
    if (!names || !texts || !lens || !out_path || ndocs == 0)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return build_from_arrays(false, names, texts, lens, ndocs, params, out_path, nullptr, nullptr);
}

cobs_gpu_status cobs_gpu_build_compact(const char* const* names, const char* const* texts, const size_t* lens,
                                       size_t ndocs, const cobs_gpu_build_params* params, const char* out_path) {
    if (!names || !texts || !lens || !out_path || ndocs == 0)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return build_from_arrays(true, names, texts, lens, ndocs, params, out_path, nullptr, nullptr);
}

// classic_construct / compact_construct straight into a resident query index: the matrix is built
// at the engine's row pitch inside the handle's HBM blob, nothing touches a file or host memory.
cobs_gpu_status cobs_gpu_build_index(uint32_t kind, const char* const* names, const char* const* texts,
                                     const size_t* lens, size_t ndocs, const cobs_gpu_build_params* params,
                                     const cobs_gpu_options* opts, cobs_gpu_index** out) {
    if (!out) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!names || !texts || !lens || ndocs == 0 || kind > 1)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return build_from_arrays(kind == 1, names, texts, lens, ndocs, params, nullptr, opts, out);
}

cobs_gpu_status cobs_gpu_build_classic_list(const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                            const char* out_path) {
    if (!dl || !out_path || dl->list.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return build_from_list(false, dl, params, out_path, nullptr, nullptr);
}

cobs_gpu_status cobs_gpu_build_compact_list(const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                            const char* out_path) {
    if (!dl || !out_path || dl->list.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return build_from_list(true, dl, params, out_path, nullptr, nullptr);
}

cobs_gpu_status cobs_gpu_build_index_list(uint32_t kind, const cobs_gpu_doclist* dl, const cobs_gpu_build_params* params,
                                          const cobs_gpu_options* opts, cobs_gpu_index** out) {
    if (!out) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!dl || dl->list.empty() || kind > 1) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return build_from_list(kind == 1, dl, params, nullptr, opts, out);
}

// ---- document lists --------------------------------------------------------------------------------
cobs_gpu_status cobs_gpu_doclist_create(cobs_gpu_doclist** out) {
    if (!out) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "out is NULL");
    return guarded([&]() -> cobs_gpu_status {
        *out = new cobs_gpu_doclist;
        return COBS_GPU_OK;
    });
}

void cobs_gpu_doclist_free(cobs_gpu_doclist* dl) { delete dl; }

cobs_gpu_status cobs_gpu_doclist_add(cobs_gpu_doclist* dl, const char* path) {
    if (!dl || !path) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status { return load_entries(path, dl->list); });
}

cobs_gpu_status cobs_gpu_doclist_add_recursive(cobs_gpu_doclist* dl, const char* root, uint32_t filter) {
    if (!dl || !root || filter > COBS_GPU_FILETYPE_LIST) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status { return add_recursive(root, (FileType)filter, dl->list); });
}

cobs_gpu_status cobs_gpu_doclist_add_memory(cobs_gpu_doclist* dl, const char* name, const char* text, size_t len) {
    if (!dl || !name || (!text && len)) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        DocEntry e;
        e.path = name;
        e.name = name;
        e.type = FileType::Memory;
        e.text.assign(text ? text : "", len);
        e.size = len + 1;
        dl->list.push_back(std::move(e));
        return COBS_GPU_OK;
    });
}

size_t cobs_gpu_doclist_size(const cobs_gpu_doclist* dl) { return dl ? dl->list.size() : 0; }

cobs_gpu_status cobs_gpu_doclist_entry(const cobs_gpu_doclist* dl, size_t i, cobs_gpu_doc_entry* out) {
    if (!dl || !out || i >= dl->list.size()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    const DocEntry& e = dl->list[i];
    out->path = e.path.c_str();
    out->name = e.name.c_str();
    out->type = (uint32_t)e.type;
    out->reserved = 0;
    out->size = e.size;
    out->subdoc_index = e.subdoc_index;
    out->term_size = e.term_size;
    out->term_count = e.term_count;
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_doclist_sort(cobs_gpu_doclist* dl, uint32_t by) {
    if (!dl || by > COBS_GPU_SORT_BY_SIZE) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        if (by == COBS_GPU_SORT_BY_PATH)        // sort_by_path, document_list.hpp:416-421: by path only
            std::stable_sort(dl->list.begin(), dl->list.end(),
                             [](const DocEntry& a, const DocEntry& b) { return a.path < b.path; });
        else
            sort_entries(dl->list, by);
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_doclist_num_terms(const cobs_gpu_doclist* dl, size_t i, uint32_t term_size, uint64_t* out) {
    if (!dl || !out || i >= dl->list.size() || term_size == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    *out = num_terms(dl->list[i], term_size);
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_doclist_terms(const cobs_gpu_doclist* dl, size_t i, uint32_t term_size, char* out,
                                       size_t cap_bytes, uint64_t* n_terms) {
    if (!dl || !n_terms || i >= dl->list.size() || term_size == 0 || (!out && cap_bytes))
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        const DocEntry& e = dl->list[i];
        std::string text((size_t)term_text_bound(e, term_size), '\0'), scratch;
        std::vector<TermSeg> segs;
        TermSink sink;
        sink.data = &text[0];
        sink.cap = text.size();
        cobs_gpu_status st = load_terms(e, term_size, sink, segs, scratch);
        if (st != COBS_GPU_OK) return st;
        uint64_t n = 0;
        const size_t k = term_size;
        for (const TermSeg& g : segs) {
            const char* p = text.data() + g.begin;
            size_t run = 0;                      // characters since the last separator
            for (size_t j = 0; j < g.len; ++j) {
                run = (!g.raw && p[j] == '\n') ? 0 : run + 1;
                if (run >= k) {
                    if ((n + 1) * k <= cap_bytes) std::memcpy(out + n * k, p + j + 1 - k, k);
                    ++n;
                }
            }
        }
        *n_terms = n;
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_filetype_from_string(const char* s, uint32_t* out) {
    if (!s || !out) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument");
    FileType ft;
    if (!parse_filetype(s, ft)) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, (std::string("Unknown file type ") + s).c_str());
    *out = (uint32_t)ft;
    return COBS_GPU_OK;
}

// classic_combine (classic_index.cpp:195-327) on the GPU: the rows of several classic indexes with
// the same parameters are concatenated at bit granularity into one index (document names in input
// order).  Row batches of at most mem_bytes (0 = 1 GiB) travel file -> HBM -> file.
// (The reference's general-bit path ORs into an output block it never clears between batches,
// :302-312; this writes what its single-batch case writes.)
cobs_gpu_status cobs_gpu_combine_classic(const char* const* in_paths, size_t n, const char* out_path,
                                         uint64_t mem_bytes, int device) {
    if (!in_paths || n == 0 || !out_path) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no inputs");
    return guarded([&]() -> cobs_gpu_status {
        std::vector<std::unique_ptr<MappedFile>> files;
        std::vector<IndexMeta> metas(n);
        std::string err;
        for (size_t i = 0; i < n; ++i) {
            files.emplace_back(new MappedFile);
            if (!in_paths[i] || !files[i]->open(in_paths[i], err)) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, err.c_str());
            if (!parse_index_header(files[i]->data(), files[i]->size(), metas[i], err) || metas[i].kind != IndexKind::Classic)
                return cobs_gpu_set_error(COBS_GPU_ERR_FORMAT, (std::string(in_paths[i]) + ": not a classic index").c_str());
            const IndexMeta &a = metas[0], &b = metas[i];
            if (a.term_size != b.term_size || a.canonicalize != b.canonicalize || a.num_hashes != b.num_hashes ||
                a.signature_sizes[0] != b.signature_sizes[0])
                return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "indexes to combine differ in term size, canonicalize, hashes or signature size");
        }
        cobs_gpu_status st = pick_device(device);
        if (st != COBS_GPU_OK) return st;
        const uint64_t sig = metas[0].signature_sizes[0];
        std::vector<uint64_t> bit_off(n + 1, 0), src_rb(n);
        uint64_t in_row_bytes = 0;
        for (size_t i = 0; i < n; ++i) {
            bit_off[i + 1] = bit_off[i] + metas[i].doc_names.size();
            src_rb[i] = metas[i].page_row_bytes();
            in_row_bytes += src_rb[i];
        }
        const uint64_t total_docs = bit_off[n];
        if (total_docs == 0 || total_docs > 0xFFFFFFF0ull) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad document count");
        const uint64_t out_rb = (total_docs + 7) / 8;
        std::string h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, metas[0].term_size);
        put<uint8_t>(h, metas[0].canonicalize);
        put<uint32_t>(h, (uint32_t)total_docs);
        put<uint64_t>(h, sig);
        put<uint64_t>(h, metas[0].num_hashes);
        for (size_t i = 0; i < n; ++i)
            for (const std::string& nm : metas[i].doc_names) { h += nm; h += '\n'; }
        h += "CLASSIC_INDEX";
        FILE* f = std::fopen(out_path, "wb");
        if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
        struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
        if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        // batch_size rows at a time (the reference: mem_bytes / new_row_bytes / 2, :236-238)
        const uint64_t budget = mem_bytes ? mem_bytes : 1ull << 30;
        const uint64_t batch = std::max<uint64_t>(1, std::min(sig, budget / (in_row_bytes + out_rb)));
        DevMem d_in, d_out, d_ptr, d_rb, d_off;
        BUILD_TRY(hipMalloc(&d_in.p, (size_t)(batch * in_row_bytes)));
        BUILD_TRY(hipMalloc(&d_out.p, (size_t)(batch * out_rb)));
        BUILD_TRY(hipMalloc(&d_ptr.p, n * sizeof(void*)));
        BUILD_TRY(hipMalloc(&d_rb.p, n * 8));
        BUILD_TRY(hipMalloc(&d_off.p, (n + 1) * 8));
        std::vector<const uint8_t*> ptrs(n);
        uint64_t pos = 0;
        for (size_t i = 0; i < n; ++i) { ptrs[i] = (const uint8_t*)d_in.p + pos; pos += batch * src_rb[i]; }
        BUILD_TRY(hipMemcpy(d_ptr.p, ptrs.data(), n * sizeof(void*), hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_rb.p, src_rb.data(), n * 8, hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_off.p, bit_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
        std::vector<uint8_t> host_out((size_t)(batch * out_rb));
        for (uint64_t r0 = 0; r0 < sig; r0 += batch) {
            const uint64_t rows = std::min(batch, sig - r0);
            for (size_t i = 0; i < n; ++i)
                if (src_rb[i])
                    BUILD_TRY(hipMemcpy(const_cast<uint8_t*>(ptrs[i]), files[i]->data() + metas[i].data_offset + r0 * src_rb[i],
                                        (size_t)(rows * src_rb[i]), hipMemcpyHostToDevice));
            CombineArgs a;
            a.src = (const uint8_t* const*)d_ptr.p;
            a.src_row_bytes = (const uint64_t*)d_rb.p;
            a.bit_off = (const uint64_t*)d_off.p;
            a.dst = (uint8_t*)d_out.p;
            a.dst_row_bytes = out_rb;
            a.rows = rows;
            a.nsrc = (uint32_t)n;
            BUILD_TRY(launch_combine(a, nullptr));
            BUILD_TRY(hipMemcpy(host_out.data(), d_out.p, (size_t)(rows * out_rb), hipMemcpyDeviceToHost));
            if (!write_all(f, host_out.data(), (size_t)(rows * out_rb))) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        }
        closer.f = nullptr;
        if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        return COBS_GPU_OK;
    });
}

// classic_construct_random (classic_index.cpp:661-725, `cobs classic-construct-random`,
// src/cobs.cpp:243-291): num_documents documents of document_size random 31-mers each,
// canonicalised, hashed num_hashes times into signature_size rows; names file_%06u; k = 31,
// canonicalize = 1.  The random stream is this library's own counter generator (see
// random_build_kernel), not std::mt19937: same distribution, different bits than the reference
// produces for the same seed.
cobs_gpu_status cobs_gpu_construct_random(const char* out_path, uint64_t signature_size, uint64_t num_documents,
                                          uint64_t document_size, uint64_t num_hashes, uint64_t seed, int device) {
    if (!out_path || signature_size == 0 || signature_size > (1ull << 46) || num_documents == 0 ||
        num_documents > 0xFFFFFFF0ull || num_hashes == 0 || num_hashes > 64)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        cobs_gpu_status st = pick_device(device);
        if (st != COBS_GPU_OK) return st;
        const uint64_t row_size = (num_documents + 7) / 8, row_bytes = (row_size + 3) / 4 * 4;
        DevMem d_mat;
        BUILD_TRY(hipMalloc(&d_mat.p, (size_t)(signature_size * row_bytes)));
        BUILD_TRY(hipMemset(d_mat.p, 0, (size_t)(signature_size * row_bytes)));
        // launches of at most 2^31 k-mers
        const uint64_t per = document_size ? std::max<uint64_t>(1, (1ull << 31) / document_size) : num_documents;
        for (uint64_t d0 = 0; d0 < num_documents && document_size; d0 += per) {
            RandomBuildArgs a;
            a.matrix = (uint32_t*)d_mat.p;
            a.signature_size = signature_size;
            a.magic = ~0ull / signature_size;
            a.row_bytes = row_bytes;
            a.doc0 = d0;
            a.document_size = document_size;
            a.seed = seed;
            a.num_hashes = (uint32_t)num_hashes;
            BUILD_TRY(launch_random_build(a, std::min(per, num_documents - d0), nullptr));
        }
        BUILD_TRY(hipStreamSynchronize(nullptr));
        std::string h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, 31);
        put<uint8_t>(h, 1);
        put<uint32_t>(h, (uint32_t)num_documents);
        put<uint64_t>(h, signature_size);
        put<uint64_t>(h, num_hashes);
        char nm[32];
        for (uint64_t i = 0; i < num_documents; ++i) { std::snprintf(nm, sizeof nm, "file_%06u\n", (unsigned)i); h += nm; }
        h += "CLASSIC_INDEX";
        FILE* f = std::fopen(out_path, "wb");
        if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
        struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
        if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        st = stream_rows_to_file(f, (const uint8_t*)d_mat.p, row_bytes, row_size, signature_size);
        if (st != COBS_GPU_OK) return st;
        closer.f = nullptr;
        if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        return COBS_GPU_OK;
    });
}

// The procedural index of cobs_gpu_open_synthetic as a FILE in the reference's format: the
// stand-in for `cobs classic-construct-random` (src/cobs.cpp:243-291,
// construction/classic_index.cpp:661-725) at sizes where hashing 10^10 random k-mers is not the
// point -- same header (classic_index_header.cpp:26-37 / compact_index_header.cpp:20-43), document
// names file_%06u (classic_index.cpp:668-670), bits of density ~0.3.  `cobs query`, the reference's
// tools and this engine (resident or streamed) read it.  Rows are generated on the device in
// chunks and streamed to the file: neither HBM nor host memory holds the matrix.
cobs_gpu_status cobs_gpu_write_synthetic(const cobs_gpu_synth* d, const char* out_path, int device) {
    if (!d || !d->signature_sizes || !out_path || d->num_pages == 0 || d->kind > 1)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad synthetic index description");
    if ((d->kind == 1 && d->page_size == 0) || (d->kind == 0 && d->num_pages != 1) || d->num_docs == 0 ||
        d->num_docs > 0xFFFFFFF0ull || (d->kind == 1 && d->num_docs > (uint64_t)d->num_pages * 8 * d->page_size))
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad synthetic index geometry");
    cobs_gpu_status st = pick_device(device);
    if (st != COBS_GPU_OK) return st;
    const uint64_t prb = d->kind ? d->page_size : (d->num_docs + 7) / 8;
    std::string h;
    char nm[32];
    if (d->kind == 0) {
        h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, d->term_size);
        put<uint8_t>(h, (uint8_t)d->canonicalize);
        put<uint32_t>(h, (uint32_t)d->num_docs);
        put<uint64_t>(h, d->signature_sizes[0]);
        put<uint64_t>(h, d->num_hashes);
        for (uint64_t i = 0; i < d->num_docs; ++i) { std::snprintf(nm, sizeof nm, "file_%06u\n", (unsigned)i); h += nm; }
        h += "CLASSIC_INDEX";
    } else {
        h = "COBS:COMPACT_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, d->term_size);
        put<uint8_t>(h, (uint8_t)d->canonicalize);
        put<uint32_t>(h, d->num_pages);
        put<uint32_t>(h, (uint32_t)d->num_docs);
        put<uint64_t>(h, d->page_size);
        for (uint32_t p = 0; p < d->num_pages; ++p) { put<uint64_t>(h, d->signature_sizes[p]); put<uint64_t>(h, d->num_hashes); }
        for (uint64_t i = 0; i < d->num_docs; ++i) { std::snprintf(nm, sizeof nm, "file_%06u\n", (unsigned)i); h += nm; }
        const uint64_t pad = (d->page_size - ((h.size() + 13) % d->page_size)) % d->page_size;
        h.append((size_t)pad, '\0');
        h += "COMPACT_INDEX";
    }
    FILE* f = std::fopen(out_path, "wb");
    if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
    struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
    if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    const uint32_t pitch = (uint32_t)((prb + 7) / 8 * 8);
    const uint64_t rows_per = std::max<uint64_t>(1, (256ull << 20) / pitch);
    DevMem d_rows;
    BUILD_TRY(hipMalloc(&d_rows.p, (size_t)(rows_per * pitch)));
    struct Pinned { void* p = nullptr; ~Pinned() { if (p) (void)hipHostFree(p); } } host[2];
    hipStream_t stream = nullptr;
    BUILD_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    for (auto& hb : host) BUILD_TRY(hipHostMalloc(&hb.p, (size_t)(rows_per * pitch), hipHostMallocDefault));
    int cur = 0;
    uint64_t pending = 0;       // bytes of host[cur ^ 1] still to be written
    for (uint32_t p = 0; p < d->num_pages; ++p) {
        const uint64_t sig = d->signature_sizes[p];
        const uint64_t first_doc = d->kind ? (uint64_t)p * 8 * d->page_size : 0;
        const uint64_t live = d->num_docs > first_doc ? d->num_docs - first_doc : 0;
        for (uint64_t r = 0; r < sig; r += rows_per) {
            const uint64_t n = std::min(rows_per, sig - r);
            SynthRowsArgs a;
            a.dst = (uint8_t*)d_rows.p;
            a.seed = d->seed;
            a.row0 = r;
            a.nrows = n;
            a.row_bytes = prb;
            a.live_docs = live;
            a.page = p;
            a.pitch = pitch;
            BUILD_TRY(launch_synth_rows(a, stream));
            // one contiguous copy incl. the pitch padding, dropped on the host (a 2-D copy of
            // millions of tiny rows takes minutes in the runtime: see stream_rows_to_file)
            BUILD_TRY(hipMemcpyAsync(host[cur].p, d_rows.p, (size_t)(n * pitch), hipMemcpyDeviceToHost, stream));
            // while the device produces this chunk the host writes the previous one
            if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
                return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
            BUILD_TRY(hipStreamSynchronize(stream));
            if (pitch != prb) {
                uint8_t* hp = (uint8_t*)host[cur].p;
                for (uint64_t q = 1; q < n; ++q) std::memmove(hp + q * prb, hp + q * (uint64_t)pitch, (size_t)prb);
            }
            pending = n * prb;
            cur ^= 1;
        }
    }
    if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
        return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    closer.f = nullptr;
    if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

// compact_combine_into_compact (compact_index.cpp:51-169; `cobs compact-construct-combine`):
// classic indexes become the sub-indexes of one compact index, rows padded to page_size bytes.
// File-to-file work without a kernel (nothing is hashed or counted); the inputs are left in place.
cobs_gpu_status cobs_gpu_combine_compact(const char* const* in_paths, size_t n, const char* out_path, uint64_t page_size) {
    if (!in_paths || n == 0 || !out_path || page_size == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument, no inputs or page_size 0");
    return guarded([&]() -> cobs_gpu_status {
        std::vector<std::unique_ptr<MappedFile>> files;
        std::vector<IndexMeta> metas(n);
        std::string err;
        size_t ndocs = 0;
        for (size_t i = 0; i < n; ++i) {
            files.emplace_back(new MappedFile);
            if (!in_paths[i] || !files[i]->open(in_paths[i], err)) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, err.c_str());
            if (!parse_index_header(files[i]->data(), files[i]->size(), metas[i], err) || metas[i].kind != IndexKind::Classic)
                return cobs_gpu_set_error(COBS_GPU_ERR_FORMAT, (std::string(in_paths[i]) + ": not a classic index").c_str());
            if (metas[i].term_size != metas[0].term_size || metas[i].canonicalize != metas[0].canonicalize)
                return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "indexes to combine differ in term size or canonicalize");      // :77-78
            // every index fills its page, the last one may be narrower (:85-90)
            const uint64_t rs = metas[i].page_row_bytes();
            if (i + 1 < n ? rs != page_size : rs > page_size)
                return cobs_gpu_set_error(COBS_GPU_ERR_ARG, (std::string(in_paths[i]) + ": row size does not match page_size").c_str());
            ndocs += metas[i].doc_names.size();
        }
        if (ndocs > 0xFFFFFFF0ull) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad document count");
        std::string h = "COBS:COMPACT_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, metas[0].term_size);
        put<uint8_t>(h, metas[0].canonicalize);
        put<uint32_t>(h, (uint32_t)n);
        put<uint32_t>(h, (uint32_t)ndocs);
        put<uint64_t>(h, page_size);
        for (const IndexMeta& m : metas) { put<uint64_t>(h, m.signature_sizes[0]); put<uint64_t>(h, m.num_hashes); }
        for (const IndexMeta& m : metas)
            for (const std::string& nm : m.doc_names) { h += nm; h += '\n'; }
        const uint64_t pad = (page_size - ((h.size() + 13) % page_size)) % page_size;
        h.append((size_t)pad, '\0');
        h += "COMPACT_INDEX";
        FILE* f = std::fopen(out_path, "wb");
        if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
        struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
        if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        std::vector<uint8_t> buf;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t rs = metas[i].page_row_bytes(), sig = metas[i].signature_sizes[0];
            const uint8_t* src = files[i]->data() + metas[i].data_offset;
            if (rs == page_size) {
                if (!write_all(f, src, (size_t)(sig * rs))) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
                continue;
            }
            const uint64_t rows_per = std::max<uint64_t>(1, (64ull << 20) / page_size);
            buf.assign((size_t)(std::min(rows_per, sig) * page_size), 0);
            for (uint64_t r0 = 0; r0 < sig; r0 += rows_per) {
                const uint64_t nr = std::min(rows_per, sig - r0);
                for (uint64_t r = 0; r < nr; ++r) std::memcpy(buf.data() + r * page_size, src + (r0 + r) * rs, (size_t)rs);
                if (!write_all(f, buf.data(), (size_t)(nr * page_size))) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
            }
        }
        closer.f = nullptr;
        if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        return COBS_GPU_OK;
    });
}

}  // extern "C"
