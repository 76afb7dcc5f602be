// cobs_amd/csrc/multi.cpp -- several GPUs of one node behind ONE handle of the C ABI
// (cobs_gpu_multi_*): the "device list" form of the multi-GPU path.  The index is sharded by
// sub-index block over the listed devices (cobs_gpu_options.shard_rank / shard_count; the cut is
// the reference's own document partition, cobs/query/compact_index/mmap_search_file.cpp:22-27),
// one worker thread per device opens its shard and joins an RCCL communicator
// (cobs_gpu_comm_create), and every search is one collective cobs_gpu_sharded_search_batch:
// each GPU scans its slice for the whole batch, counts / hit records / top-k candidates are
// exchanged over RCCL (comm.cpp), rank 0's global result goes to the caller -- except for the all-documents search
// (threshold 0, no limit), whose ranking the devices SHARE: count rows go all-to-all to query owners, every worker
// orders its queries and writes their results at their final places (cobs_gpu_sharded_search_batch_split).  Results
// are identical to cobs_gpu_search_batch on one GPU.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

using namespace cobs_amd;

struct cobs_gpu_multi {
    struct Rank {
        cobs_gpu_index* ix = nullptr;
        cobs_gpu_comm* comm = nullptr;
        cobs_gpu_status status = COBS_GPU_OK;
        std::string error;
    };
    std::vector<std::string> paths;
    std::vector<int> devices;
    cobs_gpu_options opts{};
    uint8_t id[COBS_GPU_UNIQUE_ID_BYTES];
    std::vector<Rank> ranks;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    uint64_t gen = 0;
    size_t done = 0;
    bool quit = false, failed = false;
    // the current job (read by the workers between the generation bump and their done signal)
    const char* const* queries = nullptr;
    const size_t* lens = nullptr;
    size_t nq = 0, num_results = 0, cap = 0;
    double threshold = 0.0;
    bool shared_ranking = false;     // this job is an all-documents search: the ranks split the ranking
    cobs_gpu_hit* hits = nullptr;
    size_t* hit_offsets = nullptr;
    size_t bad_query = 0;
    std::vector<std::vector<size_t>> offs_other;       // [rank] nq + 1 entries, ranks > 0

    void signal_done() {
        {
            std::lock_guard<std::mutex> g(mu);
            ++done;
        }
        cv_done.notify_all();
    }
    void wait_done() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return done == ranks.size(); });
        for (const Rank& k : ranks)
            if (!k.ix || !k.comm) failed = true;
    }
    void worker(size_t r) {
        Rank& me = ranks[r];
        const size_t n = ranks.size();
        // join the communicator first: ncclCommInitRank returns when all ranks have called it
        me.status = cobs_gpu_comm_create(id, (int)r, (int)n, devices[r], &me.comm);
        if (me.status == COBS_GPU_OK) {
            std::vector<const char*> cp;
            for (const auto& p : paths) cp.push_back(p.c_str());
            cobs_gpu_options o = opts;
            o.struct_size = sizeof o;
            o.device = devices[r];
            o.shard_rank = (uint32_t)r;
            o.shard_count = (uint32_t)n;
            me.status = cobs_gpu_open(cp.data(), cp.size(), &o, &me.ix);
        }
        if (me.status != COBS_GPU_OK) me.error = cobs_gpu_last_error();
        uint64_t seen = 0;
        signal_done();
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (quit) break;
            }
            if (failed) { signal_done(); continue; }        // a rank never came up: nothing collective may run
            size_t bad = 0;
            // ranks other than 0 take part in the collectives but keep no result (capacity 0); their
            // offset arrays were sized by the calling thread: nothing is allocated between the
            // wake-up and the collective, so no rank can drop out of it on its own
            if (shared_ranking) {
                // the all-documents search: the devices share the ranking -- every worker orders the queries its rank
                // owns and writes their results and offsets at their final places of the caller's arrays
                me.status = cobs_gpu_sharded_search_batch_split(me.ix, me.comm, queries, lens, nq, threshold, num_results,
                                                                hits, cap, hit_offsets, &bad);
                if (r != 0 && me.status == COBS_GPU_ERR_CAPACITY) me.status = COBS_GPU_OK;      // rank 0 reports it
                if (me.status != COBS_GPU_OK) me.error = cobs_gpu_last_error();
                if (r == 0) bad_query = bad;
                signal_done();
                continue;
            }
            size_t* offs = r == 0 ? hit_offsets : offs_other[r].data();
            me.status = cobs_gpu_sharded_search_batch(me.ix, me.comm, queries, lens, nq, threshold, num_results,
                                                      r == 0 ? hits : nullptr, r == 0 ? cap : 0, offs, &bad);
            if (r != 0 && me.status == COBS_GPU_ERR_CAPACITY) me.status = COBS_GPU_OK;
            if (me.status != COBS_GPU_OK) me.error = cobs_gpu_last_error();
            if (r == 0) bad_query = bad;
            signal_done();
        }
        if (me.ix) cobs_gpu_close(me.ix);
        if (me.comm) cobs_gpu_comm_destroy(me.comm);
    }
    void stop() {
        {
            std::lock_guard<std::mutex> g(mu);
            quit = true;
            ++gen;
        }
        cv.notify_all();
        for (auto& t : threads) t.join();
        threads.clear();
    }
    // the first failing rank's status and message -> this thread's last error
    cobs_gpu_status verdict() {
        for (size_t r = 0; r < ranks.size(); ++r)
            if (ranks[r].status != COBS_GPU_OK && ranks[r].status != COBS_GPU_ERR_CAPACITY)
                return fail(ranks[r].status, ranks[r].error + " (device " + std::to_string(devices[r]) + ")");
        if (ranks[0].status == COBS_GPU_ERR_CAPACITY) return fail(ranks[0].status, ranks[0].error);
        return COBS_GPU_OK;
    }
};

extern "C" {

cobs_gpu_status cobs_gpu_multi_open(const char* const* paths, size_t n_paths, const int* devices, size_t n_devices,
                                    const cobs_gpu_options* opts, cobs_gpu_multi** out) {
    if (!out) return fail(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!paths || n_paths == 0 || !devices || n_devices == 0) return fail(COBS_GPU_ERR_ARG, "no index paths or no devices");
    return guarded([&]() -> cobs_gpu_status {
        // everything that can fail on ONE rank only is checked before the ranks meet inside
        // ncclCommInitRank (a rank that never arrives would leave the others waiting)
        const int nd = cobs_gpu_device_count();
        if (nd <= 0) return fail(COBS_GPU_ERR_NO_DEVICE, "no HIP device visible; libcobs_gpu has no CPU fallback");
        for (size_t i = 0; i < n_devices; ++i) {
            if (devices[i] < 0 || devices[i] >= nd)
                return fail(COBS_GPU_ERR_ARG, "device ordinal " + std::to_string(devices[i]) + " out of range");
            // (RCCL refuses two ranks on one device; tests/mock_rccl runs N ranks of this handle on ONE GPU against a
            // stand-in for librccl, which says so by DEFINING this symbol -- nothing in a real environment does)
            static const bool ranks_may_share = dlsym(RTLD_DEFAULT, "mock_rccl_ranks_may_share_a_device") != nullptr;
            for (size_t j = 0; j < i && !ranks_may_share; ++j)
                if (devices[j] == devices[i]) return fail(COBS_GPU_ERR_ARG, "a device is listed twice");
        }
        std::unique_ptr<cobs_gpu_multi> m(new cobs_gpu_multi);
        for (size_t i = 0; i < n_paths; ++i) {
            if (!paths[i]) return fail(COBS_GPU_ERR_ARG, "NULL path");
            m->paths.emplace_back(paths[i]);
        }
        m->devices.assign(devices, devices + n_devices);
        if (opts) std::memcpy(&m->opts, opts, std::min<size_t>(opts->struct_size, sizeof m->opts));
        cobs_gpu_status st = cobs_gpu_comm_unique_id(m->id);
        if (st != COBS_GPU_OK) return st;
        m->ranks.resize(n_devices);
        cobs_gpu_multi* raw = m.get();
        for (size_t r = 0; r < n_devices; ++r) m->threads.emplace_back([raw, r]() { raw->worker(r); });
        m->wait_done();          // every rank has opened its shard and joined the communicator (or failed)
        st = m->verdict();
        if (st != COBS_GPU_OK) {
            const std::string keep = cobs_gpu_last_error();
            m->stop();
            return fail(st, keep);
        }
        *out = m.release();
        return COBS_GPU_OK;
    });
}

void cobs_gpu_multi_close(cobs_gpu_multi* m) {
    if (!m) return;
    m->stop();
    delete m;
}

size_t cobs_gpu_multi_size(const cobs_gpu_multi* m) {
    return m && !m->ranks.empty() ? (size_t)cobs_gpu_comm_size(m->ranks[0].comm) : 0;
}

cobs_gpu_index* cobs_gpu_multi_index(const cobs_gpu_multi* m, size_t rank) {
    return m && rank < m->ranks.size() ? m->ranks[rank].ix : nullptr;
}

cobs_gpu_status cobs_gpu_multi_search_batch(cobs_gpu_multi* m, const char* const* queries, const size_t* lens, size_t nq,
                                            double threshold, size_t num_results, cobs_gpu_hit* hits, size_t cap,
                                            size_t* hit_offsets, size_t* bad_query) {
    if (!m || !hit_offsets) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (nq && (!queries || !lens)) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        m->offs_other.resize(m->ranks.size());
        for (size_t r = 1; r < m->ranks.size(); ++r) m->offs_other[r].assign(nq + 1, 0);
        m->queries = queries;
        m->lens = lens;
        m->nq = nq;
        m->threshold = threshold;
        m->num_results = num_results;
        {
            const cobs_gpu_index* ix0 = m->ranks[0].ix;
            m->shared_ranking = threshold <= 0.0 && (num_results == 0 || num_results >= cobs_gpu_total_counts(ix0));
        }
        m->hits = hits;
        m->cap = cap;
        m->hit_offsets = hit_offsets;
        m->bad_query = 0;
        {
            std::lock_guard<std::mutex> g(m->mu);
            m->done = 0;
            ++m->gen;
        }
        m->cv.notify_all();
        m->wait_done();
        const cobs_gpu_status st = m->verdict();
        if (st != COBS_GPU_OK && bad_query) *bad_query = m->bad_query;
        return st;
    });
}

}  // extern "C"
