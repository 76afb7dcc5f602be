// include/cobs_gpu_search.hpp -- C++17 host-side mirror of the reference's operator
// API for the query path, implemented over the C ABI of cobs_gpu.h (header only).
//
// Reference interface being mirrored:
//   struct cobs::SearchResult { const char* doc_name; uint32_t score; }   (cobs/query/search.hpp:17-27)
//   class  cobs::Search { virtual void search(const std::string& query,
//                std::vector<SearchResult>& result, double threshold = 0.0,
//                size_t num_results = 0) = 0; }                            (cobs/query/search.hpp:29-47)
//   class  cobs::ClassicSearch : Search { ClassicSearch(std::string path); ... } (classic_search.hpp:19-37)
//
// Same names, argument meaning and defaults.  Differences, all at the error
// boundary: where the reference terminates the process (exit/abort on a short
// query, a non-ACGT base, an unreadable index: classic_search.cpp:431-433, :93-96,
// :61-63) this class throws cobs_gpu::Error carrying the C-ABI status.
// `result` is caller-owned, resized and overwritten (classic_search.cpp:147,190);
// SearchResult::doc_name points into strings owned by the Search object.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cobs_gpu.h"

namespace cobs_gpu {

struct SearchResult {
    //! string reference to document name (owned by the index handle)
    const char* doc_name = nullptr;
    //! score (number of matched k-mers)
    uint32_t score = 0;
    SearchResult() = default;
    SearchResult(const char* name, uint32_t s) : doc_name(name), score(s) {}
};

class Error : public std::runtime_error {
public:
    Error(cobs_gpu_status st, const std::string& msg) : std::runtime_error(msg), status(st) {}
    cobs_gpu_status status;
};

class Search {
public:
    virtual ~Search() = default;
    virtual void search(const std::string& query, std::vector<SearchResult>& result,
                        double threshold = 0.0, size_t num_results = 0) = 0;
};

//! Search plus the batch call (the performance path; the reference loops queries serially)
class BatchSearch : public Search {
public:
    virtual void search_batch(const std::vector<std::string>& queries,
                              std::vector<std::vector<SearchResult>>& results,
                              double threshold = 0.0, size_t num_results = 0) = 0;
    //! the index handle timers / geometry are read from (rank 0 of a sharded search)
    virtual cobs_gpu_index* handle() const = 0;
};

class ClassicSearch : public BatchSearch {
public:
    //! auto-detect classic / compact and stage the index into HBM
    //! hbm_budget_bytes > 0: an index larger than the budget is streamed chunk-wise at every
    //! search (the role of the reference's mmap / AIO back-ends for indexes beyond memory)
    explicit ClassicSearch(const std::string& path, int device = -1, uint64_t hbm_budget_bytes = 0)
        : ClassicSearch(std::vector<std::string>{path}, device, hbm_budget_bytes) {}

    //! several index files searched together (reference: vector<shared_ptr<IndexSearchFile>>)
    explicit ClassicSearch(const std::vector<std::string>& paths, int device = -1,
                           uint64_t hbm_budget_bytes = 0) {
        std::vector<const char*> cp;
        for (const auto& p : paths) cp.push_back(p.c_str());
        cobs_gpu_options o{};
        o.struct_size = sizeof o;
        o.device = device;
        o.hbm_budget_bytes = hbm_budget_bytes;
        check(cobs_gpu_open(cp.data(), cp.size(), &o, &ix_));
    }

    ~ClassicSearch() override { cobs_gpu_close(ix_); }
    ClassicSearch(const ClassicSearch&) = delete;
    ClassicSearch& operator=(const ClassicSearch&) = delete;

    void search(const std::string& query, std::vector<SearchResult>& result,
                double threshold = 0.0, size_t num_results = 0) final {
        size_t n = 0;
        const size_t total = (size_t)cobs_gpu_total_counts(ix_);
        hits_.resize(num_results == 0 || num_results > total ? total : num_results);
        check(cobs_gpu_search(ix_, query.data(), query.size(), threshold, num_results,
                              hits_.data(), hits_.size(), &n));
        result.resize(n);
        for (size_t i = 0; i < n; ++i)
            result[i] = SearchResult(cobs_gpu_doc_name(ix_, hits_[i].file_no, hits_[i].doc), hits_[i].score);
    }

    //! many queries in one device pass (the performance path; the reference loops serially)
    void search_batch(const std::vector<std::string>& queries,
                      std::vector<std::vector<SearchResult>>& results,
                      double threshold = 0.0, size_t num_results = 0) override {
        std::vector<const char*> qp;
        std::vector<size_t> ql;
        for (const auto& q : queries) { qp.push_back(q.data()); ql.push_back(q.size()); }
        std::vector<size_t> offs(queries.size() + 1, 0);
        const size_t total = (size_t)cobs_gpu_total_counts(ix_);
        // hit buffer: exact when the result count is known (a limit, or threshold 0 = every
        // document); with a threshold start small and grow to the size the library reports
        size_t cap;
        if (num_results > 0) cap = (num_results > total ? total : num_results) * queries.size();
        else if (threshold <= 0.0) cap = total * queries.size();
        else cap = 16 * queries.size() + 1024;
        size_t bad = 0;
        cobs_gpu_status st;
        for (;;) {
            hits_.resize(cap + 1);
            st = cobs_gpu_search_batch(ix_, qp.data(), ql.data(), queries.size(), threshold,
                                       num_results, hits_.data(), hits_.size(), offs.data(), &bad);
            if (st == COBS_GPU_ERR_CAPACITY && offs[queries.size()] > cap) {
                cap = offs[queries.size()];        // needed size, as documented in cobs_gpu.h
                continue;
            }
            break;
        }
        check(st);
        results.resize(queries.size());
        for (size_t q = 0; q < queries.size(); ++q) {
            results[q].resize(offs[q + 1] - offs[q]);
            for (size_t i = offs[q]; i < offs[q + 1]; ++i)
                results[q][i - offs[q]] =
                    SearchResult(cobs_gpu_doc_name(ix_, hits_[i].file_no, hits_[i].doc), hits_[i].score);
        }
    }

    cobs_gpu_index* handle() const override { return ix_; }

private:
    static void check(cobs_gpu_status st) {
        if (st != COBS_GPU_OK) throw Error(st, cobs_gpu_last_error());
    }
    cobs_gpu_index* ix_ = nullptr;
    std::vector<cobs_gpu_hit> hits_;
};

//! The same operator over SEVERAL GPUs of one node, one process: the index is sharded by
//! sub-index block over the devices (cobs_gpu_options.shard_rank / shard_count; the cut is the
//! reference's own -- sub-indexes cover disjoint document ranges,
//! compact_index/mmap_search_file.cpp:22-27), one worker thread per device opens its shard and
//! joins an RCCL communicator (cobs_gpu_comm_create), and every search is one collective
//! cobs_gpu_sharded_search_batch: each GPU scans its slice for the whole batch, the counts / hit
//! records / top-k candidates are exchanged over RCCL (xGMI) inside libcobs_gpu.so, rank 0's
//! (global) result is returned.  Results are identical to ClassicSearch on one GPU.
class ShardedClassicSearch : public BatchSearch {
public:
    ShardedClassicSearch(const std::vector<std::string>& paths, const std::vector<int>& devices,
                         uint64_t hbm_budget_bytes = 0)
        : paths_(paths), devices_(devices), budget_(hbm_budget_bytes), n_(devices.size()), ranks_(devices.size()) {
        if (devices.empty()) throw Error(COBS_GPU_ERR_ARG, "no devices");
        // everything that can fail on ONE rank only is checked before the ranks meet inside
        // ncclCommInitRank (a rank that never arrives would leave the others waiting)
        const int nd = cobs_gpu_device_count();
        if (nd <= 0) throw Error(COBS_GPU_ERR_NO_DEVICE, "no HIP device visible; libcobs_gpu has no CPU fallback");
        for (size_t i = 0; i < devices.size(); ++i) {
            if (devices[i] < 0 || devices[i] >= nd)
                throw Error(COBS_GPU_ERR_ARG, "device ordinal " + std::to_string(devices[i]) + " out of range");
            for (size_t j = 0; j < i; ++j)
                if (devices[j] == devices[i]) throw Error(COBS_GPU_ERR_ARG, "a device is listed twice");
        }
        check0(cobs_gpu_comm_unique_id(id_));
        for (size_t r = 0; r < n_; ++r) threads_.emplace_back([this, r]() { worker(r); });
        wait_done();          // every rank has opened its shard and joined the communicator (or failed)
        rethrow();
    }

    ~ShardedClassicSearch() override {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    ShardedClassicSearch(const ShardedClassicSearch&) = delete;
    ShardedClassicSearch& operator=(const ShardedClassicSearch&) = delete;

    void search(const std::string& query, std::vector<SearchResult>& result,
                double threshold = 0.0, size_t num_results = 0) final {
        std::vector<std::vector<SearchResult>> rs;
        search_batch({query}, rs, threshold, num_results);
        result = std::move(rs[0]);
    }

    void search_batch(const std::vector<std::string>& queries, std::vector<std::vector<SearchResult>>& results,
                      double threshold = 0.0, size_t num_results = 0) override {
        qp_.clear();
        ql_.clear();
        for (const auto& q : queries) { qp_.push_back(q.data()); ql_.push_back(q.size()); }
        offs_.assign(queries.size() + 1, 0);
        const size_t total = (size_t)cobs_gpu_total_counts(ranks_[0].ix);
        if (num_results > 0) cap_ = (num_results > total ? total : num_results) * queries.size();
        else if (threshold <= 0.0) cap_ = total * queries.size();
        else cap_ = 16 * queries.size() + 1024;
        threshold_ = threshold;
        num_results_ = num_results;
        for (;;) {
            hits_.resize(cap_ + 1);
            {
                std::lock_guard<std::mutex> g(mu_);
                done_ = 0;
                ++gen_;
            }
            cv_.notify_all();
            wait_done();
            // the call is collective: if rank 0's buffer was too small every rank repeats it
            if (ranks_[0].status == COBS_GPU_ERR_CAPACITY && offs_[queries.size()] > cap_) {
                cap_ = offs_[queries.size()];
                continue;
            }
            break;
        }
        rethrow();
        results.resize(queries.size());
        cobs_gpu_index* ix = ranks_[0].ix;
        for (size_t q = 0; q < queries.size(); ++q) {
            results[q].resize(offs_[q + 1] - offs_[q]);
            for (size_t i = offs_[q]; i < offs_[q + 1]; ++i)
                results[q][i - offs_[q]] = SearchResult(cobs_gpu_doc_name(ix, hits_[i].file_no, hits_[i].doc), hits_[i].score);
        }
    }

    cobs_gpu_index* handle() const override { return ranks_[0].ix; }
    //! ncclCommCount of the communicator the ranks joined
    int comm_size() const { return cobs_gpu_comm_size(ranks_[0].comm); }

private:
    struct Rank {
        cobs_gpu_index* ix = nullptr;
        cobs_gpu_comm* comm = nullptr;
        cobs_gpu_status status = COBS_GPU_OK;
        std::string error;
    };

    void worker(size_t r) {
        Rank& me = ranks_[r];
        // join the communicator first: ncclCommInitRank returns when all ranks have called it
        me.status = cobs_gpu_comm_create(id_, (int)r, (int)n_, devices_[r], &me.comm);
        if (me.status == COBS_GPU_OK) {
            std::vector<const char*> cp;
            for (const auto& p : paths_) cp.push_back(p.c_str());
            cobs_gpu_options o{};
            o.struct_size = sizeof o;
            o.device = devices_[r];
            o.shard_rank = (uint32_t)r;
            o.shard_count = (uint32_t)n_;
            o.hbm_budget_bytes = budget_;
            me.status = cobs_gpu_open(cp.data(), cp.size(), &o, &me.ix);
        }
        if (me.status != COBS_GPU_OK) me.error = cobs_gpu_last_error();
        uint64_t seen = 0;
        signal_done();
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (quit_) break;
            }
            if (failed_) { signal_done(); continue; }       // a rank never came up: nothing collective may run
            size_t bad = 0;
            // ranks other than 0 take part in the collectives but keep no result (capacity 0)
            std::vector<size_t> offs_local;
            size_t* offs = r == 0 ? offs_.data() : (offs_local.assign(qp_.size() + 1, 0), offs_local.data());
            me.status = cobs_gpu_sharded_search_batch(me.ix, me.comm, qp_.data(), ql_.data(), qp_.size(), threshold_,
                                                      num_results_, r == 0 ? hits_.data() : nullptr,
                                                      r == 0 ? hits_.size() : 0, offs, &bad);
            if (r != 0 && me.status == COBS_GPU_ERR_CAPACITY) me.status = COBS_GPU_OK;
            if (me.status != COBS_GPU_OK) me.error = cobs_gpu_last_error();
            signal_done();
        }
        if (me.ix) cobs_gpu_close(me.ix);
        if (me.comm) cobs_gpu_comm_destroy(me.comm);
    }

    void signal_done() {
        {
            std::lock_guard<std::mutex> g(mu_);
            ++done_;
        }
        cv_done_.notify_all();
    }
    void wait_done() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return done_ == n_; });
        for (const Rank& k : ranks_)
            if (!k.ix || !k.comm) failed_ = true;
    }
    void rethrow() {
        for (size_t r = 0; r < n_; ++r)
            if (ranks_[r].status != COBS_GPU_OK && ranks_[r].status != COBS_GPU_ERR_CAPACITY)
                throw Error(ranks_[r].status, ranks_[r].error + " (device " + std::to_string(devices_[r]) + ")");
        if (ranks_[0].status == COBS_GPU_ERR_CAPACITY) throw Error(ranks_[0].status, ranks_[0].error);
    }
    static void check0(cobs_gpu_status st) {
        if (st != COBS_GPU_OK) throw Error(st, cobs_gpu_last_error());
    }

    std::vector<std::string> paths_;
    std::vector<int> devices_;
    uint64_t budget_;
    size_t n_;
    uint8_t id_[COBS_GPU_UNIQUE_ID_BYTES];
    std::vector<Rank> ranks_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    uint64_t gen_ = 0;
    size_t done_ = 0;
    bool quit_ = false, failed_ = false;
    // the current job (read by the workers between the generation bump and their done signal)
    std::vector<const char*> qp_;
    std::vector<size_t> ql_;
    std::vector<size_t> offs_;
    std::vector<cobs_gpu_hit> hits_;
    size_t cap_ = 0, num_results_ = 0;
    double threshold_ = 0.0;
};

}  // namespace cobs_gpu
