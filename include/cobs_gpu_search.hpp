// include/cobs_gpu_search.hpp -- C++17 host-side mirror of the reference's operator
// API for the query path, implemented over the C ABI of cobs_gpu.h (header only).
//
// Reference interface being mirrored:
//   struct cobs::SearchResult { const char* doc_name; uint32_t score; }   (cobs/query/search.hpp:17-27)
//   class  cobs::Search { virtual void search(const std::string& query,
//                std::vector<SearchResult>& result, double threshold = 0.0,
//                size_t num_results = 0) = 0; }                            (cobs/query/search.hpp:29-47)
//   class  cobs::ClassicSearch : Search { ClassicSearch(std::string path); ... } (classic_search.hpp:19-37)
//   class  cobs::IndexSearchFile, ClassicIndexMMapSearchFile(path), CompactIndexMMapSearchFile(path) and
//          ClassicSearch(std::shared_ptr<IndexSearchFile>), ClassicSearch(std::vector<std::shared_ptr<...>>)
//          (cobs/query/index_file.hpp:19-49, */mmap_search_file.hpp, classic_search.cpp:41-49)
//
// Same names, argument meaning and defaults.  Differences, all at the error
// boundary: where the reference terminates the process (exit/abort on a short
// query, a non-ACGT base, an unreadable index: classic_search.cpp:431-433, :93-96,
// :61-63) this class throws cobs_gpu::Error carrying the C-ABI status.
// `result` is caller-owned, resized and overwritten (classic_search.cpp:147,190);
// SearchResult::doc_name points into strings owned by the Search object.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "cobs_gpu.h"
#include "cobs_gpu_batch.h"      // cobs_gpu_search_batch_view

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

//! The reference's Search carries a public `Timer timer_` with an accessor `timer()` (cobs/query/search.hpp:35-46,
//! cobs/util/timer.hpp:19-55) that its callers read and reset: `s.timer().print("search")` (src/cobs.cpp:468),
//! `s.timer().reset()` and `cobs::Timer t = s.timer(); t.get("hashes") ...` in benchmark_fpr_run (:623, :644-661).
//! Same shape here, over the handle's phase timers (cobs_gpu_timers): get(name), reset(), print(info[, os]); a COPY is a
//! snapshot (the reference copies the object to read it).  Names: the engine's own -- "hashes" (K1), "h2d" (query text),
//! "scan" (K2: row gather + AND + count, ONE kernel), "d2h", "rank" -- and the reference's, so that its callers compile
//! and print unchanged: "io" = the scan kernel (it is bound by the row gather the reference times as io), "and rows" and
//! "add rows" = 0 (inside that kernel), "sort results" = d2h + rank.  An unknown name reads 0, as a fresh entry of the
//! reference's timer does (timer.cpp:61-63).
class Timer {
public:
    Timer() = default;
    Timer(const Timer& o) { o.read(snap_); frozen_ = true; }
    Timer& operator=(const Timer& o) {
        if (this != &o) { o.read(snap_); frozen_ = true; ix_ = nullptr; }
        return *this;
    }
    //! (the Search object that owns this timer names its index handle once it is open)
    void bind(cobs_gpu_index* ix) { ix_ = ix; frozen_ = false; }
    void reset() {
        if (ix_ && !frozen_) (void)cobs_gpu_timers(ix_, nullptr, 1);
        for (double& v : snap_) v = 0;
    }
    double get(const char* name) const {
        double t[5];
        read(t);
        const std::string n(name ? name : "");
        if (n == "hashes") return t[0];
        if (n == "h2d") return t[1];
        if (n == "scan" || n == "io") return t[2];
        if (n == "d2h") return t[3];
        if (n == "rank") return t[4];
        if (n == "sort results") return t[3] + t[4];
        if (n == "total") return t[0] + t[1] + t[2] + t[3] + t[4];
        return 0.0;
    }
    //! "TIMER info=<info> name=seconds ... total=seconds" (timer.cpp:77-85)
    void print(const char* info, std::ostream& os) const {
        double t[5];
        read(t);
        os << "TIMER info=" << info << " hashes=" << t[0] << " h2d=" << t[1] << " scan=" << t[2] << " d2h=" << t[3]
           << " rank=" << t[4] << " total=" << t[0] + t[1] + t[2] + t[3] + t[4] << std::endl;
    }
    void print(const char* info) const { print(info, std::cerr); }

private:
    void read(double out[5]) const {
        for (int i = 0; i < 5; ++i) out[i] = snap_[i];
        if (ix_ && !frozen_) (void)cobs_gpu_timers(ix_, out, 0);
    }
    cobs_gpu_index* ix_ = nullptr;
    bool frozen_ = false;
    double snap_[5] = {0, 0, 0, 0, 0};
};

class Search {
public:
    virtual ~Search() = default;
    //! Returns timer_ (cobs/query/search.hpp:35-38)
    Timer& timer() { return timer_; }
    const Timer& timer() const { return timer_; }
    virtual void search(const std::string& query, std::vector<SearchResult>& result,
                        double threshold = 0.0, size_t num_results = 0) = 0;

public:
    //! timer of different query phases
    Timer timer_;
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

//! The reference hands ClassicSearch its index files as objects (cobs/query/index_file.hpp:19-49) made from a path
//! by the class of their kind, which refuses a file of the other kind when it reads the header
//! (classic_index/mmap_search_file.cpp:21-25, compact_index/mmap_search_file.cpp:20-27).  Here such an object is
//! the checked NAME of a file: the engine maps and stages it when a ClassicSearch is made from it.
class IndexSearchFile {
public:
    virtual ~IndexSearchFile() = default;
    const std::string& path() const { return path_; }

protected:
    IndexSearchFile(const std::string& path, const char* kind_word) : path_(path) {
        // every index file starts with "COBS:" and the word of its kind (cobs/file/header.cpp)
        const std::string want = std::string("COBS:") + kind_word;
        char head[32] = {0};
        std::FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw Error(COBS_GPU_ERR_OPEN, "cannot open index file " + path);
        const size_t got = std::fread(head, 1, want.size(), f);
        std::fclose(f);
        if (got != want.size() || std::memcmp(head, want.data(), want.size()) != 0)
            throw Error(COBS_GPU_ERR_FORMAT, path + " is not a " + kind_word + " file");
    }

private:
    std::string path_;
};

class ClassicIndexMMapSearchFile : public IndexSearchFile {
public:
    explicit ClassicIndexMMapSearchFile(const std::string& path) : IndexSearchFile(path, "CLASSIC_INDEX") {}
};

class CompactIndexMMapSearchFile : public IndexSearchFile {
public:
    explicit CompactIndexMMapSearchFile(const std::string& path) : IndexSearchFile(path, "COMPACT_INDEX") {}
};

class ClassicSearch : public BatchSearch {
public:
    //! one index file object / several, searched together (classic_search.cpp:41-49)
    explicit ClassicSearch(const std::shared_ptr<IndexSearchFile>& index, int device = -1, uint64_t hbm_budget_bytes = 0)
        : ClassicSearch(std::vector<std::string>{index->path()}, device, hbm_budget_bytes) {}
    explicit ClassicSearch(const std::vector<std::shared_ptr<IndexSearchFile>>& indices, int device = -1,
                           uint64_t hbm_budget_bytes = 0)
        : ClassicSearch(paths_of(indices), device, hbm_budget_bytes) {}

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
        timer_.bind(ix_);
    }

    //! adopt an index handle that is already open (e.g. built by cobs_gpu_build_index_list)
    explicit ClassicSearch(cobs_gpu_index* adopted) : ix_(adopted) { timer_.bind(ix_); }

    ~ClassicSearch() override { cobs_gpu_close(ix_); }
    ClassicSearch(const ClassicSearch&) = delete;
    ClassicSearch& operator=(const ClassicSearch&) = delete;
    ClassicSearch(ClassicSearch&& o) noexcept : ix_(o.ix_), hits_(std::move(o.hits_)) {
        o.ix_ = nullptr;
        o.timer_.bind(nullptr);
        timer_.bind(ix_);
    }

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
        else cap = 1;                                    // (thresholded without a limit: the view call below)
        size_t bad = 0;
        cobs_gpu_status st;
        if (threshold > 0.0 && num_results == 0) {
            // the number of hits is not known beforehand: collected in the arena the library grows as the passes come
            // home (cobs_gpu_batch.h) -- into a vector of a guessed size the whole search could have to run twice
            const cobs_gpu_hit* vh = nullptr;
            const size_t* vo = nullptr;
            check(cobs_gpu_search_batch_view(ix_, qp.data(), ql.data(), queries.size(), threshold, 0, &vh, &vo, &bad));
            results.resize(queries.size());
            for (size_t q = 0; q < queries.size(); ++q) {
                results[q].resize(vo[q + 1] - vo[q]);
                for (size_t i = vo[q]; i < vo[q + 1]; ++i)
                    results[q][i - vo[q]] = SearchResult(cobs_gpu_doc_name(ix_, vh[i].file_no, vh[i].doc), vh[i].score);
            }
            return;
        }
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
    static std::vector<std::string> paths_of(const std::vector<std::shared_ptr<IndexSearchFile>>& indices) {
        std::vector<std::string> out;
        for (const auto& i : indices) out.push_back(i->path());
        return out;
    }
    cobs_gpu_index* ix_ = nullptr;
    std::vector<cobs_gpu_hit> hits_;
};

//! The same operator over SEVERAL GPUs of one node, one process: the index is sharded by
//! sub-index block over the devices, every search is one scan per GPU plus one exchange over RCCL
//! inside libcobs_gpu.so (cobs_gpu_multi_*: worker thread per device, communicator, collective
//! cobs_gpu_sharded_search_batch).  Results are identical to ClassicSearch on one GPU.
class ShardedClassicSearch : public BatchSearch {
public:
    ShardedClassicSearch(const std::vector<std::string>& paths, const std::vector<int>& devices,
                         uint64_t hbm_budget_bytes = 0) {
        std::vector<const char*> cp;
        for (const auto& p : paths) cp.push_back(p.c_str());
        cobs_gpu_options o{};
        o.struct_size = sizeof o;
        o.hbm_budget_bytes = hbm_budget_bytes;
        check(cobs_gpu_multi_open(cp.data(), cp.size(), devices.data(), devices.size(), &o, &m_));
        timer_.bind(handle());
    }
    ~ShardedClassicSearch() override { cobs_gpu_multi_close(m_); }
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
        std::vector<const char*> qp;
        std::vector<size_t> ql;
        for (const auto& q : queries) { qp.push_back(q.data()); ql.push_back(q.size()); }
        std::vector<size_t> offs(queries.size() + 1, 0);
        cobs_gpu_index* ix = handle();
        const size_t total = (size_t)cobs_gpu_total_counts(ix);
        size_t cap;
        if (num_results > 0) cap = (num_results > total ? total : num_results) * queries.size();
        else if (threshold <= 0.0) cap = total * queries.size();
        else cap = std::max<size_t>(16 * queries.size(), (size_t)(hits_per_query_ * 1.25 * (double)queries.size())) + 1024;
        size_t bad = 0;
        cobs_gpu_status st;
        for (;;) {
            hits_.resize(cap + 1);
            st = cobs_gpu_multi_search_batch(m_, qp.data(), ql.data(), queries.size(), threshold, num_results,
                                             hits_.data(), hits_.size(), offs.data(), &bad);
            // the call is collective: if the buffer was too small every GPU repeats it
            if (st == COBS_GPU_ERR_CAPACITY && offs[queries.size()] > cap) {
                cap = offs[queries.size()];
                continue;
            }
            break;
        }
        check(st);
        if (threshold > 0.0 && num_results == 0 && !queries.empty())       // sizes the next thresholded calls: no second run
            hits_per_query_ = std::max((double)offs[queries.size()] / (double)queries.size(), 0.95 * hits_per_query_);
        results.resize(queries.size());
        for (size_t q = 0; q < queries.size(); ++q) {
            results[q].resize(offs[q + 1] - offs[q]);
            for (size_t i = offs[q]; i < offs[q + 1]; ++i)
                results[q][i - offs[q]] = SearchResult(cobs_gpu_doc_name(ix, hits_[i].file_no, hits_[i].doc), hits_[i].score);
        }
    }

    cobs_gpu_index* handle() const override { return cobs_gpu_multi_index(m_, 0); }
    //! every rank's shard handle (each holds the score slots of its own documents)
    std::vector<cobs_gpu_index*> shard_handles() const {
        std::vector<cobs_gpu_index*> v;
        for (size_t r = 0; r < cobs_gpu_multi_size(m_); ++r) v.push_back(cobs_gpu_multi_index(m_, r));
        return v;
    }
    //! ncclCommCount of the communicator the GPUs joined
    int comm_size() const { return (int)cobs_gpu_multi_size(m_); }

private:
    static void check(cobs_gpu_status st) {
        if (st != COBS_GPU_OK) throw Error(st, cobs_gpu_last_error());
    }
    cobs_gpu_multi* m_ = nullptr;
    std::vector<cobs_gpu_hit> hits_;
    double hits_per_query_ = 0.0;
};

}  // namespace cobs_gpu
