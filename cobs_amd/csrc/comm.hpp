// cobs_amd/csrc/comm.hpp -- internals shared by comm.cpp (communicators, the batch-level exchanges) and sharded.cpp
// (the pipelined sharded search and the device-resident sharded batch): the communicator handle, the per-batch exchange
// workspace, and the discipline around every RCCL call (counted, a failure marks the communicator, groups always
// closed, bounded stream waits).  Nothing here is part of the C ABI.
#pragma once
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

struct cobs_gpu_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    uint64_t serial = 0;          // never reused: a batch remembers which communicator its layout came from
    // What this rank entered last -- read by the CALLER's watchdog from another thread (cobs_gpu_comm_state) when a
    // step does not come back: a collective that a peer never enters does not fail, it waits.
    std::atomic<const char*> last_op{"none"};
    std::atomic<uint64_t> entered{0}, returned{0};
    std::atomic<void*> last_stream{nullptr};
    // A failed RCCL call leaves the peers' state unknown: the communicator is not used again (every later call fails
    // at once, on this rank, before any collective), and where the status says it is dead it is aborted -- after an
    // open group has been closed (GroupScope), never inside one.
    std::atomic<bool> broken{false};
    bool group_open = false, abort_wanted = false;
    std::string broken_why;
    uint32_t timeout_ms = 0;      // > 0: the stream waits this file owns give up after that long (sync_bounded)
};

namespace cobs_amd {

// per-batch exchange workspace
struct Exchange {
    uint64_t bound = 0;                          // serial of the communicator the layout below was gathered on
    size_t nparts = 0;
    std::vector<uint64_t> layout;                // [rank][part][2] = slot_begin, slot_count
    std::vector<uint64_t> local_n;               // [rank] score slots per query on that rank
    DevBuf<uint8_t> staging;                     // received slices, rank after rank
    DevBuf<uint8_t> global;                      // assembled rows
    DevBuf<uint64_t> d_meta;                     // small device scratch for size exchanges
    // ... and its PINNED host side: a copy to or from pageable memory makes hipMemcpyAsync wait for the stream, i.e.
    // for the collective in front of it -- inside the runtime, where no time limit reaches (sync_bounded below)
    PinnedBuf<uint64_t> h_meta;
    DevBuf<HitDev> hits_all;                     // gathered hit pools
    DevBuf<HitDev> hits_bucketed;                // this rank's pool, bucketed by the rank that owns each record's query
    DevBuf<unsigned long long> d_cursor;         // [nranks] bucket counts / cursors
    DevBuf<uint2> topk_all;
    DevBuf<uint32_t> topk_cnt_all;
    DevBuf<uint2> topk_merge, topk_final;        // ... side by side per (file, query) | merged by K3: the global k best, [file][query][k]
    DevBuf<uint32_t> topk_final_cnt;
    bool topk_merged = false;                    // the last exchange of best-of lists merged them on the device
    PinnedBuf<uint8_t> h_topk;                   // ... as they land on the host: the merged lists, then their counts
    uint64_t bytes_moved = 0;                    // bytes this rank received over the fabric in the last exchange
    // the ranks' agreement on a pass of the sharded search (sharded.cpp): one record per rank, all-gathered
    DevBuf<uint64_t> d_pass;                     // [1 + N][4]: this rank's record, then every rank's
    PinnedBuf<uint64_t> h_pass;                  // [N][4]
    hipEvent_t pass_ev = nullptr;                // ... have landed
    hipEvent_t x_done = nullptr;                 // the exchange of the pass is through (what ranks on the batch's own stream waits for)
    ~Exchange() {
        if (pass_ev) (void)hipEventDestroy(pass_ev);
        if (x_done) (void)hipEventDestroy(x_done);
    }
};


inline cobs_gpu_status nccl_fail(ncclResult_t r, const char* what) {
    return fail(COBS_GPU_ERR_RCCL, std::string(what) + ": " + ncclGetErrorString(r));
}

// (calls that involve no communicator: ncclGetUniqueId, ncclCommInitRank)
#define NCCL_TRY(expr)                                         \
    do {                                                       \
        ncclResult_t _r = (expr);                              \
        if (_r != ncclSuccess) return nccl_fail(_r, #expr);    \
    } while (0)

inline void comm_settle(cobs_gpu_comm* c) {
    if (c->abort_wanted && c->comm && !c->group_open) {
        (void)ncclCommAbort(c->comm);       // frees the communicator and releases kernels of it that wait for peers
        c->comm = nullptr;
        c->abort_wanted = false;
    }
}

// statuses after which the communicator itself is gone (a wrong argument or a misuse leaves it alive)
inline bool comm_is_dead(ncclResult_t r) {
    return r == ncclUnhandledCudaError || r == ncclSystemError || r == ncclInternalError || r == ncclRemoteError;
}

inline cobs_gpu_status comm_fail(cobs_gpu_comm* c, ncclResult_t r, const char* what) {
    if (!c->broken.load()) {
        c->broken_why = std::string(what) + ": " + ncclGetErrorString(r);
        c->broken.store(true);
    }
    if (comm_is_dead(r)) c->abort_wanted = true;
    comm_settle(c);
    return nccl_fail(r, what);
}

inline cobs_gpu_status comm_usable(const cobs_gpu_comm* c) {
    if (c->broken.load() || !c->comm)
        return fail(COBS_GPU_ERR_RCCL, "the communicator is unusable after an earlier failure (" + c->broken_why +
                                       "): destroy it and create a new one on every rank");
    return COBS_GPU_OK;
}

// every RCCL call on a communicator: counted for the watchdog, a failure marks the communicator
#define NCCL_C(c, st, expr)                                                    \
    do {                                                                       \
        (c)->last_op.store(#expr);                                             \
        (c)->last_stream.store((void*)(st));                                   \
        (c)->entered.fetch_add(1);                                             \
        ncclResult_t _r = (expr);                                              \
        (c)->returned.fetch_add(1);                                            \
        if (_r != ncclSuccess) return comm_fail((c), _r, #expr);               \
    } while (0)

// ncclGroupStart ... ncclGroupEnd with the end GUARANTEED: a send or receive that fails inside the group returns from
// the function through NCCL_C, and this scope's destructor still closes the group -- RCCL's group state is per
// thread, an open group would swallow every later call of this thread into a group that is never launched -- and only
// then lets comm_settle abort a dead communicator.  [VERDICT r4 9a: NCCL_TRY inside a group returned without closing it.]
struct GroupScope {
    cobs_gpu_comm* c;
    bool open = false;
    explicit GroupScope(cobs_gpu_comm* c_) : c(c_) {}
    cobs_gpu_status start() {
        NCCL_C(c, nullptr, ncclGroupStart());
        open = c->group_open = true;
        return COBS_GPU_OK;
    }
    cobs_gpu_status end(hipStream_t st) {
        open = c->group_open = false;
        NCCL_C(c, st, ncclGroupEnd());
        return COBS_GPU_OK;
    }
    ~GroupScope() {
        if (open) {
            (void)ncclGroupEnd();
            c->group_open = false;
            comm_settle(c);
        }
    }
};
#define GROUP_START(g) do { cobs_gpu_status _gs = (g).start(); if (_gs != COBS_GPU_OK) return _gs; } while (0)
#define GROUP_END(g, st) do { cobs_gpu_status _gs = (g).end(st); if (_gs != COBS_GPU_OK) return _gs; } while (0)

// Wait for a stream that carries a collective.  With a time limit on the communicator (cobs_gpu_comm_set_timeout) the
// wait gives up after it: a peer that never entered the collective would otherwise keep this rank here for ever.
// The communicator is aborted then (its kernels stop waiting) and the call fails with ERR_RCCL.
inline cobs_gpu_status sync_bounded(cobs_gpu_comm* c, hipStream_t st, const char* what) {
    if (!c || c->timeout_ms == 0) {
        HIP_TRY(hipStreamSynchronize(st));
        return COBS_GPU_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return COBS_GPU_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); HIP_TRY(e); }
        (void)hipGetLastError();
        if (spin > 4000) std::this_thread::sleep_for(std::chrono::microseconds(100));
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > (long long)c->timeout_ms) {
            if (!c->broken.load()) {
                c->broken_why = std::string(what) + " did not complete within " + std::to_string(c->timeout_ms) +
                                " ms (last call: " + c->last_op.load() + "): a peer never entered it, or the fabric stalled";
                c->broken.store(true);
            }
            c->abort_wanted = true;
            comm_settle(c);
            return fail(COBS_GPU_ERR_RCCL, c->broken_why);
        }
    }
}

// ... and for ONE event on such a stream (later work may already be queued behind it)
inline cobs_gpu_status event_bounded(cobs_gpu_comm* c, hipEvent_t ev, const char* what) {
    if (!c || c->timeout_ms == 0) {
        HIP_TRY(hipEventSynchronize(ev));
        return COBS_GPU_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return COBS_GPU_OK;
        if (e != hipErrorNotReady) { (void)hipGetLastError(); HIP_TRY(e); }
        (void)hipGetLastError();
        if (spin > 4000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > (long long)c->timeout_ms) {
            if (!c->broken.load()) {
                c->broken_why = std::string(what) + " did not complete within " + std::to_string(c->timeout_ms) +
                                " ms (last call: " + c->last_op.load() + "): a peer never entered it, or the fabric stalled";
                c->broken.store(true);
            }
            c->abort_wanted = true;
            comm_settle(c);
            return fail(COBS_GPU_ERR_RCCL, c->broken_why);
        }
    }
}

// ---- comm.cpp: the launch halves of the exchanges (everything queued on `st`, no host wait) -- what the pipelined
// sharded search strings together; the C entry points (cobs_gpu_batch_exchange_*) are these plus the waits.
// Hit records to every rank, the pool fills n[0 .. N) already known to all: *pool / *total = the records this rank
// ends up with (one rank: its own pool where it lies, no copy).
cobs_gpu_status xchg_hits_launch(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st, const uint64_t* n, const HitDev** pool,
                                 uint64_t* total);
// The shards' best-of lists: all-gathered and on their way into pinned memory | merged per (file, query) on the host
cobs_gpu_status xchg_topk_launch(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st);
cobs_gpu_status xchg_topk_collect(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st, bool waited = false);

}  // namespace cobs_amd
