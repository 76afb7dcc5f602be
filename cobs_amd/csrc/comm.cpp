// cobs_amd/csrc/comm.cpp -- the multi-GPU half of libcobs_gpu.so: RCCL communicators and the
// one exchange step of the sub-index-sharded layout.
//
// The reference has no distributed code.  What makes the path shardable is its own data
// layout: a compact index is a concatenation of sub-indexes over disjoint, contiguous
// document ranges (reference cobs/query/compact_index/mmap_search_file.cpp:22-27,
// search_file.cpp:30-32), and per-document counts never combine across sub-indexes or row-byte
// columns.  Every GPU stages and scans only its slice for the whole query batch
// (cobs_gpu_options.shard_rank / shard_count) and ends with the counts of ITS documents;
// one exchange per batch then brings the disjoint slices together:
//
//   counts, ALLTOALL  rank j owns the queries [nq*j/N, nq*(j+1)/N): every rank sends it the rows
//                     of those queries (grouped ncclSend/ncclRecv, the slices travel as bytes --
//                     RCCL has no 16-bit integer type).  Each count crosses xGMI once, to one
//                     GPU: (N-1)/N of the score matrix in total, spread over all N*(N-1) links.
//   counts, ALLGATHER every rank receives every slice (ncclAllGather when the slices have the
//                     same size, else grouped ncclSend/ncclRecv): N-1 times the traffic; the
//                     parity / "every rank ranks everything" mode.
//   hit lists         with a threshold only (query, file, doc, score) records leave a shard:
//                     sizes first (ncclAllGather of the pool fills), then the records -- to every rank
//                     (cobs_gpu_batch_exchange_hits), or each record once, to the rank that owns its query
//                     (cobs_gpu_batch_exchange_hits_owned: bucketed on the device, xchg_kernels.hip).
//   top-k             the k best of every shard (K3 output, same size on every rank):
//                     one ncclAllGather; the global top-k is a subset of their union.
//
// The received slices are assembled into rows in global document order with strided
// device-to-device copies, so that everything downstream (K3, ranking, D2H) sees the same
// layout as on one GPU.  All calls are collective: every rank of the communicator makes the
// same call with the same batch contents.
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "comm.hpp"

using namespace cobs_amd;

namespace cobs_amd {
void destroy_exchange(Exchange* x) { delete x; }
}  // namespace cobs_amd

namespace {

// every rank's score-slot layout, gathered once per (batch, communicator)
cobs_gpu_status bind_layout(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st) {
    if (!b->xchg) b->xchg = new Exchange;
    Exchange& x = *b->xchg;
    if (x.bound == c->serial) return COBS_GPU_OK;
    const cobs_gpu_index* ix = b->ix;
    const size_t np = ix->parts.size(), per = 2 * np + 2, N = (size_t)c->nranks;
    HIP_TRY(x.h_meta.reserve(per * (N + 1)));
    uint64_t* mine = x.h_meta.p;
    mine[0] = np;
    mine[1] = ix->total_counts;
    for (size_t f = 0; f < np; ++f) {
        mine[2 + 2 * f] = ix->parts[f].slot_begin;
        mine[3 + 2 * f] = ix->parts[f].slot_count;
    }
    HIP_TRY(x.d_meta.reserve(per * (N + 1)));
    HIP_TRY(hipMemcpyAsync(x.d_meta.p, mine, per * 8, hipMemcpyHostToDevice, st));
    NCCL_C(c, st, ncclAllGather(x.d_meta.p, x.d_meta.p + per, per * 8, ncclUint8, c->comm, st));
    const uint64_t* all = x.h_meta.p + per;
    HIP_TRY(hipMemcpyAsync(x.h_meta.p + per, x.d_meta.p + per, per * N * 8, hipMemcpyDeviceToHost, st));
    if (cobs_gpu_status ws = sync_bounded(c, st, "the all-gather of the shard layouts"); ws != COBS_GPU_OK) return ws;
    x.layout.assign(N * np * 2, 0);
    x.local_n.assign(N, 0);
    for (size_t r = 0; r < N; ++r) {
        const uint64_t* m = all + r * per;
        if (m[0] != np || m[1] != ix->total_counts)
            return fail(COBS_GPU_ERR_ARG, "the ranks of the communicator did not open the same index files");
        for (size_t f = 0; f < np; ++f) {
            x.layout[(r * np + f) * 2] = m[2 + 2 * f];
            x.layout[(r * np + f) * 2 + 1] = m[3 + 2 * f];
            x.local_n[r] += m[3 + 2 * f];
        }
    }
    x.nparts = np;
    x.bound = c->serial;
    return COBS_GPU_OK;
}

// The exchange of the count slices as a PLAN: which byte ranges of this rank's count rows go to
// which peer, where the peers' slices land in the staging buffer, and the strided copies that
// assemble them into rows in global document order.  Pure host arithmetic on the ranks' slot
// layouts -- the RCCL path below executes it, cobs_gpu_exchange_plan exposes it so that the
// N > 1 arithmetic (matching send / receive sizes on both ends, full coverage of the assembled
// rows) is tested without N GPUs.
struct XferPlan {
    size_t q_begin = 0, q_count = 0;             // queries whose assembled rows this rank ends up with
    bool use_allgather = false;                  // same-size slices, every rank takes everything: one ncclAllGather
    size_t staging_bytes = 0, global_bytes = 0, my_row_bytes = 0;
    std::vector<cobs_gpu_xfer> xfers;            // one per rank (self included: its sizes are 0 unless use_allgather)
    std::vector<cobs_gpu_copy2d> copies;         // assembly
};

XferPlan plan_exchange(const uint64_t* layout /*[N][F][2]*/, const uint64_t* doc_offset /*[F]*/, size_t N, size_t F,
                       uint64_t total_counts, size_t nq, size_t eb, uint32_t mode, size_t me) {
    XferPlan p;
    std::vector<uint64_t> local_n(N, 0);
    for (size_t r = 0; r < N; ++r)
        for (size_t f = 0; f < F; ++f) local_n[r] += layout[(r * F + f) * 2 + 1];
    auto q0_of = [&](size_t j) { return mode == COBS_GPU_XCHG_ALLTOALL ? nq * j / N : (size_t)0; };
    auto q1_of = [&](size_t j) { return mode == COBS_GPU_XCHG_ALLTOALL ? nq * (j + 1) / N : nq; };
    p.q_begin = q0_of(me);
    p.q_count = q1_of(me) - p.q_begin;
    p.my_row_bytes = (size_t)(local_n[me] * eb);
    p.global_bytes = (size_t)(p.q_count * total_counts * eb);
    p.use_allgather = mode == COBS_GPU_XCHG_ALLGATHER;
    for (size_t r = 0; r < N; ++r) p.use_allgather = p.use_allgather && local_n[r] == local_n[me];
    p.xfers.assign(N, cobs_gpu_xfer{});
    std::vector<size_t> src_off(N, 0);           // where rank r's block starts in the staging buffer
    if (p.use_allgather) {
        // staging = [rank][nq][n] incl. our own block
        for (size_t r = 0; r < N; ++r) {
            src_off[r] = r * nq * p.my_row_bytes;
            p.xfers[r].peer = r;
            p.xfers[r].send_offset = 0;
            p.xfers[r].send_bytes = nq * p.my_row_bytes;
            p.xfers[r].recv_offset = src_off[r];
            p.xfers[r].recv_bytes = nq * p.my_row_bytes;
        }
        p.staging_bytes = N * nq * p.my_row_bytes;
    } else {
        size_t off = 0;
        for (size_t r = 0; r < N; ++r) {
            cobs_gpu_xfer& x = p.xfers[r];
            x.peer = r;
            if (r == me) continue;               // our own slice is assembled straight from the count rows
            const size_t sq0 = q0_of(r), snq = q1_of(r) - sq0;
            x.send_offset = sq0 * p.my_row_bytes;
            x.send_bytes = snq * p.my_row_bytes;
            x.recv_offset = off;
            x.recv_bytes = (size_t)(p.q_count * local_n[r] * eb);
            src_off[r] = off;
            off += x.recv_bytes;
        }
        p.staging_bytes = off;
    }
    for (size_t r = 0; r < N; ++r) {
        uint64_t local_off = 0;
        for (size_t f = 0; f < F; ++f) {
            const uint64_t begin = layout[(r * F + f) * 2], count = layout[(r * F + f) * 2 + 1];
            if (count && p.q_count) {
                cobs_gpu_copy2d c{};
                c.src_rank = r;
                c.src_is_local = (!p.use_allgather && r == me) ? 1 : 0;
                c.src_offset = (c.src_is_local ? p.q_begin * p.my_row_bytes : src_off[r]) + local_off * eb;
                c.src_pitch = local_n[r] * eb;
                c.dst_offset = (doc_offset[f] + begin) * eb;
                c.dst_pitch = total_counts * eb;
                c.width = count * eb;
                c.height = p.q_count;
                p.copies.push_back(c);
            }
            local_off += count;
        }
    }
    return p;
}

}  // namespace

extern "C" {

cobs_gpu_status cobs_gpu_exchange_plan(const uint64_t* slot_begin, const uint64_t* slot_count, const uint64_t* doc_offset,
                                       size_t nranks, size_t nfiles, uint64_t total_counts, size_t nq, uint32_t elem_bytes,
                                       uint32_t mode, size_t rank, cobs_gpu_xfer* xfers, cobs_gpu_copy2d* copies,
                                       size_t* n_copies, uint64_t out[6]) {
    if (!slot_begin || !slot_count || !doc_offset || !xfers || !copies || !n_copies || !out || nranks == 0 ||
        rank >= nranks || nfiles == 0 || mode > COBS_GPU_XCHG_ALLTOALL || (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4))
        return fail(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        std::vector<uint64_t> layout(nranks * nfiles * 2);
        for (size_t i = 0; i < nranks * nfiles; ++i) { layout[2 * i] = slot_begin[i]; layout[2 * i + 1] = slot_count[i]; }
        const XferPlan p = plan_exchange(layout.data(), doc_offset, nranks, nfiles, total_counts, nq, elem_bytes, mode, rank);
        if (p.copies.size() > *n_copies) { *n_copies = p.copies.size(); return fail(COBS_GPU_ERR_CAPACITY, "copy list too small"); }
        for (size_t r = 0; r < nranks; ++r) xfers[r] = p.xfers[r];
        for (size_t i = 0; i < p.copies.size(); ++i) copies[i] = p.copies[i];
        *n_copies = p.copies.size();
        out[0] = p.q_begin; out[1] = p.q_count; out[2] = p.staging_bytes; out[3] = p.global_bytes;
        out[4] = p.use_allgather ? 1 : 0; out[5] = p.my_row_bytes;
        return COBS_GPU_OK;
    });
}

// RCCL prints a version banner with printf when a process first initialises it.  stdout belongs to
// the caller (`cobs query` output, bench.py's one JSON line): while a communicator is being set up
// file descriptor 1 points at stderr, and the C stdio buffer is flushed before it is put back.
// Reference-counted: the worker threads of a device list are all inside ncclCommInitRank at once.
struct QuietStdout {
    static std::mutex& mu() { static std::mutex m; return m; }
    static int& depth() { static int d = 0; return d; }
    static int& saved() { static int fd = -1; return fd; }
    QuietStdout() {
        std::lock_guard<std::mutex> g(mu());
        if (depth()++ == 0) {
            std::fflush(stdout);
            saved() = ::dup(1);
            if (saved() >= 0) (void)::dup2(2, 1);
        }
    }
    ~QuietStdout() {
        std::lock_guard<std::mutex> g(mu());
        if (--depth() == 0 && saved() >= 0) {
            std::fflush(stdout);
            (void)::dup2(saved(), 1);
            ::close(saved());
            saved() = -1;
        }
    }
};

cobs_gpu_status cobs_gpu_comm_unique_id(uint8_t id[COBS_GPU_UNIQUE_ID_BYTES]) {
    if (!id) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    static_assert(sizeof(ncclUniqueId) == COBS_GPU_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    {
        QuietStdout quiet;
        NCCL_TRY(ncclGetUniqueId(&u));
    }
    std::memcpy(id, &u, sizeof u);
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_comm_create(const uint8_t id[COBS_GPU_UNIQUE_ID_BYTES], int rank, int nranks, int device,
                                     cobs_gpu_comm** out) {
    if (!out) return fail(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!id || nranks <= 0 || rank < 0 || rank >= nranks) return fail(COBS_GPU_ERR_ARG, "bad rank / nranks / id");
    return guarded([&]() -> cobs_gpu_status {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            (void)hipGetLastError();
            return fail(COBS_GPU_ERR_NO_DEVICE, "no HIP device visible; libcobs_gpu has no CPU fallback");
        }
        if (device < 0) HIP_TRY(hipGetDevice(&device));
        if (device >= ndev) return fail(COBS_GPU_ERR_ARG, "device ordinal out of range");
        HIP_TRY(hipSetDevice(device));
        std::unique_ptr<cobs_gpu_comm> c(new cobs_gpu_comm);
        c->rank = rank;
        c->nranks = nranks;
        c->device = device;
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        {
            QuietStdout quiet;
            NCCL_TRY(ncclCommInitRank(&c->comm, nranks, u, rank));
        }
        static std::atomic<uint64_t> next_serial{1};
        c->serial = next_serial.fetch_add(1);
        // (a default time limit for callers that cannot set one -- the device-list handle's worker threads, the CLI's -d)
        if (const char* e = getenv("COBS_GPU_COMM_TIMEOUT_MS")) c->timeout_ms = (uint32_t)std::strtoul(e, nullptr, 0);
        *out = c.release();
        return COBS_GPU_OK;
    });
}

void cobs_gpu_comm_destroy(cobs_gpu_comm* c) {
    if (!c) return;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        // (a communicator that failed is aborted, not destroyed: ncclCommDestroy waits for outstanding operations, and
        // those of a broken communicator may wait for peers that are gone)
        if (c->broken.load()) (void)ncclCommAbort(c->comm);
        else (void)ncclCommDestroy(c->comm);
    }
    delete c;
}

int cobs_gpu_comm_rank(const cobs_gpu_comm* c) {
    int r = -1;
    if (c && c->comm && ncclCommUserRank(c->comm, &r) == ncclSuccess) return r;
    return -1;
}

int cobs_gpu_comm_size(const cobs_gpu_comm* c) {
    int n = 0;
    if (c && c->comm && ncclCommCount(c->comm, &n) == ncclSuccess) return n;
    return 0;
}

void cobs_gpu_comm_set_timeout(cobs_gpu_comm* c, uint32_t timeout_ms) {
    if (c) c->timeout_ms = timeout_ms;
}

// What this rank's communicator entered last, as text -- safe to call from ANOTHER thread while the owner is inside a
// call (the caller's watchdog): only atomics are read, plus one hipStreamQuery of the stream of the last call.
size_t cobs_gpu_comm_state(const cobs_gpu_comm* c, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return 0;
    const uint64_t e = c->entered.load(), r = c->returned.load();
    const char* op = c->last_op.load();
    void* st = c->last_stream.load();
    const char* stream = "no stream";
    if (st || e) {
        const hipError_t q = hipStreamQuery((hipStream_t)st);
        (void)hipGetLastError();
        stream = q == hipSuccess ? "stream idle" : q == hipErrorNotReady ? "stream busy" : "stream in error";
    }
    const int n = std::snprintf(buf, cap, "rank %d/%d device %d: rccl calls entered %llu returned %llu%s; last: %s; %s; %s", c->rank,
                                c->nranks, c->device, (unsigned long long)e, (unsigned long long)r,
                                e != r ? " (INSIDE a call)" : "", op, stream,
                                c->broken.load() ? "communicator BROKEN" : "communicator ok");
    return n < 0 ? 0 : std::min<size_t>((size_t)n, cap - 1);
}

// A communicator that came up is not yet one that moves bytes: the first collective is where a fabric or IPC
// problem shows (peer memory that cannot be mapped, a link that is down), as a hang or as an error.  This runs, under
// a time limit each, what the exchange of a batch uses -- one grouped ncclSend / ncclRecv all-to-all with a DIFFERENT
// size for every (sender, receiver) pair, one ncclAllGather, one ncclAllReduce -- on small self-describing payloads,
// and checks every received byte.  big_bytes > 0 adds a timed all-to-all of that many bytes per pair (what a link gives).
// out[0..7] = all-to-all bytes received | its microseconds | all-gather us | all-reduce us | big bytes received | big us | 0 | 0
cobs_gpu_status cobs_gpu_comm_preflight(cobs_gpu_comm* c, uint32_t timeout_ms, uint64_t big_bytes, uint64_t out[8]) {
    if (!c || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    return guarded([&]() -> cobs_gpu_status {
        for (int i = 0; i < 8; ++i) out[i] = 0;
        HIP_TRY(hipSetDevice(c->device));
        if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
        const uint32_t keep_timeout = c->timeout_ms;
        struct Restore { cobs_gpu_comm* c; uint32_t t; ~Restore() { c->timeout_ms = t; } } restore{c, keep_timeout};
        c->timeout_ms = timeout_ms;
        const size_t N = (size_t)c->nranks, me = (size_t)c->rank;
        auto pair_bytes = [&](size_t from, size_t to) { return (size_t)(1 + (from * 7 + to * 3) % 5) * 1024 + 16 * from + to; };
        auto pattern = [&](size_t from, size_t to, size_t k) { return (uint8_t)(from * 31 + to * 17 + k * 7 + 1); };
        struct StreamOwner { hipStream_t s = nullptr; ~StreamOwner() { if (s) (void)hipStreamDestroy(s); } } so;
        HIP_TRY(hipStreamCreateWithFlags(&so.s, hipStreamNonBlocking));
        hipStream_t st = so.s;
        auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                                      std::chrono::steady_clock::now().time_since_epoch()).count(); };
        // (1) all-to-all, uneven sizes
        size_t send_total = 0, recv_total = 0;
        std::vector<size_t> soff(N + 1, 0), roff(N + 1, 0);
        for (size_t j = 0; j < N; ++j) {
            soff[j + 1] = soff[j] + (j == me ? 0 : pair_bytes(me, j));
            roff[j + 1] = roff[j] + (j == me ? 0 : pair_bytes(j, me));
        }
        send_total = soff[N];
        recv_total = roff[N];
        std::vector<uint8_t> h_send(std::max<size_t>(send_total, 1)), h_recv(std::max<size_t>(recv_total, 1), 0);
        for (size_t j = 0; j < N; ++j)
            for (size_t k = 0; k < soff[j + 1] - soff[j]; ++k) h_send[soff[j] + k] = pattern(me, j, k);
        DevBuf<uint8_t> d_send, d_recv;
        HIP_TRY(d_send.reserve(std::max<size_t>(send_total, 1)));
        HIP_TRY(d_recv.reserve(std::max<size_t>(recv_total, 1)));
        HIP_TRY(hipMemcpyAsync(d_send.p, h_send.data(), send_total, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(d_recv.p, 0, std::max<size_t>(recv_total, 1), st));
        HIP_TRY(hipStreamSynchronize(st));
        uint64_t t0 = now_us();
        if (N > 1) {
            GroupScope grp(c);
            GROUP_START(grp);
            for (size_t j = 0; j < N; ++j) {
                if (j == me) continue;
                NCCL_C(c, st, ncclSend(d_send.p + soff[j], soff[j + 1] - soff[j], ncclUint8, (int)j, c->comm, st));
                NCCL_C(c, st, ncclRecv(d_recv.p + roff[j], roff[j + 1] - roff[j], ncclUint8, (int)j, c->comm, st));
            }
            GROUP_END(grp, st);
        }
        if (cobs_gpu_status ws = sync_bounded(c, st, "preflight: the grouped send / receive all-to-all"); ws != COBS_GPU_OK) return ws;
        out[0] = recv_total;
        out[1] = now_us() - t0;
        if (recv_total) HIP_TRY(hipMemcpy(h_recv.data(), d_recv.p, recv_total, hipMemcpyDeviceToHost));
        for (size_t j = 0; j < N; ++j)
            for (size_t k = 0; k < roff[j + 1] - roff[j]; ++k)
                if (h_recv[roff[j] + k] != pattern(j, me, k))
                    return comm_fail(c, ncclInternalError, ("preflight: the all-to-all delivered a wrong byte from rank " + std::to_string(j)).c_str());
        // (2) all-gather
        const size_t gb = 4096;
        DevBuf<uint8_t> d_g;
        HIP_TRY(d_g.reserve(gb * (N + 1)));
        std::vector<uint8_t> h_g(gb * (N + 1));
        for (size_t k = 0; k < gb; ++k) h_g[k] = pattern(me, N, k);
        HIP_TRY(hipMemcpyAsync(d_g.p, h_g.data(), gb, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        t0 = now_us();
        NCCL_C(c, st, ncclAllGather(d_g.p, d_g.p + gb, gb, ncclUint8, c->comm, st));
        if (cobs_gpu_status ws = sync_bounded(c, st, "preflight: ncclAllGather"); ws != COBS_GPU_OK) return ws;
        out[2] = now_us() - t0;
        HIP_TRY(hipMemcpy(h_g.data(), d_g.p, gb * (N + 1), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < N; ++j)
            for (size_t k = 0; k < gb; ++k)
                if (h_g[gb * (j + 1) + k] != pattern(j, N, k))
                    return comm_fail(c, ncclInternalError, ("preflight: ncclAllGather delivered a wrong byte from rank " + std::to_string(j)).c_str());
        // (3) all-reduce (max and sum of 32-bit words, as the status agreements use it)
        DevBuf<uint32_t> d_r;
        HIP_TRY(d_r.reserve(64));
        uint32_t h_r[32];
        for (uint32_t k = 0; k < 16; ++k) h_r[k] = (uint32_t)(me + 1) * (k + 1);
        HIP_TRY(hipMemcpyAsync(d_r.p, h_r, 64, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        t0 = now_us();
        NCCL_C(c, st, ncclAllReduce(d_r.p, d_r.p + 16, 16, ncclUint32, ncclMax, c->comm, st));
        NCCL_C(c, st, ncclAllReduce(d_r.p, d_r.p + 32, 16, ncclUint32, ncclSum, c->comm, st));
        if (cobs_gpu_status ws = sync_bounded(c, st, "preflight: ncclAllReduce"); ws != COBS_GPU_OK) return ws;
        out[3] = now_us() - t0;
        uint32_t h_o[32];
        HIP_TRY(hipMemcpy(h_o, d_r.p + 16, 128, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < 16; ++k)
            if (h_o[k] != (uint32_t)N * (k + 1) || h_o[16 + k] != (uint32_t)(N * (N + 1) / 2) * (k + 1))
                return comm_fail(c, ncclInternalError, "preflight: ncclAllReduce returned a wrong value");
        // (4) what a pair of links gives: every rank sends big_bytes to every other rank
        if (big_bytes && N > 1) {
            DevBuf<uint8_t> d_bs, d_br;
            HIP_TRY(d_bs.reserve((size_t)big_bytes));
            HIP_TRY(d_br.reserve((size_t)big_bytes * (N - 1)));
            HIP_TRY(hipMemsetAsync(d_bs.p, (int)(me + 1), (size_t)big_bytes, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (int rep_i = 0; rep_i < 2; ++rep_i) {          // (the first round sets the connections up)
                t0 = now_us();
                GroupScope grp(c);
                GROUP_START(grp);
                size_t slot = 0;
                for (size_t j = 0; j < N; ++j) {
                    if (j == me) continue;
                    NCCL_C(c, st, ncclSend(d_bs.p, (size_t)big_bytes, ncclUint8, (int)j, c->comm, st));
                    NCCL_C(c, st, ncclRecv(d_br.p + slot * big_bytes, (size_t)big_bytes, ncclUint8, (int)j, c->comm, st));
                    ++slot;
                }
                GROUP_END(grp, st);
                if (cobs_gpu_status ws = sync_bounded(c, st, "preflight: the large all-to-all"); ws != COBS_GPU_OK) return ws;
                out[5] = now_us() - t0;
            }
            out[4] = big_bytes * (N - 1);
            // first and last byte of every peer's block
            size_t slot = 0;
            for (size_t j = 0; j < N; ++j) {
                if (j == me) continue;
                uint8_t ends[2] = {0, 0};
                HIP_TRY(hipMemcpy(&ends[0], d_br.p + slot * big_bytes, 1, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(&ends[1], d_br.p + (slot + 1) * big_bytes - 1, 1, hipMemcpyDeviceToHost));
                if (ends[0] != (uint8_t)(j + 1) || ends[1] != (uint8_t)(j + 1))
                    return comm_fail(c, ncclInternalError, ("preflight: the large all-to-all delivered wrong bytes from rank " + std::to_string(j)).c_str());
                ++slot;
            }
        }
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_batch_exchange_counts(cobs_gpu_batch* b, cobs_gpu_comm* c, uint32_t mode, void* hip_stream) {
    if (!b || !c) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (!b->ran || !b->have_counts) return fail(COBS_GPU_ERR_ARG, "run the batch with score rows first");
    if (mode > COBS_GPU_XCHG_REDUCE) return fail(COBS_GPU_ERR_ARG, "unknown exchange mode");
    return guarded([&]() -> cobs_gpu_status {
        hipStream_t st = (hipStream_t)hip_stream;
        const cobs_gpu_index* ix = b->ix;
        HIP_TRY(hipSetDevice(ix->device));
        if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
        cobs_gpu_status s = bind_layout(b, c, st);
        if (s != COBS_GPU_OK) return s;
        Exchange& x = *b->xchg;
        const size_t N = (size_t)c->nranks, me = (size_t)c->rank, nq = b->nq;
        std::vector<uint64_t> doc_off(x.nparts);
        for (size_t f = 0; f < x.nparts; ++f) doc_off[f] = ix->parts[f].doc_offset;
        if (mode == COBS_GPU_XCHG_REDUCE) {
            // "per-document hit counts reduced over RCCL" taken literally (parity mode):
            // every rank lays its slices into zeroed rows of global length and the rows are summed
            // with one ncclAllReduce.  The slices are disjoint, so every BYTE of the result has one
            // non-zero contributor: a sum over ncclUint8 cannot carry and is exact for 8-, 16- and
            // 32-bit counters alike (RCCL has no 16-bit integer type).  Moves ~2x the full vector
            // through every GPU (ring all-reduce) where the gather forms move 7/8 of it once.
            const size_t eb = b->elem_bytes, row = (size_t)ix->total_counts * eb, bytes = nq * row;
            HIP_TRY(x.global.reserve(std::max<size_t>(bytes, 1)));
            HIP_TRY(hipMemsetAsync(x.global.p, 0, bytes, st));
            uint64_t local_off = 0;
            const uint64_t my_n = x.local_n[me];
            for (size_t f = 0; f < x.nparts; ++f) {
                const uint64_t begin = x.layout[(me * x.nparts + f) * 2], count = x.layout[(me * x.nparts + f) * 2 + 1];
                if (count && nq)
                    HIP_TRY(hipMemcpy2DAsync(x.global.p + (doc_off[f] + begin) * eb, row, b->counts.p + local_off * eb,
                                             (size_t)(my_n * eb), (size_t)(count * eb), nq, hipMemcpyDeviceToDevice, st));
                local_off += count;
            }
            if (bytes) NCCL_C(c, st, ncclAllReduce(x.global.p, x.global.p, bytes, ncclUint8, ncclSum, c->comm, st));
            x.bytes_moved = N > 1 ? 2 * (N - 1) * bytes / N : 0;
            b->g_rows = x.global.p;
            b->g_q0 = 0;
            b->g_qn = nq;
            b->view_global = true;
            b->rows_q0 = b->rows_q1 = 0;
            HIP_TRY(hipEventRecord(b->run_done, st));
            return COBS_GPU_OK;
        }
        const XferPlan p = plan_exchange(x.layout.data(), doc_off.data(), N, x.nparts, ix->total_counts, nq, b->elem_bytes, mode, me);
        HIP_TRY(x.staging.reserve(std::max<size_t>(p.staging_bytes, 1)));
        HIP_TRY(x.global.reserve(std::max<size_t>(p.global_bytes, 1)));
        const uint8_t* mine = b->counts.p;
        if (p.use_allgather) {
            // same-size slices (always so on one rank): the library collective
            NCCL_C(c, st, ncclAllGather(mine, x.staging.p, nq * p.my_row_bytes, ncclUint8, c->comm, st));
            x.bytes_moved = (N - 1) * nq * p.my_row_bytes;
        } else {
            x.bytes_moved = p.staging_bytes;
            if (N > 1) {
                GroupScope grp(c);
                GROUP_START(grp);
                for (size_t j = 0; j < N; ++j) {
                    const cobs_gpu_xfer& t = p.xfers[j];
                    if (j == me) continue;
                    if (t.send_bytes) NCCL_C(c, st, ncclSend(mine + t.send_offset, t.send_bytes, ncclUint8, (int)j, c->comm, st));
                    if (t.recv_bytes) NCCL_C(c, st, ncclRecv(x.staging.p + t.recv_offset, t.recv_bytes, ncclUint8, (int)j, c->comm, st));
                }
                GROUP_END(grp, st);
            }
        }
        for (const cobs_gpu_copy2d& cp : p.copies)
            HIP_TRY(hipMemcpy2DAsync(x.global.p + cp.dst_offset, cp.dst_pitch,
                                     (cp.src_is_local ? mine : x.staging.p) + cp.src_offset, cp.src_pitch, cp.width, cp.height,
                                     hipMemcpyDeviceToDevice, st));
        const size_t my_q0 = p.q_begin, my_nq = p.q_count;
        b->g_rows = x.global.p;
        b->g_q0 = my_q0;
        b->g_qn = my_nq;
        b->view_global = true;
        b->rows_q0 = b->rows_q1 = 0;
        HIP_TRY(hipEventRecord(b->run_done, st));
        return COBS_GPU_OK;
    });
}

void* cobs_gpu_batch_global_counts_device(cobs_gpu_batch* b, uint64_t* q_begin, uint64_t* q_count,
                                          uint32_t* elem_bytes, uint64_t* row_stride_bytes) {
    if (!b || !b->view_global) return nullptr;
    if (q_begin) *q_begin = b->g_q0;
    if (q_count) *q_count = b->g_qn;
    if (elem_bytes) *elem_bytes = b->elem_bytes;
    if (row_stride_bytes) *row_stride_bytes = b->ix->total_counts * b->elem_bytes;
    return const_cast<uint8_t*>(b->g_rows);
}

uint64_t cobs_gpu_batch_exchange_bytes(const cobs_gpu_batch* b) { return b && b->xchg ? b->xchg->bytes_moved : 0; }

// Hit lists: call after cobs_gpu_batch_sync of a run with a threshold.  *overflow (optional) is
// set when some rank's pool overflowed -- the lists are then incomplete everywhere and the caller
// repeats the pass with score rows (every rank sees the same flag).
cobs_gpu_status cobs_gpu_batch_exchange_hits(cobs_gpu_batch* b, cobs_gpu_comm* c, void* hip_stream, int* overflow) {
    if (!b || !c) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    // (what the CALLER did -- the same on every rank of a collective call; what only this rank's handle decides
    // must not fail here, see `no_pool` below)
    if (!b->ran || !b->synced || !(b->threshold > 0.0) || b->topk_k != 0)
        return fail(COBS_GPU_ERR_ARG, "run the batch with a threshold (and no limit) and sync it first");
    return guarded([&]() -> cobs_gpu_status {
        hipStream_t st = (hipStream_t)hip_stream;
        HIP_TRY(hipSetDevice(b->ix->device));
        if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
        if (!b->xchg) b->xchg = new Exchange;
        Exchange& x = *b->xchg;
        const size_t N = (size_t)c->nranks;
        // A rank whose last run kept score rows instead of a hit pool (the caller repeated the pass with rows after an
        // overflow, say) must not leave the collective with an error of its own while its peers wait in it [ADVICE r4]:
        // it travels through the size exchange as an impossible fill, every rank sees *overflow = 1.  (Until round 6 a
        // handle with row-range chunks was always in that state; its scans select now: pass.cpp, acc_mode.)
        const bool no_pool = !b->selected;
        // sizes first
        const uint64_t mine = no_pool ? ~0ull : b->h_nhits();
        HIP_TRY(x.d_meta.reserve(N + 1 + 64));
        HIP_TRY(x.h_meta.reserve(N + 1 + 64));
        x.h_meta.p[0] = mine;
        HIP_TRY(hipMemcpyAsync(x.d_meta.p, x.h_meta.p, 8, hipMemcpyHostToDevice, st));
        NCCL_C(c, st, ncclAllGather(x.d_meta.p, x.d_meta.p + 1, 8, ncclUint8, c->comm, st));
        HIP_TRY(hipMemcpyAsync(x.h_meta.p + 1, x.d_meta.p + 1, 8 * N, hipMemcpyDeviceToHost, st));
        if (cobs_gpu_status ws = sync_bounded(c, st, "the all-gather of the hit-pool fills"); ws != COBS_GPU_OK) return ws;
        const std::vector<uint64_t> n(x.h_meta.p + 1, x.h_meta.p + 1 + N);
        bool over = false;
        for (size_t r = 0; r < N; ++r) over = over || n[r] > b->hit_cap;       // the pool capacity is a function of the batch: equal on all ranks
        if (overflow) *overflow = over ? 1 : 0;
        if (over) return COBS_GPU_OK;
        const HitDev* pool = nullptr;
        uint64_t got = 0;
        if (cobs_gpu_status ls = xchg_hits_launch(b, c, st, n.data(), &pool, &got); ls != COBS_GPU_OK) return ls;
        // (the bounded wait first: what follows waits for the stream inside the runtime)
        if (cobs_gpu_status ws = sync_bounded(c, st, "the exchange of the hit records"); ws != COBS_GPU_OK) return ws;
        // the gathered pools of all shards, put into result order on the device (results.cpp: order_pool; round 4
        // bucketed them with a counting sort on one host thread, after a copy to pageable memory)
        if (cobs_gpu_status os = order_pool(b, pool, got, st); os != COBS_GPU_OK) return os;
        b->pool_global = true;
        return COBS_GPU_OK;
    });
}

// The owner-routed hit exchange as a plan (host arithmetic): counts[r * nranks + j] = records rank r holds for the
// queries rank j owns.  Rank `rank` sends bucket j of its bucketed pool to rank j and receives bucket `rank` of
// every other rank; received records land rank after rank (its own bucket at its place).  Offsets / sizes in bytes.
cobs_gpu_status cobs_gpu_hit_exchange_plan(const uint64_t* counts, size_t nranks, size_t rank, cobs_gpu_xfer* xfers,
                                           uint64_t out[2]) {
    if (!counts || !xfers || !out || nranks == 0 || rank >= nranks) return fail(COBS_GPU_ERR_ARG, "bad argument");
    const uint64_t rec = sizeof(HitDev);
    uint64_t send_off = 0, recv_off = 0;
    for (size_t j = 0; j < nranks; ++j) {
        cobs_gpu_xfer& x = xfers[j];
        x.peer = j;
        x.send_offset = send_off * rec;
        x.send_bytes = counts[rank * nranks + j] * rec;
        x.recv_offset = recv_off * rec;
        x.recv_bytes = counts[j * nranks + rank] * rec;
        send_off += counts[rank * nranks + j];
        recv_off += counts[j * nranks + rank];
    }
    out[0] = recv_off * rec;        // bytes this rank ends up with (its own bucket included)
    out[1] = send_off * rec;        // bytes of its bucketed pool
    return COBS_GPU_OK;
}

// Hit lists routed to query owners: call after cobs_gpu_batch_sync of a run with a threshold.  Rank j owns the
// queries [nq*j/N, nq*(j+1)/N) (as in the all-to-all exchange of count rows) and receives the records of
// exactly those queries from every shard: each record crosses the fabric once, to one GPU, where
// cobs_gpu_batch_exchange_hits sends every record to every rank.  Sizes first (one ncclAllGather of the
// per-owner counts), then grouped ncclSend / ncclRecv.  *overflow as in cobs_gpu_batch_exchange_hits.
cobs_gpu_status cobs_gpu_batch_exchange_hits_owned(cobs_gpu_batch* b, cobs_gpu_comm* c, void* hip_stream, int* overflow,
                                                   uint64_t* q_begin, uint64_t* q_count) {
    if (!b || !c) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (!b->ran || !b->synced || !(b->threshold > 0.0) || b->topk_k != 0)
        return fail(COBS_GPU_ERR_ARG, "run the batch with a threshold (and no limit) and sync it first");
    if (c->nranks > 64) return fail(COBS_GPU_ERR_UNSUPPORTED, "more than 64 ranks");
    return guarded([&]() -> cobs_gpu_status {
        hipStream_t st = (hipStream_t)hip_stream;
        HIP_TRY(hipSetDevice(b->ix->device));
        if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
        if (!b->xchg) b->xchg = new Exchange;
        Exchange& x = *b->xchg;
        const size_t N = (size_t)c->nranks, me = (size_t)c->rank;
        const uint64_t q0 = (uint64_t)b->nq * me / N, q1 = (uint64_t)b->nq * (me + 1) / N;
        if (q_begin) *q_begin = q0;
        if (q_count) *q_count = q1 - q0;
        // (a rank that kept score rows instead of a hit pool, see cobs_gpu_batch_exchange_hits, reports an overflow:
        // every rank then repeats the pass with score rows)
        const bool no_pool = !b->selected;
        const uint64_t mine = no_pool ? 0 : std::min<uint64_t>(b->h_nhits(), b->hit_cap);     // an overflowed pool is not routed (see below)
        const bool over_here = no_pool || b->h_nhits() > b->hit_cap;
        // (1) records per owner on this rank
        HIP_TRY(x.d_cursor.reserve(2 * N + 2));
        HIP_TRY(hipMemsetAsync(x.d_cursor.p, 0, 8 * N, st));
        BucketArgs ba;
        ba.hits = b->hits.p;
        ba.n = over_here ? 0 : mine;
        ba.cursor = x.d_cursor.p;
        ba.out = nullptr;
        ba.nq = (uint32_t)b->nq;
        ba.nranks = (uint32_t)N;
        HIP_TRY(launch_bucket_hits(ba, true, st));
        // (2) every rank's counts to every rank; an overflow travels as an impossible count
        HIP_TRY(x.d_meta.reserve(N * (N + 1) + 64));
        std::vector<unsigned long long> cnt(N, 0);
        HIP_TRY(hipMemcpyAsync(cnt.data(), x.d_cursor.p, 8 * N, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (over_here) cnt[0] = ~0ull;
        HIP_TRY(hipMemcpyAsync(x.d_meta.p, cnt.data(), 8 * N, hipMemcpyHostToDevice, st));
        NCCL_C(c, st, ncclAllGather(x.d_meta.p, x.d_meta.p + N, 8 * N, ncclUint8, c->comm, st));
        HIP_TRY(x.h_meta.reserve(N * N + 64));
        HIP_TRY(hipMemcpyAsync(x.h_meta.p, x.d_meta.p + N, 8 * N * N, hipMemcpyDeviceToHost, st));
        if (cobs_gpu_status ws = sync_bounded(c, st, "the all-gather of the per-owner hit counts"); ws != COBS_GPU_OK) return ws;
        const std::vector<uint64_t> all(x.h_meta.p, x.h_meta.p + N * N);
        bool over = false;
        for (size_t r = 0; r < N; ++r) over = over || all[r * N] == ~0ull;
        if (overflow) *overflow = over ? 1 : 0;
        if (over) return COBS_GPU_OK;           // every rank sees the same flag and repeats the pass with score rows
        // (3) bucket the pool: cursors = start of every owner's bucket
        std::vector<unsigned long long> start(N, 0);
        for (size_t j = 1; j < N; ++j) start[j] = start[j - 1] + all[me * N + j - 1];
        HIP_TRY(x.hits_bucketed.reserve(std::max<size_t>((size_t)mine, 1)));
        HIP_TRY(hipMemcpyAsync(x.d_cursor.p, start.data(), 8 * N, hipMemcpyHostToDevice, st));
        ba.out = x.hits_bucketed.p;
        HIP_TRY(launch_bucket_hits(ba, false, st));
        // (4) the plan, then the records
        std::vector<cobs_gpu_xfer> xf(N);
        uint64_t tot[2];
        cobs_gpu_status ps = cobs_gpu_hit_exchange_plan(all.data(), N, me, xf.data(), tot);
        if (ps != COBS_GPU_OK) return ps;
        const uint64_t total = tot[0] / sizeof(HitDev);
        HIP_TRY(x.hits_all.reserve(std::max<size_t>((size_t)total, 1)));
        uint8_t* recv = reinterpret_cast<uint8_t*>(x.hits_all.p);
        const uint8_t* send = reinterpret_cast<const uint8_t*>(x.hits_bucketed.p);
        if (N > 1) {
            GroupScope grp(c);
            GROUP_START(grp);
            for (size_t j = 0; j < N; ++j) {
                if (j == me) continue;
                if (xf[j].send_bytes) NCCL_C(c, st, ncclSend(send + xf[j].send_offset, xf[j].send_bytes, ncclUint8, (int)j, c->comm, st));
                if (xf[j].recv_bytes) NCCL_C(c, st, ncclRecv(recv + xf[j].recv_offset, xf[j].recv_bytes, ncclUint8, (int)j, c->comm, st));
            }
            GROUP_END(grp, st);
        }
        if (xf[me].send_bytes)
            HIP_TRY(hipMemcpyAsync(recv + xf[me].recv_offset, send + xf[me].send_offset, xf[me].send_bytes, hipMemcpyDeviceToDevice, st));
        if (cobs_gpu_status ws = sync_bounded(c, st, "the owner-routed exchange of the hit records"); ws != COBS_GPU_OK) return ws;
        x.bytes_moved = tot[0] - xf[me].recv_bytes;
        // the records of the owned queries from every shard, put into result order on the device (order_pool)
        if (cobs_gpu_status os = order_pool(b, x.hits_all.p, total, st); os != COBS_GPU_OK) return os;
        b->pool_global = true;
        b->pool_owned = true;
        b->own_q0 = q0;
        b->own_qn = q1 - q0;
        return COBS_GPU_OK;
    });
}

// Diagnostics / tests: the device-side bucketing of the owner-routed exchange for ANY rank count, without a
// communicator -- the hit pool of the last synced run bucketed as rank `nranks` ranks would (counts[j] records for
// owner j, buckets back to back), copied to the host as (query, file, document, score) quadruples.
cobs_gpu_status cobs_gpu_batch_bucketed_hits(cobs_gpu_batch* b, uint32_t nranks, uint64_t* counts, uint32_t* records,
                                             size_t cap_records, size_t* n_records) {
    if (!b || !counts || !n_records || nranks == 0 || nranks > 64) return fail(COBS_GPU_ERR_ARG, "bad argument");
    if (!b->ran || !b->synced || !b->selected) return fail(COBS_GPU_ERR_ARG, "run the batch with a threshold and sync it first");
    if (b->h_nhits() > b->hit_cap) return fail(COBS_GPU_ERR_CAPACITY, "the hit pool overflowed");
    return guarded([&]() -> cobs_gpu_status {
        HIP_TRY(hipSetDevice(b->ix->device));
        if (!b->xchg) b->xchg = new Exchange;
        Exchange& x = *b->xchg;
        const uint64_t n = b->h_nhits();
        *n_records = (size_t)n;
        HIP_TRY(x.d_cursor.reserve(2 * (size_t)nranks + 2));
        HIP_TRY(hipMemset(x.d_cursor.p, 0, 8 * nranks));
        BucketArgs ba;
        ba.hits = b->hits.p; ba.n = n; ba.cursor = x.d_cursor.p; ba.out = nullptr; ba.nq = (uint32_t)b->nq; ba.nranks = nranks;
        HIP_TRY(launch_bucket_hits(ba, true, nullptr));
        std::vector<unsigned long long> cnt(nranks, 0), start(nranks, 0);
        HIP_TRY(hipMemcpy(cnt.data(), x.d_cursor.p, 8 * nranks, hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < nranks; ++j) counts[j] = cnt[j];
        for (uint32_t j = 1; j < nranks; ++j) start[j] = start[j - 1] + cnt[j - 1];
        if (n > cap_records || (n && !records)) return fail(COBS_GPU_ERR_CAPACITY, "record buffer too small");
        HIP_TRY(x.hits_bucketed.reserve(std::max<size_t>((size_t)n, 1)));
        HIP_TRY(hipMemcpy(x.d_cursor.p, start.data(), 8 * nranks, hipMemcpyHostToDevice));
        ba.out = x.hits_bucketed.p;
        HIP_TRY(launch_bucket_hits(ba, false, nullptr));
        static_assert(sizeof(HitDev) == 16, "records are four u32");
        if (n) HIP_TRY(hipMemcpy(records, x.hits_bucketed.p, n * sizeof(HitDev), hipMemcpyDeviceToHost));
        return COBS_GPU_OK;
    });
}

// Top-k candidates: call after a run with num_results > 0 (K3 ran).  Every rank ends up with the
// k best documents of every shard; cobs_gpu_batch_hits_host then merges them per query.
cobs_gpu_status cobs_gpu_batch_exchange_topk(cobs_gpu_batch* b, cobs_gpu_comm* c, void* hip_stream) {
    if (!b || !c) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    if (!b->ran || b->topk_k == 0) return fail(COBS_GPU_ERR_ARG, "run the batch with num_results > 0 first");
    return guarded([&]() -> cobs_gpu_status {
        hipStream_t st = (hipStream_t)hip_stream;
        if (cobs_gpu_status ls = xchg_topk_launch(b, c, st); ls != COBS_GPU_OK) return ls;
        return xchg_topk_collect(b, c, st);
    });
}

}  // extern "C"

namespace cobs_amd {

cobs_gpu_status xchg_hits_launch(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st, const uint64_t* n, const HitDev** pool,
                                 uint64_t* total_out) {
    HIP_TRY(hipSetDevice(b->ix->device));
    if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
    if (!b->xchg) b->xchg = new Exchange;
    Exchange& x = *b->xchg;
    const size_t N = (size_t)c->nranks, me = (size_t)c->rank;
    const uint64_t mine = n[me];
    if (N == 1) {                       // one rank: the pool where the scan left it
        *pool = b->hits.p;
        *total_out = mine;
        x.bytes_moved = 0;
        return COBS_GPU_OK;
    }
    uint64_t total = 0;
    std::vector<uint64_t> off(N + 1, 0);
    for (size_t r = 0; r < N; ++r) {
        off[r + 1] = off[r] + n[r];
        total += n[r];
    }
    HIP_TRY(x.hits_all.reserve(std::max<size_t>((size_t)total, 1)));
    {
        GroupScope grp(c);
        GROUP_START(grp);
        for (size_t j = 0; j < N; ++j) {
            if (j == me) continue;
            if (mine) NCCL_C(c, st, ncclSend(b->hits.p, mine * sizeof(HitDev), ncclUint8, (int)j, c->comm, st));
            if (n[j]) NCCL_C(c, st, ncclRecv(x.hits_all.p + off[j], n[j] * sizeof(HitDev), ncclUint8, (int)j, c->comm, st));
        }
        GROUP_END(grp, st);
    }
    if (mine)
        HIP_TRY(hipMemcpyAsync(x.hits_all.p + off[me], b->hits.p, mine * sizeof(HitDev), hipMemcpyDeviceToDevice, st));
    x.bytes_moved = (total - mine) * sizeof(HitDev);
    *pool = x.hits_all.p;
    *total_out = total;
    return COBS_GPU_OK;
}

// The shards' best-of lists: all-gathered, merged ON THE DEVICE into the global k best of every (file, query) -- K3's pool
// mode over the lists laid side by side (merge_lists_kernel) --, and on their way into pinned memory.  [Until round 6
// the N x k candidates of every (file, query) crossed PCIe and every query's list was merged by a host sort: ~0.5 us
// per query at eight ranks, several times the 2.4 ms a rank scans for at N = 8.]
cobs_gpu_status xchg_topk_launch(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st) {
    HIP_TRY(hipSetDevice(b->ix->device));
    if (cobs_gpu_status us = comm_usable(c); us != COBS_GPU_OK) return us;
    if (!b->xchg) b->xchg = new Exchange;
    Exchange& x = *b->xchg;
    const size_t N = (size_t)c->nranks, k = b->topk_k, nq = b->nq, np = b->ix->parts.size();
    const size_t ne = k * nq * np, nc = nq * np;
    const size_t stride = (size_t)round_up(N * k, 8);
    if (N * k > 0xFFFFFFF0ull) return fail(COBS_GPU_ERR_UNSUPPORTED, "too many candidates per query");
    HIP_TRY(x.topk_all.reserve(std::max<size_t>(N * ne, 1)));
    HIP_TRY(x.topk_cnt_all.reserve(std::max<size_t>(N * nc, 1)));
    HIP_TRY(x.topk_merge.reserve(std::max<size_t>(nc * stride, 1)));
    HIP_TRY(x.topk_final.reserve(std::max<size_t>(ne, 1)));
    HIP_TRY(x.topk_final_cnt.reserve(std::max<size_t>(nc, 1)));
    // (the device merge cuts ties at the k-th score by POSITION, which is document order only where every shard's list is
    // ordered: k up to kTopkSortLimit.  Beyond that the lists travel home as they are and the host sorts: merged = false)
    x.topk_merged = k <= kTopkSortLimit;
    // (pinned landing: a copy to pageable memory would wait for the stream -- for the collective -- inside the runtime)
    HIP_TRY(x.h_topk.reserve(std::max<size_t>((x.topk_merged ? 1 : N) * (ne * sizeof(uint2) + nc * 4), 16)));
    if (ne) NCCL_C(c, st, ncclAllGather(b->topk_out.p, x.topk_all.p, ne * sizeof(uint2), ncclUint8, c->comm, st));
    if (nc) NCCL_C(c, st, ncclAllGather(b->topk_cnt.p, x.topk_cnt_all.p, nc * 4, ncclUint8, c->comm, st));
    x.bytes_moved = (N - 1) * (ne * sizeof(uint2) + nc * 4);
    if (ne == 0) return COBS_GPU_OK;
    if (!x.topk_merged) {
        HIP_TRY(hipMemcpyAsync(x.h_topk.p, x.topk_all.p, N * ne * sizeof(uint2), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(x.h_topk.p + N * ne * sizeof(uint2), x.topk_cnt_all.p, N * nc * 4, hipMemcpyDeviceToHost, st));
        return COBS_GPU_OK;
    }
    HIP_TRY(launch_merge_lists(x.topk_all.p, x.topk_cnt_all.p, x.topk_merge.p, (uint32_t)N, nc, (uint32_t)k, (uint32_t)stride, st));
    for (size_t f = 0; f < np; ++f) {
        TopkArgs ta{};
        ta.counts = x.topk_merge.p + f * nq * stride;
        ta.from_pool = 1;
        ta.counts_stride = stride;
        ta.counts_offset = 0;
        ta.nslots = (uint32_t)(N * k);
        ta.thresholds = nullptr;                 // (every shard applied the threshold when it selected)
        ta.out = x.topk_final.p + f * nq * k;
        ta.out_count = x.topk_final_cnt.p + f * nq;
        ta.k = (uint32_t)k;
        ta.nq = (uint32_t)nq;
        ta.score_bytes = b->elem_bytes;
        ta.score_bits = (uint32_t)b->planes;
        ta.levels = ((uint32_t)b->planes + 11u) / 12u;
        ta.level_bits = ((uint32_t)b->planes + ta.levels - 1u) / ta.levels;
        ta.sort_limit = k <= kTopkSortLimit ? (uint32_t)k : 0u;
        ta.num_docs = (uint32_t)b->ix->parts[f].meta.doc_names.size();
        HIP_TRY(launch_topk(ta, st));
    }
    HIP_TRY(hipMemcpyAsync(x.h_topk.p, x.topk_final.p, ne * sizeof(uint2), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(x.h_topk.p + ne * sizeof(uint2), x.topk_final_cnt.p, nc * 4, hipMemcpyDeviceToHost, st));
    return COBS_GPU_OK;
}

cobs_gpu_status xchg_topk_collect(cobs_gpu_batch* b, cobs_gpu_comm* c, hipStream_t st, bool waited) {
    HIP_TRY(hipSetDevice(b->ix->device));
    if (!b->xchg) return fail(COBS_GPU_ERR_ARG, "no exchange of best-of lists is in flight");
    Exchange& x = *b->xchg;
    const size_t k = b->topk_k, nq = b->nq, np = b->ix->parts.size();
    const size_t ne = k * nq * np, nc = nq * np;
    if (!waited)
        if (cobs_gpu_status ws = sync_bounded(c, st, "the all-gather of the shards' best-of lists"); ws != COBS_GPU_OK) return ws;
    if (!x.topk_merged && ne) {
        // [file][query][rank * k]: the candidates of all ranks side by side, packed to the front; merged per query on the host
        const size_t N = (size_t)c->nranks;
        const uint2* all = reinterpret_cast<const uint2*>(x.h_topk.p);
        const uint32_t* cnt_all = reinterpret_cast<const uint32_t*>(x.h_topk.p + N * ne * sizeof(uint2));
        b->h_topk.assign(N * ne, make_uint2(0, 0));
        b->h_topk_cnt.assign(nc, 0);
        for (size_t f = 0; f < np; ++f)
            for (size_t q = 0; q < nq; ++q) {
                uint2* dst = b->h_topk.data() + (f * nq + q) * (N * k);
                uint32_t m = 0;
                for (size_t r = 0; r < N; ++r) {
                    const uint2* src = all + r * ne + (f * nq + q) * k;
                    const uint32_t cr = std::min<uint32_t>(cnt_all[r * nc + f * nq + q], (uint32_t)k);
                    for (uint32_t i = 0; i < cr; ++i) dst[m++] = src[i];
                }
                b->h_topk_cnt[f * nq + q] = m;
            }
        b->topk_stride = (uint32_t)(N * k);
        b->topk_fetched = true;
        return COBS_GPU_OK;
    }
    // [file][query][k]: the global k best of every (file, query), in result order where K3 orders them -- the layout a
    // one-GPU run leaves, so that cobs_gpu_batch_hits_host reads it the same way (no per-rank stride any more)
    const uint2* fin = reinterpret_cast<const uint2*>(x.h_topk.p);
    const uint32_t* cnt = reinterpret_cast<const uint32_t*>(x.h_topk.p + ne * sizeof(uint2));
    b->h_topk.assign(fin, fin + ne);
    b->h_topk_cnt.assign(cnt, cnt + nc);
    b->topk_stride = 0;
    b->topk_fetched = true;
    return COBS_GPU_OK;
}

}  // namespace cobs_amd
