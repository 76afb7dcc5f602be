// tests/mock_rccl/mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so for ONE process whose ranks are
// threads that share ONE GPU.
//
// RCCL refuses two ranks on one device, and the boxes this project is built on have one GPU: the native multi-GPU
// exchange of libcobs_gpu.so (cobs_amd/csrc/comm.cpp, multi.cpp) had therefore only ever run with ONE rank.  This
// file implements the dozen RCCL entry points comm.cpp calls with the semantics RCCL documents for them, so that the
// shipped library with this one preloaded (LD_PRELOAD=cobs_amd/libmockrccl.so, tests/mock_rccl/build.sh) runs the very
// code an N-GPU node runs -- the layout all-gather, plans, grouped send / receive pairs, all-reduces, pass loops,
// the ranks' agreement protocol, the device-list handle's worker threads -- with N > 1 ranks on one MI355X
// (tests/test_gpu_mock_ranks.py).  It moves bytes with hipMemcpy through host staging; it says nothing about xGMI.
//
// What it checks that hardware would answer with a hang or with silent corruption:
//   * every rank enters the same collective in the same order (kind, element size, count),
//   * every ncclSend has a ncclRecv of the same size posted by its peer in the same group, and vice versa,
//   * a rank that waits 120 s for peers that never arrive gets ncclSystemError instead of waiting for ever.
// A violation prints "[mock rccl] ..." on stderr and returns an error status to every rank involved.
//
// Fault injection (round 5; mock_rccl_inject below): the next ncclSend of a chosen rank returns an error, or posts one
// byte less than it was asked to -- what comm.cpp has to survive without leaving a group open or a peer waiting for
// ever (tests/test_gpu_mock_ranks.py, "fault" cases).  ncclCommAbort is local, as in RCCL: the aborting rank leaves,
// its peers find out by their own time limit.
//
// Semantics kept: operations are stream-ordered (the mock synchronises the stream it is given, exchanges, and
// returns; later work on that stream sees the data), in-place operation (sendbuff inside recvbuff) is allowed, a
// group's sends and receives complete together at ncclGroupEnd.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

enum Kind : int { kNone = 0, kAllGather, kAllReduce, kGroup };

struct P2P {
    int peer;
    size_t bytes;
    void* dst;                       // receive: where it goes
    std::vector<uint8_t> data;       // send: the bytes, staged on the host when the group is posted
};

struct Posted {
    int kind = kNone;
    size_t bytes = 0;                // all-gather: per rank; all-reduce: total
    int dtype = 0, op = 0;
    std::vector<uint8_t> data;       // this rank's contribution
    std::vector<P2P> sends, recvs;
};

struct Group {
    int nranks = 0;
    int joined = 0, left = 0;
    std::mutex mu;
    std::condition_variable cv;
    // barrier
    int waiting = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<Posted> posted;
};

std::mutex g_registry_mu;
std::map<std::string, Group*> g_registry;

// fault injection: armed by the test, consumed by the first matching ncclSend
std::mutex g_fault_mu;
int g_fault_kind = 0, g_fault_rank = -1, g_fault_skip = 0;      // kind 1: return an error, 2: post one byte less
int g_aborts = 0;

int take_fault(int rank) {
    std::lock_guard<std::mutex> lk(g_fault_mu);
    if (g_fault_kind == 0 || g_fault_rank != rank) return 0;
    if (g_fault_skip > 0) { --g_fault_skip; return 0; }
    const int k = g_fault_kind;
    g_fault_kind = 0;
    return k;
}

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 1;
    }
}

// -> false if the peers did not all arrive within the time limit (or an earlier barrier broke)
bool barrier(Group* g) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return false;
    const uint64_t gen = g->generation;
    if (++g->waiting == g->nranks) {
        g->waiting = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    static const int limit_s = getenv("MOCK_RCCL_TIMEOUT_S") ? atoi(getenv("MOCK_RCCL_TIMEOUT_S")) : 120;
    const bool ok = g->cv.wait_for(lk, std::chrono::seconds(limit_s), [&] { return g->generation != gen || g->broken; });
    if (!ok || g->broken) {
        if (!g->broken) std::fprintf(stderr, "[mock rccl] a rank waited for peers that never entered the collective (MOCK_RCCL_TIMEOUT_S, default 120 s)\n");
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}

}  // namespace

struct ncclComm {
    Group* g = nullptr;
    int rank = 0, nranks = 0;
};

namespace {

thread_local int t_group_depth = 0;
// the communicator this thread joined last: a group in which a rank posts nothing still has to meet its peers here
// (RCCL itself returns at once from such a group: point-to-point operations involve their two ends only; the mock
// meets all ranks per operation, and every rank of comm.cpp walks through the same groups in the same order)
thread_local ncclComm* t_last_comm = nullptr;
thread_local ncclComm* t_group_comm = nullptr;
thread_local hipStream_t t_group_stream = nullptr;
thread_local Posted t_group;

ncclResult_t fail(const char* what, int rank) {
    std::fprintf(stderr, "[mock rccl] rank %d: %s\n", rank, what);
    return ncclInvalidUsage;
}

// publish `mine`, meet the peers, let `take` read everybody's posting, meet again (postings may be reused after)
template <typename Take>
ncclResult_t exchange(ncclComm* c, Posted&& mine, Take take) {
    Group* g = c->g;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->posted[c->rank] = std::move(mine);
    }
    if (!barrier(g)) return ncclSystemError;
    ncclResult_t r = ncclSuccess;
    const Posted& me = g->posted[c->rank];
    for (int j = 0; j < c->nranks; ++j) {
        const Posted& p = g->posted[j];
        if (p.kind != me.kind || (me.kind != kGroup && (p.bytes != me.bytes || p.dtype != me.dtype || p.op != me.op))) {
            std::fprintf(stderr, "[mock rccl] rank %d is in collective %d (%zu bytes), rank %d in %d (%zu bytes): on hardware this hangs\n",
                         c->rank, me.kind, me.bytes, j, p.kind, p.bytes);
            r = ncclInvalidUsage;
        }
    }
    if (r == ncclSuccess) r = take(g->posted);
    if (!barrier(g)) return ncclSystemError;
    return r;
}

}  // namespace

// ---- virtual device ordinals (round 6) ----
// With MOCK_RCCL_VIRTUAL_DEVICES=N in the environment the process sees N HIP devices -- all of them the one GPU of
// the box: hipGetDeviceCount answers N, hipSetDevice(d) remembers d for the calling thread and selects device 0,
// hipGetDevice answers the thread's d, hipDeviceGetPCIBusId answers for device 0.  Preloaded, these definitions come
// before libamdhip64's for the SHIPPED libcobs_gpu.so, whose device-list handle (multi.cpp) then runs with ordinals
// 0 .. N-1 as it does on an N-GPU node: the range and duplicate checks of cobs_gpu_multi_open, a communicator and an index
// handle per ordinal, every hipSetDevice(ix->device) of the pass / exchange / ranking code with an ordinal other than 0,
// the NUMA look-up of the ranking's host threads (until now every rank of the stand-in was "device 0").
static int virtual_devices() {
    static const int n = [] { const char* e = getenv("MOCK_RCCL_VIRTUAL_DEVICES"); return e ? atoi(e) : 0; }();
    return n;
}
static thread_local int t_virtual_device = 0;
template <typename F>
static F real_hip(const char* name) {
    static_assert(sizeof(F) == sizeof(void*), "function pointer");
    void* p = dlsym(RTLD_NEXT, name);
    if (!p) { std::fprintf(stderr, "[mock rccl] no %s behind the stand-in\n", name); std::abort(); }
    F f;
    std::memcpy(&f, &p, sizeof f);
    return f;
}
extern "C" {

// ---- virtual device ordinals: the HIP entry points the stand-in answers itself (see above the extern "C" block) ----
hipError_t hipGetDeviceCount(int* count) {
    static auto real = real_hip<hipError_t (*)(int*)>("hipGetDeviceCount");
    const hipError_t e = real(count);
    if (e == hipSuccess && virtual_devices() > 0 && count && *count > 0) *count = virtual_devices();
    return e;
}
hipError_t hipSetDevice(int d) {
    static auto real = real_hip<hipError_t (*)(int)>("hipSetDevice");
    if (virtual_devices() <= 0) return real(d);
    if (d < 0 || d >= virtual_devices()) return hipErrorInvalidDevice;
    t_virtual_device = d;
    return real(0);
}
hipError_t hipGetDevice(int* d) {
    static auto real = real_hip<hipError_t (*)(int*)>("hipGetDevice");
    const hipError_t e = real(d);
    if (e == hipSuccess && virtual_devices() > 0 && d) *d = t_virtual_device;
    return e;
}
hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d) {
    static auto real = real_hip<hipError_t (*)(char*, int, int)>("hipDeviceGetPCIBusId");
    return real(buf, len, virtual_devices() > 0 ? 0 : d);
}
// ---- test controls (symbols only this stand-in defines) ----
// multi.cpp refuses a device that is listed twice unless this symbol exists in the process
int mock_rccl_ranks_may_share_a_device = 1;
// arm a fault for the ncclSend calls of `rank`: kind 1 = it returns ncclSystemError, kind 2 = it posts one byte less
// than asked (a size mismatch with the peer's receive); `skip` matching calls pass first.  kind 0 disarms.
void mock_rccl_inject(int kind, int rank, int skip) {
    std::lock_guard<std::mutex> lk(g_fault_mu);
    g_fault_kind = kind;
    g_fault_rank = rank;
    g_fault_skip = skip;
}
int mock_rccl_group_depth(void) { return t_group_depth; }      // open groups of the CALLING thread
int mock_rccl_aborts(void) { std::lock_guard<std::mutex> lk(g_fault_mu); return g_aborts; }

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidUsage: return "invalid usage (mock rccl: see stderr)";
        case ncclSystemError: return "unhandled system error (mock rccl: peers never arrived)";
        case ncclInvalidArgument: return "invalid argument";
        default: return "mock rccl error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::mutex mu;
    static std::mt19937_64 rng(0x6d6f636bull);
    std::lock_guard<std::mutex> lk(mu);
    std::memset(id, 0, sizeof *id);
    const uint64_t a = rng(), b = rng();
    std::memcpy(id->internal, &a, 8);
    std::memcpy(id->internal + 8, &b, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof id.internal);
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_registry_mu);
        Group*& slot = g_registry[key];
        if (!slot) {
            slot = new Group;
            slot->nranks = nranks;
            slot->posted.resize((size_t)nranks);
        }
        g = slot;
        if (g->nranks != nranks) return fail("ncclCommInitRank with another rank count for the same id", rank);
        ++g->joined;
    }
    ncclComm* c = new ncclComm;
    c->g = g;
    c->rank = rank;
    c->nranks = nranks;
    if (!barrier(g)) { delete c; return ncclSystemError; }      // returns when every rank has called it
    t_last_comm = c;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (t_last_comm == c) t_last_comm = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_registry_mu);
        if (++c->g->left == c->g->nranks) {
            for (auto it = g_registry.begin(); it != g_registry.end(); ++it)
                if (it->second == c->g) { g_registry.erase(it); break; }
            delete c->g;
        }
    }
    delete c;
    return ncclSuccess;
}

// local, as in RCCL: this rank is gone, the peers notice when they wait for it
ncclResult_t ncclCommAbort(ncclComm_t c) {
    { std::lock_guard<std::mutex> lk(g_fault_mu); ++g_aborts; }
    return ncclCommDestroy(c);
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t* e) {
    if (e) *e = ncclSuccess;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t st) {
    if (t_group_depth) return fail("a collective inside ncclGroupStart / End is not something comm.cpp does: not mocked", c->rank);
    const size_t bytes = count * dtype_size(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    Posted p;
    p.kind = kAllGather;
    p.bytes = bytes;
    p.dtype = (int)dt;
    p.data.resize(bytes);
    if (bytes && hipMemcpy(p.data.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    return exchange(c, std::move(p), [&](const std::vector<Posted>& all) -> ncclResult_t {
        for (int j = 0; j < c->nranks && bytes; ++j)
            if (hipMemcpy(static_cast<uint8_t*>(recvbuff) + (size_t)j * bytes, all[j].data.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        return ncclSuccess;
    });
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t st) {
    if (t_group_depth) return fail("a collective inside ncclGroupStart / End is not something comm.cpp does: not mocked", c->rank);
    if (!((dt == ncclUint8 || dt == ncclUint32) && (op == ncclSum || op == ncclMax)))
        return fail("all-reduce of this type / operation is not something comm.cpp does: not mocked", c->rank);
    const size_t bytes = count * dtype_size(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    Posted p;
    p.kind = kAllReduce;
    p.bytes = bytes;
    p.dtype = (int)dt;
    p.op = (int)op;
    p.data.resize(bytes);
    if (bytes && hipMemcpy(p.data.data(), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    return exchange(c, std::move(p), [&](const std::vector<Posted>& all) -> ncclResult_t {
        std::vector<uint8_t> out(all[0].data);
        for (int j = 1; j < c->nranks; ++j) {
            const std::vector<uint8_t>& in = all[j].data;
            if (dt == ncclUint8) {
                for (size_t i = 0; i < bytes; ++i) out[i] = op == ncclSum ? (uint8_t)(out[i] + in[i]) : std::max(out[i], in[i]);
            } else {
                for (size_t i = 0; i < count; ++i) {
                    uint32_t a, b;
                    std::memcpy(&a, out.data() + 4 * i, 4);
                    std::memcpy(&b, in.data() + 4 * i, 4);
                    a = op == ncclSum ? a + b : std::max(a, b);
                    std::memcpy(out.data() + 4 * i, &a, 4);
                }
            }
        }
        if (bytes && hipMemcpy(recvbuff, out.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        return ncclSuccess;
    });
}

ncclResult_t ncclGroupStart() {
    if (t_group_depth++ == 0) {
        t_group = Posted();
        t_group.kind = kGroup;
        t_group_comm = nullptr;
        t_group_stream = nullptr;
    }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    if (!t_group_depth) return fail("ncclSend outside a group: comm.cpp always groups its pairs; not mocked", c->rank);
    if (peer < 0 || peer >= c->nranks) return fail("ncclSend to a rank outside the communicator", c->rank);
    if (t_group_comm && t_group_comm != c) return fail("one group over two communicators: not mocked", c->rank);
    t_group_comm = c;
    t_group_stream = st;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    const int fault = take_fault(c->rank);
    if (fault == 1) {
        std::fprintf(stderr, "[mock rccl fault] rank %d: injected failure of ncclSend to rank %d\n", c->rank, peer);
        return ncclSystemError;
    }
    P2P s;
    s.peer = peer;
    s.bytes = count * dtype_size(dt);
    if (fault == 2 && s.bytes > 0) {
        std::fprintf(stderr, "[mock rccl fault] rank %d: injected short ncclSend to rank %d (%zu bytes instead of %zu)\n", c->rank, peer,
                     s.bytes - 1, s.bytes);
        s.bytes -= 1;
    }
    s.dst = nullptr;
    s.data.resize(s.bytes);
    if (s.bytes && hipMemcpy(s.data.data(), sendbuff, s.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    t_group.sends.push_back(std::move(s));
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
    if (!t_group_depth) return fail("ncclRecv outside a group: comm.cpp always groups its pairs; not mocked", c->rank);
    if (peer < 0 || peer >= c->nranks) return fail("ncclRecv from a rank outside the communicator", c->rank);
    if (t_group_comm && t_group_comm != c) return fail("one group over two communicators: not mocked", c->rank);
    t_group_comm = c;
    t_group_stream = st;
    P2P r;
    r.peer = peer;
    r.bytes = count * dtype_size(dt);
    r.dst = recvbuff;
    t_group.recvs.push_back(std::move(r));
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth != 0) return ncclSuccess;
    ncclComm* c = t_group_comm ? t_group_comm : t_last_comm;
    if (!c) return ncclSuccess;
    if (t_group_comm && hipStreamSynchronize(t_group_stream) != hipSuccess) return ncclUnhandledCudaError;
    return exchange(c, std::move(t_group), [&](const std::vector<Posted>& all) -> ncclResult_t {
        ncclResult_t r = ncclSuccess;
        const Posted& me = all[c->rank];
        // my receives: the k-th receive from peer j takes peer j's k-th send to me
        std::vector<size_t> next((size_t)c->nranks, 0);
        for (const P2P& rv : me.recvs) {
            const Posted& pj = all[rv.peer];
            const P2P* match = nullptr;
            size_t seen = 0;
            for (const P2P& s : pj.sends)
                if (s.peer == c->rank && seen++ == next[rv.peer]) { match = &s; break; }
            ++next[rv.peer];
            if (!match) {
                std::fprintf(stderr, "[mock rccl] rank %d waits for %zu bytes from rank %d, which sends nothing (more) to it: on hardware this hangs\n",
                             c->rank, rv.bytes, rv.peer);
                r = ncclInvalidUsage;
                continue;
            }
            if (match->bytes != rv.bytes) {
                std::fprintf(stderr, "[mock rccl] rank %d expects %zu bytes from rank %d, which sends %zu\n", c->rank, rv.bytes, rv.peer, match->bytes);
                r = ncclInvalidUsage;
                continue;
            }
            if (rv.bytes && hipMemcpy(rv.dst, match->data.data(), rv.bytes, hipMemcpyHostToDevice) != hipSuccess) r = ncclUnhandledCudaError;
        }
        // my sends: the peer must have posted as many receives from me as I send to it
        for (int j = 0; j < c->nranks; ++j) {
            size_t ns = 0, nr = 0;
            for (const P2P& s : me.sends) ns += s.peer == j;
            for (const P2P& rv : all[j].recvs) nr += rv.peer == c->rank;
            if (ns != nr) {
                std::fprintf(stderr, "[mock rccl] rank %d sends %zu message(s) to rank %d, which posted %zu receive(s) from it: on hardware this hangs\n",
                             c->rank, ns, j, nr);
                r = ncclInvalidUsage;
                continue;
            }
            // ... of the same sizes, in order (the receiver reports the mismatch too)
            size_t k = 0;
            for (const P2P& s : me.sends) {
                if (s.peer != j) continue;
                size_t seen = 0;
                for (const P2P& rv : all[j].recvs)
                    if (rv.peer == c->rank && seen++ == k) {
                        if (rv.bytes != s.bytes) r = ncclInvalidUsage;
                        break;
                    }
                ++k;
            }
        }
        return r;
    });
}

}  // extern "C"
