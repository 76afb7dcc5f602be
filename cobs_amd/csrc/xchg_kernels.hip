// cobs_amd/csrc/xchg_kernels.hip -- device side of the owner-routed hit exchange (comm.cpp): the hit pool of a
// thresholded pass is an unordered list of (query, file, document, score) records; before it leaves the GPU it
// is bucketed by the rank that OWNS each record's query (rank j owns the queries [nq*j/N, nq*(j+1)/N), as in the
// all-to-all exchange of count rows), so that every record crosses xGMI once, to one GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "device_types.hpp"
#include "kernels.hpp"

namespace cobs_amd {

namespace {

__device__ __forceinline__ uint32_t owner_of(uint32_t q, uint32_t nq, uint32_t nranks) {
    uint32_t j = (uint32_t)(((uint64_t)q * nranks) / nq);
    while (j + 1u < nranks && (uint64_t)nq * (j + 1u) / nranks <= q) ++j;      // q_begin(j + 1) <= q
    return j;
}

// COUNT: per-owner record counts (LDS histogram per block, then one atomic per owner and block);
// else: scatter every record behind its owner's cursor
template <bool COUNT>
__global__ __launch_bounds__(256) void bucket_hits_kernel(BucketArgs a) {
    __shared__ unsigned int hist[64];
    if (COUNT) {
        if (threadIdx.x < 64u) hist[threadIdx.x] = 0u;
        __syncthreads();
    }
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < a.n; i += stride) {
        const HitDev h = a.hits[i];
        const uint32_t o = owner_of(h.query < a.nq ? h.query : a.nq - 1u, a.nq, a.nranks);
        if (COUNT) atomicAdd(&hist[o], 1u);
        else a.out[atomicAdd(&a.cursor[o], 1ull)] = h;
    }
    if (COUNT) {
        __syncthreads();
        if (threadIdx.x < a.nranks && hist[threadIdx.x]) atomicAdd(&a.cursor[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
    }
}

// ---- the hit pool in result order (PoolArgs) ----
__global__ __launch_bounds__(256) void pool_count_kernel(PoolArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;          // (64-bit: i + stride must not wrap for n near 2^32)
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < a.n; i += stride) {
        const uint32_t q = a.in[i].query;
        if (q < a.nq) atomicAdd(&a.cnt[q], 1u);
    }
}

// exclusive scan of cnt[0 .. nq] by ONE work-group of 1024 threads (a pass holds up to a few 100 000 queries)
__global__ __launch_bounds__(1024) void pool_scan_kernel(PoolArgs a) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, total = a.nq + 1u;
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t i0 = t * per, i1 = i0 + per < total ? i0 + per : total;
    uint32_t s = 0;
    for (uint32_t i = i0; i < i1; ++i) s += a.cnt[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {          // Hillis-Steele over the 1024 partial sums
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0u;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t c = a.cnt[i];
        a.off[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void pool_scatter_kernel(PoolArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < a.n; i += stride) {
        const HitDev h = a.in[i];
        if (h.query < a.nq) a.tmp[a.off[h.query] + atomicAdd(&a.cur[h.query], 1u)] = h;
    }
}

__device__ __forceinline__ bool pool_before(const HitDev& x, const HitDev& y, bool single) {
    if (!single && x.score != y.score) return x.score > y.score;
    if (x.part != y.part) return x.part < y.part;
    return x.doc < y.doc;
}

// one wave per query: its bucket in LDS, every record placed at its RANK (the number of records that precede it:
// (file, document) is unique inside a query, so the order is total and every rank is taken once)
constexpr uint32_t kPoolSeg = 1024;
__global__ __launch_bounds__(64) void pool_sort_kernel(PoolArgs a) {
    __shared__ HitDev seg[kPoolSeg];
    const uint32_t q = blockIdx.x;
    const uint32_t b0 = a.off[q], n = a.off[q + 1] - b0;
    if (n == 0) return;
    const uint32_t lane = threadIdx.x;
    if (n > a.seg_max || n > kPoolSeg) {           // (left in pool order: the host orders this one)
        for (uint32_t i = lane; i < n; i += 64u) a.out[b0 + i] = a.tmp[b0 + i];
        return;
    }
    for (uint32_t i = lane; i < n; i += 64u) seg[i] = a.tmp[b0 + i];
    __syncthreads();
    const bool single = a.single && a.single[q];
    for (uint32_t i = lane; i < n; i += 64u) {
        const HitDev me = seg[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += pool_before(seg[j], me, single) ? 1u : 0u;
        a.out[b0 + rank] = me;
    }
}

// The shards' best-of lists of a limited search, all-gathered as [rank][file][query][k] with their counts, laid side by
// side per (file, query) for K3's pool mode: [file][query][stride] with rank r's entries at [r * k, r * k + count) and
// everything else marked unused (document 0xFFFFFFFF).  A rank holds a contiguous range of a file's documents and its
// list is in (score desc, document asc) order, so equal scores stay in document order by position -- what the merge
// needs to cut ties at the k-th score as counts_to_result does (classic_search.cpp:136-145).
__global__ __launch_bounds__(256) void merge_lists_kernel(const uint2* all, const uint32_t* cnt, uint2* out, uint32_t nranks,
                                                          uint64_t lists, uint32_t k, uint32_t stride) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= lists * stride) return;
    const uint64_t fq = idx / stride;
    const uint32_t j = (uint32_t)(idx - fq * stride), r = j / k, i = j - r * k;
    uint2 v = make_uint2(0xFFFFFFFFu, 0u);
    if (r < nranks && i < min(cnt[(uint64_t)r * lists + fq], k)) v = all[((uint64_t)r * lists + fq) * k + i];
    out[idx] = v;
}

// What the ranks of a sharded search agree on after the scan of a pass, as ONE record per rank that an all-gather
// carries (sharded.cpp): the rank's host-side status, K1's first invalid query, the fill of its hit pool -- read from
// the batch's flag words where the scan left them, no host round trip in between.
__global__ void pass_meta_kernel(const uint32_t* flags, uint64_t* rec, uint64_t status, uint64_t extra) {
    if (threadIdx.x != 0u) return;
    rec[0] = status;
    rec[1] = flags ? flags[0] : 0u;
    rec[2] = flags ? ((uint64_t)flags[3] << 32 | flags[2]) : 0ull;
    rec[3] = extra;
}

}  // namespace

hipError_t launch_merge_lists(const uint2* all, const uint32_t* cnt, uint2* out, uint32_t nranks, uint64_t lists, uint32_t k,
                              uint32_t stride, hipStream_t stream) {
    if (lists == 0 || k == 0) return hipSuccess;
    const uint64_t blocks = (lists * stride + 255u) / 256u;
    if (blocks > 0x7FFFFFFFull || stride < (uint64_t)nranks * k) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_lists_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, all, cnt, out, nranks, lists, k, stride);
    return hipGetLastError();
}

hipError_t launch_pass_meta(const uint32_t* flags, uint64_t* rec, uint64_t status, uint64_t extra, hipStream_t stream) {
    hipLaunchKernelGGL(pass_meta_kernel, dim3(1), dim3(64), 0, stream, flags, rec, status, extra);
    return hipGetLastError();
}

hipError_t launch_order_pool(const PoolArgs& a, hipStream_t stream) {
    if (a.nq == 0) return hipSuccess;
    if (a.n) {
        const uint32_t blocks = std::min<uint32_t>((a.n + 255u) / 256u, 4096u);
        hipLaunchKernelGGL(pool_count_kernel, dim3(blocks), dim3(256), 0, stream, a);
    }
    hipLaunchKernelGGL(pool_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
    if (a.n) {
        const uint32_t blocks = std::min<uint32_t>((a.n + 255u) / 256u, 4096u);
        hipLaunchKernelGGL(pool_scatter_kernel, dim3(blocks), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(pool_sort_kernel, dim3(a.nq), dim3(64), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_bucket_hits(const BucketArgs& a, bool count, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    if (a.nranks == 0 || a.nranks > 64 || a.nq == 0) return hipErrorInvalidValue;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((a.n + 255) / 256, 4096);
    if (count) hipLaunchKernelGGL(bucket_hits_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(bucket_hits_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cobs_amd
