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

}  // namespace

hipError_t launch_bucket_hits(const BucketArgs& a, bool count, hipStream_t stream) {
    if (a.n == 0) return hipSuccess;
    if (a.nranks == 0 || a.nranks > 64 || a.nq == 0) return hipErrorInvalidValue;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((a.n + 255) / 256, 4096);
    if (count) hipLaunchKernelGGL(bucket_hits_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(bucket_hits_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cobs_amd
