// cobs_amd/csrc/rank_kernels.hip -- ranking of EVERY document on the device (gfx950, wave64).
//
// The reference's default call -- search(query, result) with threshold 0 and no limit, which is also
// what its own benchmark times (src/cobs.cpp:618-626) -- ends in counts_to_result
// (cobs/query/classic_search.cpp:109-202): keep the documents with score >= threshold, order them by
// (score descending, then (file, document) ascending) with std::partial_sort, emit the first
// num_results.  That order is a STABLE sort by score of the documents taken in (file, document)
// order, so it is computed here as a least-significant-digit radix sort whose every pass is a stable
// counting sort inside one work-group:
//
//   rank_kernel   one work-group (4 waves) per query.  A pass looks at `bits` (<= 12) bits of the
//                 score: (1) per-wave digit histograms in LDS, (2) an exclusive scan over (digit
//                 descending, wave ascending) turns them into the first output position of every
//                 (wave, digit), (3) every wave walks ITS contiguous range of the elements in order, 64 at
//                 a time: lanes with the same digit find each other with `bits` ballots
//                 (match-any), a lane's position is base[wave][digit] + its rank among them, the
//                 first lane of each digit advances the base.
//                 Scores of up to 12 bits (queries of up to 4095 terms) need ONE pass straight from
//                 the score rows to the result records (slot, score: 4 bytes packed where both fit 32 bits, else 8);
//                 wider scores take 2 (up to 24 bits) or 3
//                 passes through (score, slot) pairs in HBM.
//
// The elements of the first pass are the local score slots of the row (the files' slices back to
// back, each a multiple of 8 slots: 16-byte loads of 8 scores per lane, 512 slots per wave-load,
// staged through LDS so that lane i then ranks slot 64 s + i of the block); slots of padding
// documents and documents below their file's threshold take no part.  A query with a single hash in
// total is not ordered by score at all (max_counts <= 1, classic_search.cpp:136,179): every digit
// reads as 0 and the stable sort leaves index order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/cobs_gpu.h"
#include "device_types.hpp"
#include "kernels.hpp"

namespace cobs_amd {

namespace {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;

// LDS written by some lanes of a wave is read by others: order the compiler (the hardware's LDS queue is in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t lanes_below(uint32_t lane, uint32_t lo, uint32_t hi) {
    // popcount of the 64-bit mask (hi:lo) restricted to lanes < lane
    const uint32_t mlo = lane >= 32u ? 0xFFFFFFFFu : ((1u << lane) - 1u);
    const uint32_t mhi = lane > 32u ? ((1u << (lane - 32u)) - 1u) : 0u;
    return (uint32_t)__popc(lo & mlo) + (uint32_t)__popc(hi & mhi);
}

template <typename ST>
__device__ __forceinline__ void load8(const ST* row, uint32_t i, uint32_t (&s)[8]) {
    if constexpr (sizeof(ST) == 1) {
        const uint2 v = *reinterpret_cast<const uint2*>(row + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ((j < 4 ? v.x : v.y) >> ((j & 3) * 8)) & 0xFFu;
    } else if constexpr (sizeof(ST) == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
    } else {
        const uint4 a = *reinterpret_cast<const uint4*>(row + i);
        const uint4 b = *reinterpret_cast<const uint4*>(row + i + 4);
        s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
        s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    }
}

// the file a local slot belongs to (slices are back to back in slot order)
__device__ __forceinline__ uint32_t part_of(const RankArgs& a, uint32_t slot) {
    uint32_t p = 0;
    while (p + 1u < a.nparts && slot >= a.parts[p + 1].slot0) ++p;
    return p;
}

// FIRST: elements are the slots of the score row; else (score, slot) pairs of the previous pass.
// LAST: results leave as (slot, score) records in rank order; else as (score, slot) pairs for the next pass.
// SEG (single-pass sorts of long rows; a work-group alone on its CU is latency-bound -- 0.36 ms for 100 000 documents,
// the head of every default call): the row is cut into gridDim.x segments, a work-group per (segment, query).
// SEG = 1 leaves the per-wave digit histograms of its segment in a.seg_hist [query][4 * segment + wave][2^bits];
// rank_prefix_kernel turns them into first output positions in place; SEG = 2 starts from those and scatters its segment.
// SEG = 0: one work-group does it all (every pass of the multi-pass sorts, short rows).
template <typename ST, bool FIRST, bool LAST, int SEG>
__global__ __launch_bounds__(256) void rank_kernel(RankArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t NB = 1u << a.bits;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);          // [4 waves][NB], reversed digits: bin = NB-1 - digit
    uint32_t* partial = hist + 4u * NB;                          // [256]
    uint32_t* stage = partial + 256;                             // [4 waves][512] keys of one block (FIRST)
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t qb = SEG ? blockIdx.y : blockIdx.x;           // query inside the window
    const uint32_t nranges = SEG ? gridDim.x * 4u : 4u;          // contiguous element ranges, one per wave
    const uint32_t range0 = SEG ? blockIdx.x * 4u : 0u;          // ... the first of this work-group
    const uint32_t q = a.q0 + qb;                                // query inside the batch
    const bool by_score = a.by_score[q] != 0;
    const uint32_t dmask = NB - 1u;
    const ST* row = reinterpret_cast<const ST*>(a.rows) + (uint64_t)(q - a.row_q0) * a.row_stride;
    const uint2* src = a.src + (uint64_t)qb * a.pair_stride;
    const uint32_t n = FIRST ? a.nslots : a.npass[qb];
    (void)row; (void)src;
    // wave w owns the contiguous element range [w0, w1), a multiple of 512 long
    const uint32_t per = ((n + nranges - 1u) / nranges + 511u) / 512u * 512u;
    const uint32_t w0 = (uint32_t)min((uint64_t)(range0 + wave) * per, (uint64_t)n), w1 = min(w0 + per, n);
    uint32_t* myh = hist + wave * NB;
    uint32_t* mystage = stage + wave * 512u;

    // score of slot `slot` if it takes part (a real document at or above its file's threshold), else kInvalid;
    // the 8 slots of one lane lie in one file
    // (the raw scores are loaded one block ahead of their use: a work-group is alone on its CU, nothing else hides
    // the latency of these dependent loads)
    auto keys8 = [&](uint32_t i, const uint32_t (&s)[8], uint32_t (&key)[8]) {
        const RankPart pt = a.parts[part_of(a, i)];
        const uint32_t thr = pt.thr ? pt.thr[q] : 0u;
        const uint32_t d = i - pt.slot0;                         // document index inside the held slice
#pragma unroll
        for (int j = 0; j < 8; ++j) key[j] = (d + j < pt.ndocs && s[j] >= thr) ? s[j] : kInvalid;
    };

    uint32_t* seg_rows = nullptr;                                // this work-group's four rows of a.seg_hist
    if constexpr (SEG != 0) seg_rows = a.seg_hist + ((uint64_t)qb * nranges + range0) * NB;
    if constexpr (SEG == 2) {
        for (uint32_t i = tid; i < 4u * NB; i += 256) hist[i] = seg_rows[i];
    } else {
        for (uint32_t i = tid; i < 4u * NB; i += 256) hist[i] = 0;
    }
    __syncthreads();
    // ---- (1) histograms
    if constexpr (SEG == 2) {
    } else if constexpr (FIRST) {
        uint32_t cur[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nxt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (w0 + lane * 8u < w1) load8<ST>(row, w0 + lane * 8u, cur);
        for (uint32_t i0 = w0; i0 < w1; i0 += 512) {
            const uint32_t i = i0 + lane * 8u;
            if (i + 512u < w1) load8<ST>(row, i + 512u, nxt);
            if (i < w1) {
                uint32_t key[8];
                keys8(i, cur, key);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (key[j] != kInvalid)
                        atomicAdd(&myh[dmask - (by_score ? (key[j] >> a.shift) & dmask : 0u)], 1u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
        }
    } else {
        for (uint32_t i = w0 + lane; i < w1; i += 64)
            atomicAdd(&myh[dmask - (by_score ? (src[i].x >> a.shift) & dmask : 0u)], 1u);
    }
    __syncthreads();
    if constexpr (SEG == 1) {
        for (uint32_t i = tid; i < 4u * NB; i += 256) seg_rows[i] = hist[i];
        return;
    }
    // ---- (2) first output position of every (wave, bin): bins ascending (= digits descending), waves ascending
    if constexpr (SEG == 0) {
        const uint32_t seg = (NB + 255u) / 256u;
        const uint32_t b0 = tid * seg, b1 = min(b0 + seg, NB);
        uint32_t sum = 0;
        for (uint32_t b = b0; b < b1; ++b) sum += hist[b] + hist[NB + b] + hist[2 * NB + b] + hist[3 * NB + b];
        partial[tid] = sum;
        __syncthreads();
        if (wave == 0u) {           // exclusive scan of the 256 segment sums: 4 per lane + a wave scan
            const uint32_t p0 = partial[4 * lane], p1 = partial[4 * lane + 1], p2 = partial[4 * lane + 2], p3 = partial[4 * lane + 3];
            uint32_t incl = p0 + p1 + p2 + p3;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= (uint32_t)off) incl += t;
            }
            const uint32_t ex = incl - (p0 + p1 + p2 + p3);
            partial[4 * lane] = ex;
            partial[4 * lane + 1] = ex + p0;
            partial[4 * lane + 2] = ex + p0 + p1;
            partial[4 * lane + 3] = ex + p0 + p1 + p2;
            if (lane == 63u) {
                if constexpr (FIRST && !LAST) a.npass[qb] = incl;
                if constexpr (LAST) a.out_count[qb] = min(incl, a.limit);
            }
        }
        __syncthreads();
        uint32_t run = partial[tid];
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t before = run;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t c = hist[w * NB + b];
                hist[w * NB + b] = run;
                run += c;
            }
            if constexpr (FIRST && LAST)
                if (a.bin_count) a.bin_count[(uint64_t)qb * NB + b] = run - before;
        }
        __syncthreads();
    }
    // ---- (3) stable scatter: every wave walks its range in order
    uint2* dst = a.dst + (uint64_t)qb * a.pair_stride;
    uint2* out = reinterpret_cast<uint2*>(a.out) + (uint64_t)qb * a.out_stride;
    uint32_t* out32 = reinterpret_cast<uint32_t*>(a.out) + (uint64_t)qb * a.out_stride;
    (void)out; (void)out32;
    auto place = [&](bool valid, uint32_t score, uint32_t slot) {
        const uint32_t bin = dmask - (by_score ? (score >> a.shift) & dmask : 0u);
        // lanes of this step with the same bin
        const unsigned long long vm = __ballot(valid);
        uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
        for (uint32_t b = 0; b < a.bits; ++b) {
            const bool bit = ((bin >> b) & 1u) != 0u;
            const unsigned long long m = __ballot(valid && bit);
            plo &= bit ? (uint32_t)m : ~(uint32_t)m;
            phi &= bit ? (uint32_t)(m >> 32) : ~(uint32_t)(m >> 32);
        }
        const uint32_t before = lanes_below(lane, plo, phi);
        uint32_t pos = 0;
        if (valid) pos = myh[bin] + before;
        wave_sync();                              // every lane has read the base before the leaders advance it
        if (valid && before == 0u) myh[bin] = pos + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
        wave_sync();
        if (!valid) return;
        if constexpr (LAST) {
            // (slot of the ranked row, score) -- what crosses PCIe per result; the host turns the slot into
            // (file, document) while it copies the window into the caller's cobs_gpu_hit array (rank.cpp).
            // Where slot and score fit 32 bits together (C3: 17 + 10) the record is ONE word: the default
            // call is bound by these bytes, not by the ordering.
            if (pos < a.limit) {
                if (a.pack_bits) out32[pos] = score << a.pack_bits | slot;
                else out[pos] = make_uint2(slot, score);
            }
        } else {
            dst[pos] = make_uint2(score, slot);
        }
    };
    if constexpr (FIRST) {
        uint32_t cur[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nxt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (w0 + lane * 8u < w1) load8<ST>(row, w0 + lane * 8u, cur);
        for (uint32_t i0 = w0; i0 < w1; i0 += 512) {
            const uint32_t i = i0 + lane * 8u;
            if (i + 512u < w1) load8<ST>(row, i + 512u, nxt);
            uint32_t key[8];
            if (i < w1) keys8(i, cur, key);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) key[j] = kInvalid;
            }
            wave_sync();                          // the previous block's keys have been read
            *reinterpret_cast<uint4*>(mystage + lane * 8u) = make_uint4(key[0], key[1], key[2], key[3]);
            *reinterpret_cast<uint4*>(mystage + lane * 8u + 4u) = make_uint4(key[4], key[5], key[6], key[7]);
            wave_sync();
            const uint32_t steps = min(8u, (w1 - i0 + 63u) / 64u);
            for (uint32_t s = 0; s < steps; ++s) {
                const uint32_t k = mystage[s * 64u + lane];
                place(k != kInvalid, k, i0 + s * 64u + lane);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
        }
    } else {
        for (uint32_t i0 = w0; i0 < w1; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool valid = i < w1;
            const uint2 e = valid ? src[i] : make_uint2(0u, 0u);
            place(valid, e.x, e.y);
        }
    }
}

// Between the two halves of a segmented single-pass sort: the histograms of the nranges = 4 * segments element ranges of
// a query, a.seg_hist [query][range][2^bits], become the first output position of every (range, bin) -- bins ascending
// (= digits descending), ranges ascending inside a bin: rank_kernel's step (2) over all the segments' waves.  One
// work-group per query; also the query's result count and, for the slim records, its records per bin.
__global__ __launch_bounds__(256) void rank_prefix_kernel(RankArgs a, uint32_t nranges) {
    __shared__ uint32_t partial[256];
    const uint32_t NB = 1u << a.bits;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t qb = blockIdx.x;
    uint32_t* g = a.seg_hist + (uint64_t)qb * nranges * NB;
    const uint32_t seg = (NB + 255u) / 256u;
    const uint32_t b0 = min(tid * seg, NB), b1 = min(b0 + seg, NB);
    uint32_t sum = 0;
    for (uint32_t r = 0; r < nranges; ++r)
        for (uint32_t b = b0; b < b1; ++b) sum += g[(uint64_t)r * NB + b];
    partial[tid] = sum;
    __syncthreads();
    if (wave == 0u) {
        const uint32_t p0 = partial[4 * lane], p1 = partial[4 * lane + 1], p2 = partial[4 * lane + 2], p3 = partial[4 * lane + 3];
        uint32_t incl = p0 + p1 + p2 + p3;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += t;
        }
        const uint32_t ex = incl - (p0 + p1 + p2 + p3);
        partial[4 * lane] = ex;
        partial[4 * lane + 1] = ex + p0;
        partial[4 * lane + 2] = ex + p0 + p1;
        partial[4 * lane + 3] = ex + p0 + p1 + p2;
        if (lane == 63u) a.out_count[qb] = min(incl, a.limit);
    }
    __syncthreads();
    uint32_t run = partial[tid];
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t before = run;
        for (uint32_t r = 0; r < nranges; ++r) {
            const uint32_t c = g[(uint64_t)r * NB + b];
            g[(uint64_t)r * NB + b] = run;
            run += c;
        }
        if (a.bin_count) a.bin_count[(uint64_t)qb * NB + b] = run - before;
    }
}

template <typename ST>
hipError_t launch_rank_t(const RankArgs& a, bool first, bool last, size_t lds, hipStream_t stream) {
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(a.nq), dim3(256), lds, stream, a);
        return hipGetLastError();
    };
    if (first && last && a.nseg > 1 && a.seg_hist && a.nq <= 65535u) {
        auto seg = [&](auto kern) -> hipError_t {
            if (lds > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(kern, dim3(a.nseg, a.nq), dim3(256), lds, stream, a);
            return hipGetLastError();
        };
        hipError_t e = seg(rank_kernel<ST, true, true, 1>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(rank_prefix_kernel, dim3(a.nq), dim3(256), 0, stream, a, a.nseg * 4u);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        return seg(rank_kernel<ST, true, true, 2>);
    }
    if (first && last) return go(rank_kernel<ST, true, true, 0>);
    if (first) return go(rank_kernel<ST, true, false, 0>);
    if (last) return go(rank_kernel<ST, false, true, 0>);
    return go(rank_kernel<ST, false, false, 0>);
}

// one thread per output dword: the records whose bits it holds (at most 32 / slot_bits + 2 of them, cached reads)
__global__ __launch_bounds__(256) void pack_slots_kernel(SlotPackArgs a) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= a.words) return;
    const uint32_t s = a.slot_bits, mask = (1u << s) - 1u;
    const uint64_t bit0 = (uint64_t)w * 32u;
    const uint32_t r0 = (uint32_t)(bit0 / s);
    const uint32_t o = (uint32_t)(bit0 - (uint64_t)r0 * s);         // bits of record r0 that lie in earlier dwords
    for (uint32_t q = blockIdx.y; q < a.nq; q += gridDim.y) {
        const uint32_t* in = a.in + (uint64_t)q * a.in_stride;
        uint32_t r = r0;
        uint32_t val = r < a.n ? (in[r] & mask) >> o : 0u;
        uint32_t filled = s - o;
        for (++r; filled < 32u; ++r, filled += s)
            if (r < a.n) val |= (in[r] & mask) << filled;
        a.out[(uint64_t)q * a.words + w] = val;
    }
}

// one work-group per (query, range): an LDS histogram of up to 4096 bins, flushed with one 64-bit atomic per
// non-empty bin; wider score ranges go to global atomics directly (queries of more than 4095 terms)
constexpr uint32_t kHistLds = 4096;
template <typename ST>
__global__ __launch_bounds__(256) void score_hist_kernel(HistArgs a) {
    __shared__ uint32_t h[kHistLds];
    const uint32_t q = blockIdx.x, r = blockIdx.y;
    const bool lds = a.nbins <= kHistLds;
    if (lds) {
        for (uint32_t i = threadIdx.x; i < a.nbins; i += 256u) h[i] = 0u;
        __syncthreads();
    }
    const ST* row = static_cast<const ST*>(a.rows) + (uint64_t)q * a.row_stride;
    const HistRange rg = a.ranges[r];
    for (uint32_t i = rg.begin + threadIdx.x; i < rg.end; i += 256u) {
        uint32_t s = (uint32_t)row[i];
        if (s >= a.nbins) s = a.nbins - 1u;
        if (lds) atomicAdd(&h[s], 1u);
        else atomicAdd(&a.hist[s], 1ull);
    }
    if (lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < a.nbins; i += 256u)
            if (h[i]) atomicAdd(&a.hist[i], (unsigned long long)h[i]);
    }
}

}  // namespace

hipError_t launch_score_hist(const HistArgs& a, hipStream_t stream) {
    if (a.nq == 0 || a.nranges == 0) return hipSuccess;
    if (a.nbins == 0) return hipErrorInvalidValue;
    const dim3 grid(a.nq, a.nranges);
    if (a.score_bytes == 1) hipLaunchKernelGGL(score_hist_kernel<uint8_t>, grid, dim3(256), 0, stream, a);
    else if (a.score_bytes == 2) hipLaunchKernelGGL(score_hist_kernel<uint16_t>, grid, dim3(256), 0, stream, a);
    else if (a.score_bytes == 4) hipLaunchKernelGGL(score_hist_kernel<uint32_t>, grid, dim3(256), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_pack_slots(const SlotPackArgs& a, hipStream_t stream) {
    if (a.nq == 0 || a.words == 0) return hipSuccess;
    if (a.slot_bits == 0 || a.slot_bits > 31) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_slots_kernel, dim3((a.words + 255u) / 256u, std::min<uint32_t>(a.nq, 65535u)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rank(const RankArgs& a, bool first, bool last, hipStream_t stream) {
    if (a.nq == 0) return hipSuccess;
    if (a.bits == 0 || a.bits > 12 || a.nparts == 0) return hipErrorInvalidValue;
    const size_t lds = ((size_t)4 * (1u << a.bits) + 256 + 4 * 512) * sizeof(uint32_t);
    if (!first) return launch_rank_t<uint32_t>(a, first, last, lds, stream);      // pairs: the score type is not used
    if (a.score_bytes == 1) return launch_rank_t<uint8_t>(a, first, last, lds, stream);
    if (a.score_bytes == 2) return launch_rank_t<uint16_t>(a, first, last, lds, stream);
    if (a.score_bytes == 4) return launch_rank_t<uint32_t>(a, first, last, lds, stream);
    return hipErrorInvalidValue;
}

}  // namespace cobs_amd
