// cobs_amd/csrc/fetch_kernels.hip -- row-selective access to an index that is NOT resident in HBM.
//
// The reference's out-of-core back-ends touch only the rows a query addresses: the mmap back-end maps the file
// with MADV_RANDOM and copies the T*H addressed rows (cobs/util/query.cpp:43-55,
// compact_index/mmap_search_file.cpp:34-67), the AIO back-end issues one pread per (sub-index, hash)
// (aio_search_file.cpp:58-97).  Streaming a whole sub-index through HBM is the right thing only when a batch
// looks up more bytes than the sub-index holds; for a single query it moves gigabytes to touch megabytes.
//
//   count_rows_kernel   right after K1: how many rows does the batch look up in every streamed piece of the file (a whole
//                       slice, or each row range of a sub-index that is cut by rows)?  One counter each; the host reads
//                       them and decides per chunk -- exactly, not by expectation -- whether its looked-up rows are
//                       fewer bytes than the chunk.
//   gather_assign_kernel / gather_copy_kernel
//                       for one unit (a streamed chunk, or a run of them merged): a slot of the gathered buffer for
//                       exactly the looked-up rows of each of its pages (a wave-aggregated atomic per page: the order of
//                       the slots does not matter), the second row-index table (entry -> its slot; padding entries and
//                       rows outside a row range -> the page's zero row) and the PageDev array describing the buffer;
//                       then the rows themselves, read from the index file -- whose mapping is registered with HIP, so
//                       the loads go over PCIe straight from the page cache, 16 bytes per lane, a row's pieces on
//                       consecutive lanes (1 KiB per wave-load) -- into the buffer, laid out like a resident chunk (same
//                       pitch, a zero row behind every page's rows).  K2 then scans it with the code it runs on resident
//                       data.  [Rounds 3-4 gave every table entry a slot: (entries + 1) x pitch bytes per page, which a
//                       256 MiB stream buffer does not hold for 256 queries.  Packed, the gathered rows fit whenever
//                       they are fewer bytes than the chunk -- which is when they are fetched at all.]
//
// A row looked up twice is fetched twice: the engine chooses this path only when the batch's lookups are a
// fraction of the chunk's rows (pass.cpp: run_impl), where repeats are rare.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "device_types.hpp"
#include "kernels.hpp"

namespace cobs_amd {

namespace {

// 16 bytes from an arbitrarily aligned address, reading only aligned dwords that hold wanted bytes
template <bool NT = false>
__device__ __forceinline__ uint4 load16_any(const uint8_t* p, uint32_t nvalid) {
    const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p - mis);
    if (mis == 0u && nvalid == 16u && ((uintptr_t)p & 15u) == 0u) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        if (NT) { const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
        return *reinterpret_cast<const uint4*>(p);
    }
    uint32_t r[5];
#pragma unroll
    for (uint32_t j = 0; j < 5; ++j) r[j] = (4u * j < mis + nvalid) ? w[j] : 0u;
    uint32_t o[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        o[j] = mis == 0u ? r[j] : (uint32_t)(((uint64_t)r[j] | ((uint64_t)r[j + 1] << 32)) >> (8u * mis));
    // bytes at and beyond nvalid read as zero (pitch padding of the resident layout)
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t have = nvalid > 4u * j ? nvalid - 4u * j : 0u;
        if (have < 4u) o[j] &= have == 0u ? 0u : (1u << (8u * have)) - 1u;
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// query of table entry n of ONE sub-index ([query][block + padding block][hash][8]): the last q with (blk_off[q] + q) * per <= n
__device__ __forceinline__ uint32_t query_of_entry(const uint64_t* blk_off, uint32_t nq, uint64_t per, uint64_t n) {
    uint32_t lo = 0, hi = nq;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((blk_off[mid] + mid) * per <= n) lo = mid; else hi = mid;
    }
    return lo;
}

// How many rows does the batch look up in every streamed piece of the file?  One thread per table entry of the part
// (all sub-indexes); an LDS histogram per work-group when there are few counters, one 64-bit atomic per non-empty bin.
constexpr uint32_t kCountLds = 2048;
template <typename IdxT>
__global__ __launch_bounds__(256) void count_rows_kernel(CountArgs a, uint64_t total) {
    __shared__ uint32_t h[kCountLds];
    const bool lds = a.ncounters <= kCountLds;
    if (lds) {
        for (uint32_t i = threadIdx.x; i < a.ncounters; i += 256u) h[i] = 0u;
        __syncthreads();
    }
    const uint64_t per = 8ull * a.num_hashes;
    for (uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x; g < total; g += (uint64_t)gridDim.x * 256u) {
        // entry g of the whole table: [query][sub-index][block + padding block][hash][8]
        const uint64_t perq = per * a.table_npages;
        uint32_t lo = 0, hi = a.nq;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((a.blk_off[mid] + mid) * perq <= g) lo = mid; else hi = mid;
        }
        const uint64_t nblk1 = a.blk_off[lo + 1] - a.blk_off[lo] + 1u;
        const uint64_t within = g - (a.blk_off[lo] + lo) * perq;
        const uint32_t p = (uint32_t)(within / (nblk1 * per));
        const CountPage cp = a.cpages[p];
        if (cp.first == 0xFFFFFFFFu) continue;
        const uint64_t r = (uint64_t)reinterpret_cast<const IdxT*>(a.table)[g];
        if (r >= a.tpages[p].sig) continue;               // K1 points padded terms at row S_p
        uint32_t k = cp.first;
        if (cp.per) {
            const uint64_t i = r / cp.per;
            k += (uint32_t)(i < cp.n ? i : cp.n - 1u);
        }
        if (lds) atomicAdd(&h[k], 1u);
        else atomicAdd(&a.counts[k], 1ull);
    }
    if (lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < a.ncounters; i += 256u)
            if (h[i]) atomicAdd(&a.counts[i], (unsigned long long)h[i]);
    }
}

// Slots of the gathered buffer for exactly the DISTINCT looked-up rows of a unit's pages, in ascending row order (round 6).
// A row looked up by several terms of the batch crossed PCIe once per look-up until now: 10 000 queries x 1000 terms look
// up 10 M rows in every sub-index, of which 8.9 M are distinct in a 40 M-row sub-index and 7.7 M in an 18 M-row one --
// 17 % of the bytes of the 184 GB pass, which is bound by exactly those bytes.  Four small kernels per unit, all on the
// stream beside the previous unit's copy:
//   mark    grid.y = page (leaders only do work), grid.x over the table entries of its sub-index: bit r of the page's
//           bitmap = some entry looks up row row0 + r
//   rank    one work-group per leader page: set bits in front of every bitmap word (exclusive scan), their total = the
//           page's distinct rows (GatherArgs::cursor)
//   assign  the second row-index table: entry -> bprefix[word] + bits below its row (padding entries and rows outside a
//           row range -> the page's zero row).  No atomics: a slot is a function of the row.
//   list    the source row of every slot, from the bitmap: ascending -- the copy sweeps the file front to back
template <typename IdxT>
__device__ __forceinline__ bool gather_entry(const GatherArgs& a, const GatherPage& pg, uint64_t n, uint64_t* e, uint64_t* r) {
    const uint64_t per = 8ull * a.num_hashes;
    const uint32_t q = query_of_entry(a.blk_off, a.nq, per, n);
    const uint64_t b0 = a.blk_off[q];
    const uint64_t nblk1 = a.blk_off[q + 1] - b0 + 1u;
    *e = ((b0 + q) * a.table_npages + (uint64_t)pg.tpage * nblk1) * per + (n - (b0 + q) * per);
    // (a row outside the page's range -- and K1's padding row S_p, which lies beyond every range -- is not gathered:
    // its entry names the page's zero row)
    *r = (uint64_t)reinterpret_cast<const IdxT*>(a.table)[*e] - pg.row0;
    return *r < pg.nrows;
}

template <typename IdxT>
__global__ __launch_bounds__(256) void gather_mark_kernel(GatherArgs a) {
    const uint32_t i = blockIdx.y;
    const GatherPage pg = a.pages[i];
    if (pg.leader != i) return;
    const uint64_t n = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (n >= a.entries) return;
    uint64_t e, r;
    if (gather_entry<IdxT>(a, pg, n, &e, &r)) atomicOr(&a.bitmap[pg.bm_off + (r >> 5)], 1u << (uint32_t)(r & 31u));
}

__global__ __launch_bounds__(1024) void gather_rank_kernel(GatherArgs a) {
    __shared__ uint32_t part[1024];
    const uint32_t i = blockIdx.x;
    const GatherPage pg = a.pages[i];
    if (pg.leader != i) return;
    const uint32_t t = threadIdx.x;
    const uint32_t per = (pg.bm_words + 1023u) / 1024u;
    const uint32_t w0 = min(t * per, pg.bm_words), w1 = min(w0 + per, pg.bm_words);
    const uint32_t* bm = a.bitmap + pg.bm_off;
    uint32_t* pre = a.bprefix + pg.bm_off;
    uint32_t s = 0;
    for (uint32_t w = w0; w < w1; ++w) s += (uint32_t)__popc(bm[w]);
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = t ? part[t - 1] : 0u;
    for (uint32_t w = w0; w < w1; ++w) {
        pre[w] = run;
        run += (uint32_t)__popc(bm[w]);
    }
    if (t == 1023u) a.cursor[i] = part[1023];
}

template <typename IdxT>
__global__ __launch_bounds__(256) void gather_assign_kernel(GatherArgs a) {
    const uint32_t i = blockIdx.y;
    const GatherPage pg = a.pages[i];
    if (pg.leader != i) return;
    const uint64_t n = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (n >= a.entries) return;
    uint64_t e, r;
    uint32_t slot = pg.count;
    if (gather_entry<IdxT>(a, pg, n, &e, &r)) {
        const uint64_t w = pg.bm_off + (r >> 5);
        slot = a.bprefix[w] + (uint32_t)__popc(a.bitmap[w] & ((1u << (uint32_t)(r & 31u)) - 1u));
    }
    reinterpret_cast<IdxT*>(a.table2)[e] = (IdxT)slot;
}

__global__ __launch_bounds__(256) void gather_list_kernel(GatherArgs a) {
    const uint32_t i = blockIdx.y;
    const GatherPage pg = a.pages[i];
    if (pg.leader != i) return;
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= pg.bm_words) return;
    uint32_t x = a.bitmap[pg.bm_off + w];
    uint64_t slot = pg.slot0 + a.bprefix[pg.bm_off + w];
    while (x != 0u) {
        const uint32_t b = (uint32_t)__ffs((int)x) - 1u;
        x &= x - 1u;
        a.rowlist[slot++] = (uint64_t)w * 32u + b;       // (slot < count always: the counts bound the distinct rows)
    }
}

// One thread per (gathered row, 16-byte piece): the row's pieces on consecutive lanes, read over PCIe straight from the
// registered mapping of the index file (unaligned sources -- classic rows at odd file offsets -- through aligned dwords).
struct CopyPiece { const uint8_t* src; uint8_t* out; uint32_t nvalid; bool store; };
__device__ __forceinline__ CopyPiece gather_copy_plan(const GatherArgs& a, uint32_t cpp, uint64_t gid) {
    const uint64_t g = gid / cpp;                          // gathered row
    const uint32_t c = (uint32_t)(gid - g * cpp);
    uint32_t lo = 0, hi = a.npages;                        // its page: the last i with slot0[i] <= g
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.pages[mid].slot0 <= g) lo = mid; else hi = mid;
    }
    const GatherPage pg = a.pages[lo];
    const uint64_t local = g - pg.slot0;
    CopyPiece p;
    p.out = a.dst + g * a.pitch + (uint64_t)c * 16u;
    p.src = nullptr;
    p.nvalid = 0;
    // rows [0, distinct rows) are gathered rows, row `count` is the page's zero row; the counts bound the distinct rows
    // (a capacity per page), the rows in between are never named
    const uint64_t have = a.cursor[pg.leader];
    if (local == 0u && c == 0u) {                          // the page as the gathered buffer holds it
        PageDev d = a.pages_in[lo];
        d.base = pg.slot0 * a.pitch;
        d.sig = pg.count;
        d.magic = 0;
        d.row0 = 0;
        a.pages2[lo] = d;
        if (a.fetched_bytes) atomicAdd(a.fetched_bytes, (unsigned long long)min(have, (unsigned long long)pg.count) * a.pitch);
    }
    p.store = local == pg.count || local < have;
    if (local < have && local < pg.count) {
        const uint64_t r = a.rowlist[a.pages[pg.leader].slot0 + local];
        p.nvalid = pg.valid_bytes > c * 16u ? min(pg.valid_bytes - c * 16u, 16u) : 0u;
        p.src = a.file + pg.src + r * a.src_pitch + (uint64_t)c * 16u;
    }
    return p;
}
template <bool NT>
__device__ __forceinline__ uint4 gather_copy_load(const CopyPiece& p) {
    return p.nvalid ? load16_any<NT>(p.src, p.nvalid) : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void gather_copy_piece(const GatherArgs& a, uint32_t cpp, uint64_t gid) {
    const CopyPiece p = gather_copy_plan(a, cpp, gid);
    if (!p.store) return;
    *reinterpret_cast<uint4*>(p.out) = (a.exp & 1u) ? gather_copy_load<true>(p) : gather_copy_load<false>(p);
}

// A BOUNDED grid with a grid-stride loop: the kernel waits for PCIe (a few MB in flight saturate the link), and a grid of
// one thread per piece -- 45 000 work-groups for a 186 MB unit -- filled every CU with waiting waves: the scan and the
// partial-score addition of the previous unit, on their own streams, then took 3.4 ms instead of 0.2 for want of a slot.
__global__ __launch_bounds__(256) void gather_copy_kernel(GatherArgs a) {
    const uint32_t cpp = a.pitch / 16u;
    const uint64_t items = a.total_rows * cpp;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (a.exp & 2u) {
        for (; gid + stride < items; gid += 2u * stride) {          // two pieces per thread in flight over PCIe
            const CopyPiece p0 = gather_copy_plan(a, cpp, gid), p1 = gather_copy_plan(a, cpp, gid + stride);
            const uint4 v0 = gather_copy_load<false>(p0), v1 = gather_copy_load<false>(p1);
            if (p0.store) *reinterpret_cast<uint4*>(p0.out) = v0;
            if (p1.store) *reinterpret_cast<uint4*>(p1.out) = v1;
        }
    }
    for (; gid < items; gid += stride) gather_copy_piece(a, cpp, gid);
}

// Row-range chunk, streamed whole: K1's row indices of its sub-index, rewritten for a buffer that holds rows
// [row0, row0 + nrows) only.  One thread per table entry of that sub-index ([query][block + padding block][hash][8]).
template <typename IdxT>
__global__ __launch_bounds__(256) void remap_rows_kernel(RemapArgs a, uint64_t entries) {
    const uint64_t n = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (n >= entries) return;
    const uint64_t per = 8ull * a.num_hashes;
    uint32_t lo = 0, hi = a.nq;                            // query of entry n: the last q with (blk_off[q] + q) * per <= n
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((a.blk_off[mid] + mid) * per <= n) lo = mid; else hi = mid;
    }
    const uint64_t b0 = a.blk_off[lo];
    const uint64_t nblk1 = a.blk_off[lo + 1] - b0 + 1u;
    const uint64_t e = ((b0 + lo) * a.table_npages + (uint64_t)a.tpage * nblk1) * per + (n - (b0 + lo) * per);
    const uint64_t r = (uint64_t)reinterpret_cast<const IdxT*>(a.table)[e] - a.row0;
    reinterpret_cast<IdxT*>(a.table2)[e] = (IdxT)(r < a.nrows ? r : a.nrows);
}

// ---- the in-range terms of a row-range unit, per query (CompactArgs) ----
// one wave per query: how many entries of its sub-index `tpage` name a row of the unit
__global__ __launch_bounds__(64) void compact_count_kernel(CompactArgs a) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint64_t b0 = a.blk_off[q];
    const uint32_t nblk = (uint32_t)(a.blk_off[q + 1] - b0);
    const uint32_t* t = a.table2 + ((b0 + q) * a.table_npages + (uint64_t)a.tpage * (nblk + 1u)) * 8ull;
    uint32_t n = 0;
    for (uint32_t i = lane; i < nblk * 8u; i += 64u) n += t[i] != a.zero_idx ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += (uint32_t)__shfl_xor((int)n, off);
    if (lane == 0u) a.cnt[q] = (n + 7u) / 8u;
}

// blk2 = exclusive scan of cnt[0 .. nq), blk2[nq] = total; ONE work-group of 1024 threads
__global__ __launch_bounds__(1024) void compact_scan_kernel(CompactArgs a) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (a.nq + 1023u) / 1024u;
    const uint32_t i0 = min(t * per, a.nq), i1 = min(i0 + per, a.nq);
    uint64_t s = 0;
    for (uint32_t i = i0; i < i1; ++i) s += a.cnt[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = t ? part[t - 1] : 0ull;
    for (uint32_t i = i0; i < i1; ++i) {
        a.blk2[i] = run;
        run += a.cnt[i];
    }
    if (t == 1023u) a.blk2[a.nq] = part[1023];
}

// one wave per query: its in-range entries packed to the front of its compact blocks, the rest (and the extra all-padding
// block) point at the zero row
__global__ __launch_bounds__(64) void compact_write_kernel(CompactArgs a) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint64_t b0 = a.blk_off[q];
    const uint32_t nblk = (uint32_t)(a.blk_off[q + 1] - b0);
    const uint32_t* t = a.table2 + ((b0 + q) * a.table_npages + (uint64_t)a.tpage * (nblk + 1u)) * 8ull;
    const uint64_t c0 = a.blk2[q];
    const uint32_t nb2 = (uint32_t)(a.blk2[q + 1] - c0);
    uint32_t* o = a.table3 + ((c0 + q) * a.table_npages + (uint64_t)a.tpage * (nb2 + 1u)) * 8ull;
    uint32_t off = 0;
    for (uint32_t i0 = 0; i0 < nblk * 8u; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const uint32_t v = i < nblk * 8u ? t[i] : a.zero_idx;
        const bool in = v != a.zero_idx;
        const unsigned long long mask = __ballot(in);
        if (in) o[off + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = v;
        off += (uint32_t)__popcll(mask);
    }
    for (uint32_t i = off + lane; i < (nb2 + 1u) * 8u; i += 64u) o[i] = a.zero_idx;
}

// dst[q][dst_offset + i] += src[q][i]: whole 32-bit words (slots come in multiples of 8; the counters of a word
// cannot carry into each other: every sum is a count of the query's terms, which the score type holds)
__global__ __launch_bounds__(256) void add_scores_kernel(AddScoresArgs a) {
    const uint64_t words_per_row = (uint64_t)a.nslots * a.elem_bytes / 4u;
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t q = gid / words_per_row;
    if (q >= a.nq) return;
    const uint64_t w = gid - q * words_per_row;
    uint32_t* d = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(a.dst) + (q * a.dst_stride + a.dst_offset) * a.elem_bytes) + w;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(a.src) + q * (uint64_t)a.nslots * a.elem_bytes) + w;
    *d += *s;
}

}  // namespace

hipError_t launch_compact_terms(const CompactArgs& a, hipStream_t stream) {
    if (a.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_count_kernel, dim3(a.nq), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(compact_write_kernel, dim3(a.nq), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_remap_rows(const RemapArgs& a, uint64_t entries, bool idx64, hipStream_t stream) {
    if (entries == 0 || a.nq == 0) return hipSuccess;
    const uint64_t blocks = (entries + 255u) / 256u;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (idx64) hipLaunchKernelGGL(remap_rows_kernel<uint64_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a, entries);
    else hipLaunchKernelGGL(remap_rows_kernel<uint32_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a, entries);
    return hipGetLastError();
}

// The four flag words of a batch (first invalid query | - | 64-bit hit count) back to zero at the head of a pass.
// A kernel, not hipMemsetD32Async: small host passes are captured into hipGraphs, and a captured MEMSET node of a graph
// that is replayed after other graphs were instantiated on the stream wrote 16 bytes of stale host data over the
// flags instead of zeros (ROCm 7.2, MI355X; the pass then reported "invalid base pair in query 998395903" for a
// valid query -- found by tests/test_gpu_fuzz.py::test_random_ties under COBS_FUZZ_SEED=14).  A kernel node carries
// its arguments by value.
__global__ void clear_flags_kernel(uint32_t* flags) {
    if (threadIdx.x < 4u) flags[threadIdx.x] = 0u;
}

hipError_t launch_clear_flags(uint32_t* flags, hipStream_t stream) {
    hipLaunchKernelGGL(clear_flags_kernel, dim3(1), dim3(64), 0, stream, flags);
    return hipGetLastError();
}

hipError_t launch_add_scores(const AddScoresArgs& a, hipStream_t stream) {
    if (a.nq == 0 || a.nslots == 0) return hipSuccess;
    if ((a.nslots % 8u) != 0 || (a.dst_offset % 8u) != 0 || (a.dst_stride % 8u) != 0) return hipErrorInvalidValue;
    const uint64_t items = (uint64_t)a.nq * ((uint64_t)a.nslots * a.elem_bytes / 4u);
    const uint64_t blocks = (items + 255u) / 256u;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_scores_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_count_rows(const CountArgs& a, uint64_t total_entries, bool idx64, hipStream_t stream) {
    if (total_entries == 0 || a.nq == 0 || a.ncounters == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total_entries + 255u) / 256u, 8192u);
    if (idx64) hipLaunchKernelGGL(count_rows_kernel<uint64_t>, dim3(blocks), dim3(256), 0, stream, a, total_entries);
    else hipLaunchKernelGGL(count_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, stream, a, total_entries);
    return hipGetLastError();
}

// max_words: the longest bitmap among the unit's pages (grid of the list kernel)
hipError_t launch_gather_assign(const GatherArgs& a, bool idx64, uint32_t max_words, hipStream_t stream) {
    if (a.npages == 0 || a.nq == 0 || a.entries == 0) return hipSuccess;
    const uint64_t ablocks = (a.entries + 255u) / 256u;
    if (ablocks > 0x7FFFFFFFull || a.npages > 65535u) return hipErrorInvalidValue;
    const dim3 agrid((uint32_t)ablocks, a.npages);
    if (idx64) hipLaunchKernelGGL(gather_mark_kernel<uint64_t>, agrid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gather_mark_kernel<uint32_t>, agrid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(gather_rank_kernel, dim3(a.npages), dim3(1024), 0, stream, a);
    if (idx64) hipLaunchKernelGGL(gather_assign_kernel<uint64_t>, agrid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gather_assign_kernel<uint32_t>, agrid, dim3(256), 0, stream, a);
    if (max_words) hipLaunchKernelGGL(gather_list_kernel, dim3((max_words + 255u) / 256u, a.npages), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_gather_copy(const GatherArgs& a, uint32_t grid_limit, hipStream_t stream) {
    if (a.npages == 0 || a.nq == 0 || a.entries == 0) return hipSuccess;
    const uint64_t items = a.total_rows * (a.pitch / 16u);
    const uint64_t cblocks = (items + 255u) / 256u;
    static const uint64_t forced = getenv("COBS_GPU_GATHER_BLOCKS") ? std::strtoull(getenv("COBS_GPU_GATHER_BLOCKS"), nullptr, 0) : 0u;   // (A/B)
    const uint64_t max_blocks = forced ? forced : std::max<uint32_t>(grid_limit, 1u);
    hipLaunchKernelGGL(gather_copy_kernel, dim3((uint32_t)std::min<uint64_t>(cblocks, max_blocks)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cobs_amd
