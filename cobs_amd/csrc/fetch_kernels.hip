// cobs_amd/csrc/fetch_kernels.hip -- row-selective access to an index that is NOT resident in HBM.
//
// The reference's out-of-core back-ends touch only the rows a query addresses: the mmap back-end maps the file
// with MADV_RANDOM and copies the T*H addressed rows (cobs/util/query.cpp:43-55,
// compact_index/mmap_search_file.cpp:34-67), the AIO back-end issues one pread per (sub-index, hash)
// (aio_search_file.cpp:58-97).  Streaming a whole sub-index through HBM is the right thing only when a batch
// looks up more bytes than the sub-index holds; for a single query it moves gigabytes to touch megabytes.
//
//   fetch_rows_kernel   for one streamed chunk (a group of equal-width sub-index slices): reads the row
//                       indices K1 wrote, fetches exactly those rows from the index file -- whose mapping is
//                       registered with HIP, so the loads go over PCIe straight from the page cache, 16
//                       bytes per lane, a row's pieces on consecutive lanes (1 KiB per wave-load) -- into a
//                       gathered buffer in HBM laid out like a resident chunk (same pitch, a zero row behind
//                       every slice's rows), and writes the row-index table of that buffer (entry e -> gathered
//                       row e; padding entries -> the zero row) plus the PageDev array describing it.  K2 then scans the
//                       gathered buffer with the code it runs on resident data.
//
// A row looked up twice is fetched twice: the engine chooses this path only when the batch's lookups are a
// fraction of the sub-index's rows (engine.cpp: run_impl), where repeats are rare.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_types.hpp"
#include "kernels.hpp"

namespace cobs_amd {

namespace {

// 16 bytes from an arbitrarily aligned address, reading only aligned dwords that hold wanted bytes
__device__ __forceinline__ uint4 load16_any(const uint8_t* p, uint32_t nvalid) {
    const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p - mis);
    if (mis == 0u && nvalid == 16u && ((uintptr_t)p & 15u) == 0u) return *reinterpret_cast<const uint4*>(p);
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

template <typename IdxT>
__global__ __launch_bounds__(256) void fetch_rows_kernel(FetchArgs a) {
    const uint32_t cpp = a.pitch / 16u;
    const uint64_t E = a.entries;
    const uint64_t rows = (uint64_t)a.npages * (E + 1u);   // every page: its E gathered rows, then its zero row
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t rowno = gid / cpp;                      // page-major gathered row
    const uint32_t c = (uint32_t)(gid - rowno * cpp);
    if (rowno >= rows) return;
    uint8_t* out = a.dst + rowno * a.pitch + (uint64_t)c * 16u;
    const uint32_t i = (uint32_t)(rowno / (E + 1u));       // page of the chunk
    const uint64_t n = rowno - (uint64_t)i * (E + 1u);     // entry of that page: [query][block + padding block][hash][8]
    if (n == E) {                                          // the page's zero row: what its padding entries point at
        *reinterpret_cast<uint4*>(out) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const PageDev pd = a.pages[i];
    if (n == 0u && c == 0u) {                              // the page as the gathered buffer holds it
        PageDev g = pd;
        g.base = (uint64_t)i * (E + 1u) * a.pitch;
        g.sig = E;
        g.row0 = 0;
        a.pages2[i] = g;
    }
    const uint64_t per = 8ull * a.num_hashes;
    // query of entry n: the last q with (blk_off[q] + q) * per <= n
    uint32_t lo = 0, hi = a.nq;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((a.blk_off[mid] + mid) * per <= n) lo = mid; else hi = mid;
    }
    const uint64_t b0 = a.blk_off[lo];
    const uint64_t nblk1 = a.blk_off[lo + 1] - b0 + 1u;   // blocks of the query incl. its padding block
    const uint64_t within = n - (b0 + lo) * per;
    const uint64_t e = ((b0 + lo) * a.table_npages + (uint64_t)pd.tpage * nblk1) * per + within;
    // (a row-range chunk holds rows [row0, row0 + sig) of its sub-index: a row outside it -- and K1's padding row S_p,
    // which lies beyond every range -- reads as the zero row; page_src points at the range's first row)
    const uint64_t r = (uint64_t)reinterpret_cast<const IdxT*>(a.table)[e] - pd.row0;
    const bool pad = r >= pd.sig;                          // K1 points padded terms at row S_p
    if (c == 0u)
        reinterpret_cast<IdxT*>(a.table2)[e] = (IdxT)(pad ? E : n);     // (several slices of one sub-index share these entries)
    if (pad) return;                                       // never read: its table entry names the zero row
    const uint32_t nvalid = pd.valid_bytes > c * 16u ? min(pd.valid_bytes - c * 16u, 16u) : 0u;   // slices of one pitch may differ in width
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (nvalid) v = load16_any(a.file + a.page_src[i] + r * a.src_pitch + (uint64_t)c * 16u, nvalid);
    *reinterpret_cast<uint4*>(out) = v;
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

hipError_t launch_fetch_rows(const FetchArgs& a, bool idx64, hipStream_t stream) {
    if (a.npages == 0 || a.nq == 0 || a.entries == 0) return hipSuccess;
    const uint64_t items = (uint64_t)a.npages * (a.entries + 1u) * (a.pitch / 16u);
    const uint64_t blocks = (items + 255u) / 256u;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (idx64) hipLaunchKernelGGL(fetch_rows_kernel<uint64_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(fetch_rows_kernel<uint32_t>, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cobs_amd
