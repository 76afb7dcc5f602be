// cobs_amd/csrc/stage.cpp -- index data into HBM: device allocations of a planned part (alloc_part), the
// resident upload from the mapped file (upload_resident: slabs through two pinned buffers, re-pitched on the
// device where the file's row size is not the device pitch; replaces initialize_mmap, reference
// cobs/util/query.cpp:38-88), and the out-of-core path's whole-chunk copy (stream_chunk_in; the successor of
// the reference's mmap / AIO back-ends).  The row-selective alternative to stream_chunk_in is in pass.cpp /
// fetch_kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

using namespace cobs_amd;

namespace cobs_amd {

cobs_gpu_status alloc_part(cobs_gpu_index* ix, Part& pt) {
    for (Chunk& c : pt.chunks) {
        HIP_TRY(hipMalloc((void**)&c.d_pages, sizeof(PageDev) * c.pages.size()));
        HIP_TRY(hipMemcpy(c.d_pages, c.pages.data(), sizeof(PageDev) * c.pages.size(), hipMemcpyHostToDevice));
        if (c.row_range) {     // partial scores (a later range, or any range of a pass without score rows) go to a scratch matrix of the slice's own width
            std::vector<PageDev> acc = c.pages;
            for (PageDev& pd : acc) pd.slot0 = 0;
            HIP_TRY(hipMalloc((void**)&c.d_pages_acc, sizeof(PageDev) * acc.size()));
            HIP_TRY(hipMemcpy(c.d_pages_acc, acc.data(), sizeof(PageDev) * acc.size(), hipMemcpyHostToDevice));
        }
    }
    if (pt.chunks.empty()) return COBS_GPU_OK;
    HIP_TRY(hipMalloc((void**)&pt.d_tpages, sizeof(PageDev) * pt.tpages.size()));
    HIP_TRY(hipMemcpy(pt.d_tpages, pt.tpages.data(), sizeof(PageDev) * pt.tpages.size(), hipMemcpyHostToDevice));
    if (!pt.streamed) {
        for (Chunk& c : pt.chunks) HIP_TRY(hipMalloc((void**)&c.d_data, c.bytes));
        return COBS_GPU_OK;
    }
    StreamBufs& sb = ix->stream;
    size_t dev = 0, host = 0;
    for (Chunk& c : pt.chunks) {
        if (c.resident) {            // a slice the budget keeps in HBM beside the stream buffers
            HIP_TRY(hipMalloc((void**)&c.d_data, c.bytes));
            continue;
        }
        dev = std::max(dev, c.bytes);
        host = std::max(host, c.stage_bytes);
    }
    sb.stage_need = std::max(sb.stage_need, host);
    for (int i = 0; i < 2; ++i) {
        if (sb.sbuf[i].cap < dev) {
            // grow keeping nothing: buffers are only (re)allocated while the index is opened
            HIP_TRY(sb.sbuf[i].reserve(dev));
        }
        if (!sb.copied[i]) HIP_TRY(hipEventCreateWithFlags(&sb.copied[i], hipEventDisableTiming));
        if (!sb.scanned[i]) HIP_TRY(hipEventCreateWithFlags(&sb.scanned[i], hipEventDisableTiming));
    }
    if (!sb.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&sb.copy_stream, hipStreamNonBlocking));
    if (!sb.prep_stream) HIP_TRY(hipStreamCreateWithFlags(&sb.prep_stream, hipStreamNonBlocking));
    for (auto& e : sb.assigned) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return COBS_GPU_OK;
}

// Resident chunks: copy the held columns of every held sub-index from the mapped file into HBM.
// The index file (an mmap of the page cache) -> HBM.  Rows travel in slabs of up to 256 MiB: host
// threads copy a slab from the mapping into one of two pinned buffers while the previous slab is on
// its way over PCIe (and, when the device pitch differs from the file's row size, through the
// re-pitch kernel) -- a plain hipMemcpy from pageable memory measured 10-25 GB/s here.
cobs_gpu_status upload_resident(Part& pt, const uint8_t* file) {
    const IndexMeta& m = pt.meta;
    const uint64_t src_pitch = m.page_row_bytes();
    constexpr uint64_t kSlab = 256ull << 20;
    struct Slab {
        PinnedBuf<uint8_t> host;
        DevBuf<uint8_t> dev;                 // raw rows on the device, only when they are re-pitched
        hipEvent_t done = nullptr;
        bool busy = false;
        ~Slab() { if (done) (void)hipEventDestroy(done); }
    } slab[2];
    hipStream_t stream = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } sg{stream};
    for (Slab& sl : slab) HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    const size_t nthreads = std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    auto copy_in = [&](uint8_t* dst, const uint8_t* src, uint64_t bytes) {
        if (bytes < (8u << 20) || nthreads == 1) { std::memcpy(dst, src, (size_t)bytes); return; }
        std::vector<std::thread> pool;
        const uint64_t per = (bytes / nthreads + 4095) / 4096 * 4096;
        for (size_t t = 0; t < nthreads; ++t) {
            const uint64_t o = t * per;
            if (o >= bytes) break;
            pool.emplace_back([=]() { std::memcpy(dst + o, src + o, (size_t)std::min(per, bytes - o)); });
        }
        for (auto& t : pool) t.join();
    };
    int cur = 0;
    for (Chunk& c : pt.chunks) {
        if (!c.d_data) continue;         // (a streamed chunk of a part that keeps only some of its slices resident)
        for (size_t lp = 0; lp < c.vp.size(); ++lp) {
            const PageDev& pd = c.pages[lp];
            const VPage& v = c.vp[lp];
            const uint8_t* src = file + m.page_offset(v.fp);
            uint8_t* dst = c.d_data + pd.base;
            const bool straight = src_pitch == c.pitch && v.col0 == 0;      // rows already have the device pitch
            const uint64_t rows_per = std::max<uint64_t>(1, kSlab / src_pitch);
            for (uint64_t r = 0; r < pd.sig; r += rows_per) {
                const uint64_t n = std::min(rows_per, pd.sig - r), bytes = n * src_pitch;
                Slab& sl = slab[cur];
                cur ^= 1;
                if (sl.busy) { HIP_TRY(hipEventSynchronize(sl.done)); sl.busy = false; }
                HIP_TRY(sl.host.reserve((size_t)(std::min(rows_per, pd.sig) * src_pitch)));
                copy_in(sl.host.p, src + r * src_pitch, bytes);
                if (straight) {
                    HIP_TRY(hipMemcpyAsync(dst + r * src_pitch, sl.host.p, (size_t)bytes, hipMemcpyHostToDevice, stream));
                } else {
                    HIP_TRY(sl.dev.reserve(sl.host.cap));
                    HIP_TRY(hipMemcpyAsync(sl.dev.p, sl.host.p, (size_t)bytes, hipMemcpyHostToDevice, stream));
                    RepitchArgs ra;
                    ra.src = sl.dev.p;
                    ra.dst = dst + r * c.pitch;
                    ra.rows = n;
                    ra.src_pitch = (uint32_t)src_pitch;
                    ra.dst_pitch = c.pitch;
                    ra.copy_bytes = (uint32_t)v.ncols;
                    ra.src_col0 = (uint32_t)v.col0;
                    HIP_TRY(launch_repitch(ra, stream));
                }
                HIP_TRY(hipEventRecord(sl.done, stream));
                sl.busy = true;
            }
            HIP_TRY(hipMemsetAsync(dst + pd.sig * (uint64_t)c.pitch, 0, c.pitch, stream));   // zero row
        }
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return COBS_GPU_OK;
}

SynthArgs synth_args(const Part& pt, const Chunk& c, uint8_t* data) {
    SynthArgs sa;
    sa.blob = data;
    sa.pages = c.d_pages;
    sa.seed = pt.synth_seed;
    sa.row_bytes = pt.meta.page_row_bytes();
    sa.col0 = c.vp[0].col0;
    sa.num_docs = pt.meta.doc_names.size();
    sa.page_docs = pt.meta.kind == IndexKind::Compact ? 8 * pt.meta.header_page_size : 0;
    sa.npages = (uint32_t)c.vp.size();
    sa.first_page = c.vp[0].fp;
    sa.pitch = c.pitch;
    return sa;
}

// Streamed chunk: bring it into device buffer `buf` on the copy stream (file-backed:
// DMA from the pinned mapping, or pack the needed columns into pinned staging first;
// procedural: regenerate).
cobs_gpu_status stream_chunk_in(cobs_gpu_index* ix, Part& pt, const Chunk& c, int buf) {
    StreamBufs& sb = ix->stream;
    uint8_t* dev = sb.sbuf[buf].p;
    if (pt.synthetic) {
        HIP_TRY(launch_synth(synth_args(pt, c, dev), sb.copy_stream));
        return COBS_GPU_OK;
    }
    const IndexMeta& m = pt.meta;
    const uint64_t prb = m.page_row_bytes();
    if (!pt.file_pinned) HIP_TRY(sb.stage[buf].reserve(sb.stage_need));
    uint8_t* host = sb.stage[buf].p;
    uint64_t hoff = 0;
    for (size_t i = 0; i < c.vp.size(); ++i) {
        const VPage& v = c.vp[i];
        const PageDev& pd = c.pages[i];
        const uint8_t* src = pt.file->data() + m.page_offset(v.fp) + v.row0 * prb;     // (row-range chunk: its first row)
        uint8_t* dst = dev + pd.base;
        if (pt.file_pinned) {
            if (c.pitch == prb && v.col0 == 0 && v.ncols == prb)       // rows as the file holds them: one linear copy
                HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(pd.sig * prb), hipMemcpyHostToDevice, sb.copy_stream));
            else
                HIP_TRY(hipMemcpy2DAsync(dst, c.pitch, src + v.col0, (size_t)prb, (size_t)v.ncols, (size_t)pd.sig,
                                         hipMemcpyHostToDevice, sb.copy_stream));
            HIP_TRY(hipMemsetAsync(dst + pd.sig * (uint64_t)c.pitch, 0, c.pitch, sb.copy_stream));
            continue;
        }
        uint8_t* hp = host + hoff;
        {   // pack the needed columns into pinned staging with a few host threads
            const unsigned nthr = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(8, pd.sig * v.ncols >> 24));
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nthr; ++t) {
                const uint64_t r0 = pd.sig * t / nthr, r1 = pd.sig * (t + 1) / nthr;
                pool.emplace_back([=]() {
                    if (v.ncols == prb) {
                        std::memcpy(hp + r0 * prb, src + r0 * prb, (size_t)((r1 - r0) * prb));
                    } else {
                        for (uint64_t r = r0; r < r1; ++r)
                            std::memcpy(hp + r * v.ncols, src + r * prb + v.col0, (size_t)v.ncols);
                    }
                });
            }
            for (auto& th : pool) th.join();
        }
        if (c.pitch == v.ncols)
            HIP_TRY(hipMemcpyAsync(dst, hp, (size_t)(pd.sig * v.ncols), hipMemcpyHostToDevice, sb.copy_stream));
        else
            HIP_TRY(hipMemcpy2DAsync(dst, c.pitch, hp, (size_t)v.ncols, (size_t)v.ncols, (size_t)pd.sig,
                                     hipMemcpyHostToDevice, sb.copy_stream));
        HIP_TRY(hipMemsetAsync(dst + pd.sig * (uint64_t)c.pitch, 0, c.pitch, sb.copy_stream));
        hoff += pd.sig * v.ncols;
    }
    return COBS_GPU_OK;
}


}  // namespace cobs_amd
