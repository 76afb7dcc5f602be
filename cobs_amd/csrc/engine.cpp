// cobs_amd/csrc/engine.cpp -- host side of libcobs_gpu.so: index staging into
// HBM, batch workspaces, kernel orchestration, result ranking, and the C ABI of
// include/cobs_gpu.h.  There is no CPU fallback: without a HIP device every entry
// point that needs one fails with COBS_GPU_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>

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

// ---------------------------------------------------------------------------
// errors

namespace {
thread_local std::string g_last_error;
}

namespace cobs_amd {

std::string& last_error_text() { return g_last_error; }

cobs_gpu_status fail(cobs_gpu_status st, const std::string& msg) {
    g_last_error = msg;
    return st;
}

cobs_gpu_status hip_fail(hipError_t e, const char* what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    const bool nodev = e == hipErrorNoDevice || e == hipErrorInvalidDevice ||
                       e == hipErrorInsufficientDriver;
    (void)hipGetLastError();
    return fail(nodev ? COBS_GPU_ERR_NO_DEVICE : COBS_GPU_ERR_HIP, m);
}

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

Tuning Tuning::from_env() {
    Tuning t;
    if (const char* e = getenv("COBS_GPU_ROW_ALIGN")) {
        const uint64_t v = std::strtoull(e, nullptr, 10);
        if (v >= 16 && v <= 4096 && v % 16 == 0) t.row_align = (uint32_t)v;
    }
    if (const char* e = getenv("COBS_GPU_WAVES")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) t.waves = (uint32_t)v;
    }
    if (const char* e = getenv("COBS_GPU_TILE_W")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) t.tile_w = (uint32_t)v;
    }
    if (const char* e = getenv("COBS_GPU_MQ")) t.mq = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_PASS_BYTES")) t.pass_bytes = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    if (const char* e = getenv("COBS_GPU_PIPE_CHARS")) t.pipe_chars = std::strtoull(e, nullptr, 10);
    if (getenv("COBS_GPU_NO_PIN")) t.no_pin = true;
    if (const char* e = getenv("COBS_GPU_GRAPH")) t.graph = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_LDS_STAGED")) t.lds_staged = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_DEVICE_RANK")) t.device_rank = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_TRACE")) t.trace = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_ROW_RANGES")) t.row_ranges = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_STREAM_PACKED")) t.stream_packed = atoi(e) != 0;
    if (const char* e = getenv("COBS_GPU_ROW_RANGE_MIN")) t.row_range_min = (uint32_t)std::max(1, atoi(e));
    if (const char* e = getenv("COBS_GPU_STREAM_BUF_KIB")) t.stream_buf_kib = (uint32_t)std::strtoul(e, nullptr, 0);
    if (const char* e = getenv("COBS_GPU_EXP")) t.exp = (uint32_t)std::strtoul(e, nullptr, 0);      // A/B variants, also under the test suite
    return t;
}

Part::~Part() {
    for (Chunk& c : chunks) {
        if (c.d_data) (void)hipFree(c.d_data);
        if (c.d_pages) (void)hipFree(c.d_pages);
        if (c.d_pages_acc) (void)hipFree(c.d_pages_acc);
        for (auto* p2 : c.d_pages2) if (p2) (void)hipFree(p2);
    }
    for (Chunk& c : fetch_groups) {
        if (c.d_pages) (void)hipFree(c.d_pages);
        for (auto* p2 : c.d_pages2) if (p2) (void)hipFree(p2);
    }
    if (d_tpages) (void)hipFree(d_tpages);
    if (d_cpages) (void)hipFree(d_cpages);
    if (file_pinned && file && pin_base) (void)hipHostUnregister(pin_base);       // (`file`: null in a moved-from Part)
}

cobs_gpu_status ResultArena::reserve(size_t n) {
    if (n <= cap) return COBS_GPU_OK;
    if (p) (void)munmap(p, round_up(cap * sizeof(cobs_gpu_hit), 2u << 20));
    p = nullptr;
    cap = 0;
    const size_t bytes = round_up(n * sizeof(cobs_gpu_hit), 2u << 20);
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return fail(COBS_GPU_ERR_CAPACITY, "no memory for the result arena (" + std::to_string(bytes) + " bytes)");
    (void)madvise(m, bytes, MADV_HUGEPAGE);
    p = static_cast<cobs_gpu_hit*>(m);
    cap = bytes / sizeof(cobs_gpu_hit);
    return COBS_GPU_OK;
}

cobs_gpu_status ResultArena::grow_keep(size_t n, size_t used) {
    if (n <= cap) return COBS_GPU_OK;
    if (!p || used == 0) return reserve(n);
    const size_t old_bytes = round_up(cap * sizeof(cobs_gpu_hit), 2u << 20);
    const size_t bytes = round_up(n * sizeof(cobs_gpu_hit), 2u << 20);
    void* m = mremap(p, old_bytes, bytes, MREMAP_MAYMOVE);      // (pages move, nothing is copied)
    if (m == MAP_FAILED) return fail(COBS_GPU_ERR_CAPACITY, "no memory for the result arena (" + std::to_string(bytes) + " bytes)");
    (void)madvise(m, bytes, MADV_HUGEPAGE);
    p = static_cast<cobs_gpu_hit*>(m);
    cap = bytes / sizeof(cobs_gpu_hit);
    return COBS_GPU_OK;
}

ResultArena::~ResultArena() {
    if (p) (void)munmap(p, round_up(cap * sizeof(cobs_gpu_hit), 2u << 20));
}

StreamBufs::~StreamBufs() {
    for (auto& e : copied) if (e) (void)hipEventDestroy(e);
    for (auto& e : scanned) if (e) (void)hipEventDestroy(e);
    if (hashed) (void)hipEventDestroy(hashed);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (prep_stream) (void)hipStreamDestroy(prep_stream);
    for (auto& e : assigned) if (e) (void)hipEventDestroy(e);
}

}  // namespace cobs_amd

cobs_gpu_batch::~cobs_gpu_batch() {
    if (xchg) destroy_exchange(xchg);
    if (rank) destroy_rank_work(rank);
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    for (auto& e : graph_more) if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (graph_stream) (void)hipStreamDestroy(graph_stream);
    for (auto& r : ev) for (auto& e : r) if (e) (void)hipEventDestroy(e);
    if (run_done) (void)hipEventDestroy(run_done);
    if (done) (void)hipEventDestroy(done);
    if (graph_t0) (void)hipEventDestroy(graph_t0);
    if (graph_t1) (void)hipEventDestroy(graph_t1);
    if (own_stream) (void)hipStreamDestroy(own_stream);
    if (hashed) (void)hipEventDestroy(hashed);
    if (hash_stream) (void)hipStreamDestroy(hash_stream);
}

cobs_gpu_index::~cobs_gpu_index() {
    for (auto* b : scratch) delete b;
    if (xchg_stream) (void)hipStreamDestroy(xchg_stream);
}

namespace {

// options fields appended after the first release are honoured only if the caller's struct has them
bool has_field(const cobs_gpu_options* o, size_t end_offset) { return o && o->struct_size >= end_offset; }

cobs_gpu_status read_options(const cobs_gpu_options* o, cobs_gpu_index* ix) {
    ix->tune = Tuning::from_env();
    ix->shard_rank = 0;
    ix->shard_count = 1;
    ix->shard_mode = 0;
    ix->hbm_budget = 0;
    if (!o) return COBS_GPU_OK;
    if (o->shard_count > 1) {
        if (o->shard_rank >= o->shard_count) return fail(COBS_GPU_ERR_ARG, "shard_rank >= shard_count");
        ix->shard_rank = o->shard_rank;
        ix->shard_count = o->shard_count;
    }
    if (o->waves_per_group == 1 || o->waves_per_group == 2 || o->waves_per_group == 4)
        ix->waves_per_group = o->waves_per_group;
    if (o->shard_mode > 2) return fail(COBS_GPU_ERR_ARG, "unknown shard_mode");
    ix->shard_mode = o->shard_mode;
    if (has_field(o, offsetof(cobs_gpu_options, hbm_budget_bytes) + sizeof(uint64_t))) ix->hbm_budget = o->hbm_budget_bytes;
    return COBS_GPU_OK;
}

}  // namespace


// shared with build.cpp
__attribute__((visibility("hidden"))) cobs_gpu_status cobs_gpu_set_error(cobs_gpu_status st, const char* msg) { return fail(st, msg ? msg : ""); }
// ===========================================================================
// C ABI

extern "C" {

uint32_t cobs_gpu_abi_version(void) { return COBS_GPU_ABI_VERSION; }

const char* cobs_gpu_last_error(void) { return g_last_error.c_str(); }

int cobs_gpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// stage every part: resident chunks are uploaded (file) or generated (procedural), streamed
// parts keep their source and get the handle's shared buffers
static cobs_gpu_status stage_index(cobs_gpu_index* ix, std::vector<std::unique_ptr<MappedFile>>& files) {
    for (size_t i = 0; i < ix->parts.size(); ++i) {
        Part& pt = ix->parts[i];
        cobs_gpu_status st = alloc_part(ix, pt);
        if (st != COBS_GPU_OK) return st;
        if (pt.chunks.empty()) continue;
        if (pt.built) {          // rows are written by the caller (index construction straight into HBM)
            for (Chunk& c : pt.chunks) HIP_TRY(hipMemset(c.d_data, 0, c.bytes));
            continue;
        }
        if (pt.synthetic) {
            // (resident chunks are generated once -- all chunks of a resident part, the kept ones of a streamed part;
            // streamed chunks are regenerated at every pass)
            for (Chunk& c : pt.chunks)
                if (c.d_data) HIP_TRY(launch_synth(synth_args(pt, c, c.d_data), nullptr));
            HIP_TRY(hipStreamSynchronize(nullptr));
            continue;
        }
        if (pt.streamed) {
            // the slices the budget keeps resident beside the stream buffers are uploaded once (plan.cpp: chunk_part)
            st = upload_resident(pt, files[i]->data());
            if (st != COBS_GPU_OK) return st;
            pt.file = std::move(files[i]);       // the other chunks are read from the mapping at every pass
            // Pin the read-only mapping so that the copy engine reads it directly (no packing
            // through staging buffers).  Not every kernel/driver allows pinning file pages;
            // if it fails the staged path is used.
            // Only the pages this handle HOLDS are pinned (whole 4 KiB pages around them): a shard of an N-way sharded
            // open reads one contiguous block of sub-indexes, and N ranks pinning all of a 184 GB mapping each would
            // pay N times for pages N - 1 of them never touch.
            uint64_t lo = pt.file->size(), hi = 0;
            for (const VPage& v : pt.held) {
                const uint64_t o = pt.meta.page_offset(v.fp);
                lo = std::min(lo, o);
                hi = std::max(hi, o + pt.meta.signature_sizes[v.fp] * pt.meta.page_row_bytes());
            }
            lo = lo / 4096 * 4096;
            hi = std::min<uint64_t>(round_up(hi, 4096), pt.file->size());
            if (!ix->tune.no_pin && lo < hi) {
                uint8_t* base = const_cast<uint8_t*>(pt.file->data()) + lo;
                hipError_t pe = hipHostRegister(base, hi - lo, hipHostRegisterReadOnly);
                if (pe != hipSuccess) {
                    (void)hipGetLastError();
                    pe = hipHostRegister(base, hi - lo, hipHostRegisterDefault);
                }
                if (pe == hipSuccess) { pt.file_pinned = true; pt.pin_base = base; }
                else (void)hipGetLastError();
            }
            if (pt.file_pinned) {
                // the device-visible address of the mapping and, per chunk, where its slices start in the
                // file: the row-selective pass (fetch_kernels.hip) reads looked-up rows straight from it
                void* dp = nullptr;
                if (hipHostGetDevicePointer(&dp, pt.pin_base, 0) == hipSuccess && dp) {
                    pt.file_dev = static_cast<const uint8_t*>(dp) - lo;      // (offsets stay offsets into the file)
                    // one look-up counter per streamed piece of every sub-index: a whole slice (column slices of one
                    // sub-index see the same rows: one counter), or each of its row ranges (equal length but the last)
                    pt.cpages.assign(pt.tpages.size(), CountPage{0, 0xFFFFFFFFu, 0});
                    pt.ncounters = 0;
                    for (Chunk& c : pt.chunks) {
                        if (c.resident) continue;
                        c.fetch_ok = true;
                        c.src.resize(c.vp.size());
                        c.cp.resize(c.vp.size());
                        for (size_t k = 0; k < c.vp.size(); ++k) {
                            c.src[k] = pt.meta.page_offset(c.vp[k].fp) + c.vp[k].row0 * pt.meta.page_row_bytes() + c.vp[k].col0;
                            CountPage& cpg = pt.cpages[c.vp[k].fp - pt.first_page];
                            if (c.row_range) {
                                if (c.range_no == 0) { cpg.first = pt.ncounters; cpg.per = c.vp[k].nrows; cpg.n = 0; }
                                c.cp[k] = {cpg.first + c.range_no, 1u};
                                cpg.n = std::max(cpg.n, c.range_no + 1u);
                                pt.ncounters = std::max(pt.ncounters, cpg.first + c.range_no + 1u);
                            } else {
                                if (cpg.first == 0xFFFFFFFFu) { cpg.first = pt.ncounters++; cpg.per = 0; cpg.n = 1; }
                                c.cp[k] = {cpg.first, 1u};
                            }
                        }
                        for (auto& p2 : c.d_pages2) HIP_TRY(hipMalloc((void**)&p2, sizeof(PageDev) * c.vp.size()));
                    }
                    HIP_TRY(hipMalloc((void**)&pt.d_cpages, sizeof(CountPage) * std::max<size_t>(pt.cpages.size(), 1)));
                    HIP_TRY(hipMemcpy(pt.d_cpages, pt.cpages.data(), sizeof(CountPage) * pt.cpages.size(), hipMemcpyHostToDevice));
                    // every RUN of consecutive chunks of equal pitch as one unit (pass.cpp uses them when every chunk is
                    // fetched by rows).  Runs, not all chunks of a pitch: the units of a pass are scanned in this
                    // order, and a top-k pass without score rows leaves its candidates in unit order, which K3 takes
                    // to be DOCUMENT order when it cuts ties at the k-th score (topk_kernel<.., POOL>).  [Merging
                    // pages 0, 1, 3 around a column-sliced page 2 returned a document of page 3 in place of one of
                    // page 2 with the same score: scripts/fuzz_soak.sh, seed 35.]
                    if (pt.chunks.size() > 1) {
                        bool run_open = false;       // (a resident chunk between two streamed ones ends the run: unit order = document order)
                        for (size_t ci = 0; ci < pt.chunks.size(); ++ci) {
                            const Chunk& c = pt.chunks[ci];
                            if (c.resident) { run_open = false; continue; }
                            Chunk* g = nullptr;
                            if (run_open && !pt.fetch_groups.empty() && pt.fetch_groups.back().pitch == c.pitch) g = &pt.fetch_groups.back();
                            if (!g) {
                                pt.fetch_groups.emplace_back();
                                g = &pt.fetch_groups.back();
                                g->pitch = c.pitch;
                                g->cpp = c.cpp;
                                g->first_chunk = (uint32_t)ci;
                            }
                            run_open = true;
                            if (c.row_range) {
                                // the row ranges of one sub-index are ONE unit here, with all of its rows: a pass that
                                // fetches by rows gathers the looked-up rows wherever they are
                                if (c.range_no != 0) continue;
                                VPage whole = c.vp[0];
                                whole.row0 = whole.nrows = 0;
                                PageDev pd = c.pages[0];
                                pd.sig = pt.meta.signature_sizes[whole.fp];
                                pd.row0 = 0;
                                pd.magic = ~0ull / pd.sig;
                                pd.base = 0;                  // (a gathered buffer has its own bases: fetch_rows_kernel)
                                g->vp.push_back(whole);
                                g->pages.push_back(pd);
                                g->bytes += round_up((pd.sig + 1) * (uint64_t)c.pitch, 256);     // (= slice_bytes at this chunk's pitch)
                                continue;
                            }
                            g->vp.insert(g->vp.end(), c.vp.begin(), c.vp.end());
                            g->pages.insert(g->pages.end(), c.pages.begin(), c.pages.end());
                            g->bytes += c.bytes;
                        }
                        for (Chunk& g : pt.fetch_groups) {
                            g.total_chunks = (uint32_t)g.vp.size() * g.cpp;
                            g.fetch_ok = true;
                            g.src.resize(g.vp.size());
                            g.cp.resize(g.vp.size());
                            for (size_t k = 0; k < g.vp.size(); ++k) {
                                g.src[k] = pt.meta.page_offset(g.vp[k].fp) + g.vp[k].col0;
                                const CountPage& cpg = pt.cpages[g.vp[k].fp - pt.first_page];
                                g.cp[k] = {cpg.first, cpg.n};          // (all row ranges of the sub-index)
                            }
                            HIP_TRY(hipMalloc((void**)&g.d_pages, sizeof(PageDev) * g.pages.size()));
                            HIP_TRY(hipMemcpy(g.d_pages, g.pages.data(), sizeof(PageDev) * g.pages.size(), hipMemcpyHostToDevice));
                            for (auto& p2 : g.d_pages2) HIP_TRY(hipMalloc((void**)&p2, sizeof(PageDev) * g.vp.size()));
                        }
                    }
                    if (ix->tune.trace) {
                        auto show = [&](const char* what, const Chunk& c) {
                            std::fprintf(stderr, "[cobs_gpu] file %zu %s: pitch %u, %zu bytes%s:", i, what, c.pitch, (size_t)c.bytes,
                                         c.row_range ? " (row range)" : "");
                            for (size_t k = 0; k < c.vp.size(); ++k)
                                std::fprintf(stderr, " [page %u cols %llu+%llu rows %llu+%llu slot0 %u]", c.vp[k].fp,
                                             (unsigned long long)c.vp[k].col0, (unsigned long long)c.vp[k].ncols,
                                             (unsigned long long)c.pages[k].row0, (unsigned long long)c.pages[k].sig, c.pages[k].slot0);
                            std::fprintf(stderr, "\n");
                        };
                        for (const Chunk& c : pt.chunks) show("chunk", c);
                        for (const Chunk& c : pt.fetch_groups) show("fetch group", c);
                    }
                    if (!ix->stream.hashed) HIP_TRY(hipEventCreateWithFlags(&ix->stream.hashed, hipEventDisableTiming));
                } else {
                    (void)hipGetLastError();
                }
            }
        } else {
            st = upload_resident(pt, files[i]->data());
            if (st != COBS_GPU_OK) return st;
        }
    }
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_open(const char* const* paths, size_t n_paths,
                              const cobs_gpu_options* opts, cobs_gpu_index** out) {
    if (!out) return fail(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!paths || n_paths == 0) return fail(COBS_GPU_ERR_ARG, "no index paths");
    return guarded([&]() -> cobs_gpu_status {
        // parse all headers first: format errors are reported even without a device
        std::vector<std::unique_ptr<MappedFile>> files;
        std::unique_ptr<cobs_gpu_index> ix(new cobs_gpu_index);
        for (size_t i = 0; i < n_paths; ++i) {
            if (!paths[i]) return fail(COBS_GPU_ERR_ARG, "NULL path");
            std::string err;
            files.emplace_back(new MappedFile);
            if (!files.back()->open(paths[i], err)) return fail(COBS_GPU_ERR_OPEN, err);
            Part pt;
            if (!parse_index_header(files.back()->data(), files.back()->size(), pt.meta, err))
                return fail(COBS_GPU_ERR_FORMAT, std::string("Could not open index path \"") + paths[i] + "\": " + err);
            ix->parts.push_back(std::move(pt));
        }
        cobs_gpu_status st = read_options(opts, ix.get());
        if (st != COBS_GPU_OK) return st;
        st = plan_index(ix.get());
        if (st != COBS_GPU_OK) return st;
        st = select_device(opts, &ix->device);
        if (st != COBS_GPU_OK) return st;
        st = stage_index(ix.get(), files);
        if (st != COBS_GPU_OK) return st;
        *out = ix.release();
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_open_synthetic(const cobs_gpu_synth* d, const cobs_gpu_options* opts,
                                        cobs_gpu_index** out) {
    if (!out) return fail(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!d || !d->signature_sizes || d->num_pages == 0 || d->kind > 1)
        return fail(COBS_GPU_ERR_ARG, "bad synthetic index description");
    if (d->kind == 1 && d->page_size == 0) return fail(COBS_GPU_ERR_ARG, "page_size is zero");
    if (d->kind == 0 && d->num_pages != 1) return fail(COBS_GPU_ERR_ARG, "classic index has one sub-index");
    if (d->kind == 1 && d->num_docs > (uint64_t)d->num_pages * 8 * d->page_size)
        return fail(COBS_GPU_ERR_ARG, "more documents than sub-index slots");
    if (d->num_docs == 0 || d->num_docs > 0xFFFFFFF0ull) return fail(COBS_GPU_ERR_ARG, "bad num_docs");
    return guarded([&]() -> cobs_gpu_status {
        std::unique_ptr<cobs_gpu_index> ix(new cobs_gpu_index);
        Part pt;
        pt.meta.kind = d->kind ? IndexKind::Compact : IndexKind::Classic;
        pt.meta.term_size = d->term_size;
        pt.meta.canonicalize = (uint8_t)d->canonicalize;
        pt.meta.num_hashes = d->num_hashes;
        pt.meta.header_page_size = d->kind ? d->page_size : 0;
        pt.meta.signature_sizes.assign(d->signature_sizes, d->signature_sizes + d->num_pages);
        {   // geometry first: an absurd size must not cost a multi-gigabyte name table
            IndexMeta probe = pt.meta;
            if (!d->kind) probe.doc_names.resize(1);
            const uint64_t prb = d->kind ? d->page_size : (d->num_docs + 7) / 8;
            if (prb == 0 || prb > (1ull << 28)) return fail(COBS_GPU_ERR_UNSUPPORTED, "row too wide");
            for (uint64_t sg : probe.signature_sizes) {
                uint64_t bytes = 0;
                if (sg == 0 || sg > (1ull << 46) || __builtin_mul_overflow(sg + 1, round_up(prb, 128), &bytes) ||
                    bytes > (1ull << 47))
                    return fail(COBS_GPU_ERR_UNSUPPORTED, "index geometry too large (a sub-index beyond 128 TiB)");
            }
        }
        pt.meta.doc_names.resize(d->num_docs);
        char nm[32];
        for (uint64_t i = 0; i < d->num_docs; ++i) {      // names as classic_construct_random, classic_index.cpp:668-670
            std::snprintf(nm, sizeof nm, "file_%06u", (unsigned)i);
            pt.meta.doc_names[i] = nm;
        }
        pt.synthetic = true;
        pt.synth_seed = d->seed;
        ix->parts.push_back(std::move(pt));
        cobs_gpu_status st = read_options(opts, ix.get());
        if (st != COBS_GPU_OK) return st;
        st = plan_index(ix.get());
        if (st != COBS_GPU_OK) return st;
        st = select_device(opts, &ix->device);
        if (st != COBS_GPU_OK) return st;
        std::vector<std::unique_ptr<MappedFile>> none;
        st = stage_index(ix.get(), none);
        if (st != COBS_GPU_OK) return st;
        *out = ix.release();
        return COBS_GPU_OK;
    });
}

void cobs_gpu_close(cobs_gpu_index* ix) { delete ix; }

}  // extern "C"

// An all-zero resident index of the given geometry (unsharded, no budget): index construction
// fills its rows in place (build.cpp), so that build -> query needs no file.
cobs_gpu_status cobs_amd::open_zeroed(IndexMeta&& meta, const cobs_gpu_options* opts, cobs_gpu_index** out) {
    return guarded([&]() -> cobs_gpu_status {
        std::unique_ptr<cobs_gpu_index> ix(new cobs_gpu_index);
        Part pt;
        pt.meta = std::move(meta);
        pt.built = true;
        ix->parts.push_back(std::move(pt));
        cobs_gpu_status st = read_options(opts, ix.get());
        if (st != COBS_GPU_OK) return st;
        if (ix->shard_count > 1 || ix->hbm_budget)
            return fail(COBS_GPU_ERR_UNSUPPORTED, "an index is built into HBM whole: no shard, no budget");
        st = plan_index(ix.get());
        if (st != COBS_GPU_OK) return st;
        st = select_device(opts, &ix->device);
        if (st != COBS_GPU_OK) return st;
        std::vector<std::unique_ptr<MappedFile>> none;
        st = stage_index(ix.get(), none);
        if (st != COBS_GPU_OK) return st;
        *out = ix.release();
        return COBS_GPU_OK;
    });
}

extern "C" {

cobs_gpu_status cobs_gpu_plan_shards(const char* path, uint32_t shard_count, uint32_t shard_mode,
                                     uint64_t* slot_begin, uint64_t* slot_count, uint64_t* bytes) {
    if (!path || !slot_begin || !slot_count || shard_count == 0 || shard_mode > 2)
        return fail(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        MappedFile file;
        std::string err;
        if (!file.open(path, err)) return fail(COBS_GPU_ERR_OPEN, err);
        cobs_gpu_index ix;
        ix.tune = Tuning::from_env();
        ix.shard_count = shard_count;
        ix.shard_mode = shard_mode;
        for (uint32_t r = 0; r < shard_count; ++r) {
            Part pt;
            if (!parse_index_header(file.data(), file.size(), pt.meta, err))
                return fail(COBS_GPU_ERR_FORMAT, std::string("Could not open index path \"") + path + "\": " + err);
            ix.shard_rank = r;
            cobs_gpu_status st = plan_part(pt, &ix);
            if (st != COBS_GPU_OK) return st;
            slot_begin[r] = pt.slot_begin;
            slot_count[r] = pt.slot_count;
            if (bytes) bytes[r] = pt.resident_bytes;
        }
        return COBS_GPU_OK;
    });
}

// The out-of-core plan of one index file as host arithmetic (no device): what cobs_gpu_open would keep resident and
// what it would stream under `hbm_budget_bytes` (shard options as in cobs_gpu_open; COBS_GPU_* tuning from the
// environment).  out as cobs_gpu_stream_plan; resident[i] (optional, `cap` entries) = 1 if held slice i stays in HBM.
cobs_gpu_status cobs_gpu_plan_stream(const char* path, uint64_t hbm_budget_bytes, uint32_t shard_rank, uint32_t shard_count,
                                     uint32_t shard_mode, uint64_t out[4], uint8_t* resident, size_t cap, size_t* n_slices) {
    if (!path || !out || shard_count == 0 || shard_rank >= shard_count || shard_mode > 2) return fail(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        MappedFile file;
        std::string err;
        if (!file.open(path, err)) return fail(COBS_GPU_ERR_OPEN, err);
        cobs_gpu_index ix;
        ix.tune = Tuning::from_env();
        ix.shard_rank = shard_rank;
        ix.shard_count = shard_count;
        ix.shard_mode = shard_mode;
        ix.hbm_budget = hbm_budget_bytes;
        ix.parts.emplace_back();
        Part& pt = ix.parts.back();
        if (!parse_index_header(file.data(), file.size(), pt.meta, err))
            return fail(COBS_GPU_ERR_FORMAT, std::string("Could not open index path \"") + path + "\": " + err);
        cobs_gpu_status st = plan_index(&ix);
        if (st != COBS_GPU_OK) return st;
        uint64_t chunks = 0;
        std::vector<uint8_t> keep(pt.held.size(), pt.streamed ? 0 : 1);
        for (const Chunk& c : pt.chunks) {
            chunks += (pt.streamed && !c.resident) ? 1 : 0;
            if (!c.resident) continue;
            for (const VPage& v : c.vp)            // (a resident chunk of a streamed file holds whole slices)
                for (size_t j = 0; j < pt.held.size(); ++j)
                    if (pt.held[j].fp == v.fp && pt.held[j].col0 == v.col0 && pt.held[j].ncols == v.ncols) keep[j] = 1;
        }
        out[0] = ix.stream.cap;
        out[1] = pt.streamed ? ix.stream.resident_bytes : pt.resident_bytes;
        out[2] = ix.stream.pass_bytes;
        out[3] = chunks;
        if (n_slices) *n_slices = pt.held.size();
        if (resident)
            for (size_t i = 0; i < std::min(cap, keep.size()); ++i) resident[i] = keep[i];
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_set_tuning(cobs_gpu_index* ix, const char* key, int64_t value) {
    if (!ix || !key) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    const std::string k = key;
    Tuning& t = ix->tune;
    if (k == "waves") {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(COBS_GPU_ERR_ARG, "waves: 0, 1, 2 or 4");
        t.waves = (uint32_t)value;
    } else if (k == "tile_w") {
        if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64)
            return fail(COBS_GPU_ERR_ARG, "tile_w: 0, 4, 8, 16, 32 or 64");
        t.tile_w = (uint32_t)value;
    } else if (k == "mq") {
        t.mq = value < 0 ? -1 : value != 0;
    } else if (k == "pass_bytes") {
        t.pass_bytes = value > 0 ? (uint64_t)value : 16ull << 30;
    } else if (k == "pipe_chars") {
        t.pipe_chars = value < 0 ? 4ull << 20 : (uint64_t)value;
    } else if (k == "graph") {
        t.graph = value < 0 ? -1 : value != 0;
    } else if (k == "lds_staged") {
        t.lds_staged = value > 0;
    } else if (k == "device_rank") {
        t.device_rank = value != 0;
    } else if (k == "rank_window_kib") {
        t.rank_window_kib = value > 0 ? (uint32_t)std::min<int64_t>(value, 1 << 20) : 16u << 10;
    } else if (k == "row_ranges") {
        return fail(COBS_GPU_ERR_ARG, "row_ranges shapes the chunks of an index: set COBS_GPU_ROW_RANGES before it is opened");
    } else if (k == "compact_terms") {
        t.compact_terms = value != 0;
    } else if (k == "rank_pack") {
        t.rank_pack = value != 0;
    } else if (k == "rank_slim") {
        t.rank_slim = value != 0;
    } else if (k == "rank_segments") {
        if (value < 0 || value > 64) return fail(COBS_GPU_ERR_ARG, "rank_segments: 0 (by row length) .. 64");
        t.rank_segments = (int)value;
    } else if (k == "hash_stream") {
        t.hash_stream = value != 0;
    } else if (k == "tile_topk") {
        t.tile_topk = value != 0;
    } else if (k == "min_score_bytes") {
        if (value != 0 && value != 1 && value != 2 && value != 4) return fail(COBS_GPU_ERR_ARG, "min_score_bytes: 0, 1, 2 or 4");
        t.min_score_bytes = (uint32_t)value;
    } else if (k == "exp") {
        t.exp = (uint32_t)value;
    } else if (k == "row_fetch") {
        t.row_fetch = value != 0;
    } else if (k == "row_fetch_alpha") {
        t.row_fetch_alpha = value >= 0 ? (uint32_t)std::min<int64_t>(value, 1 << 20) : 1;   // 0: whenever the rows fit
    } else if (k == "phase_slots") {
        t.phase_slots = value > 0 ? (uint32_t)std::min<int64_t>(value, 1 << 20) : 0;
    } else {
        return fail(COBS_GPU_ERR_ARG, "unknown tuning key (waves, tile_w, mq, pass_bytes, pipe_chars, graph, lds_staged, device_rank, rank_pack, rank_slim, rank_segments, rank_window_kib, hash_stream, tile_topk, row_fetch, row_fetch_alpha, min_score_bytes)");
    }
    return COBS_GPU_OK;
}

size_t cobs_gpu_num_files(const cobs_gpu_index* ix) { return ix ? ix->parts.size() : 0; }

cobs_gpu_status cobs_gpu_info(const cobs_gpu_index* ix, size_t f, cobs_gpu_index_info* o) {
    if (!ix || !o || f >= ix->parts.size()) return fail(COBS_GPU_ERR_ARG, "bad file number");
    const Part& p = ix->parts[f];
    std::memset(o, 0, sizeof *o);
    o->kind = (uint32_t)p.meta.kind;
    o->term_size = p.meta.term_size;
    o->canonicalize = p.meta.canonicalize;
    o->num_pages = p.meta.num_pages();
    o->num_hashes = p.meta.num_hashes;
    o->page_size = p.meta.page_size();
    o->row_size = p.meta.row_size();
    o->counts_size = p.meta.counts_size();
    o->num_docs = p.meta.doc_names.size();
    o->doc_offset = p.doc_offset;
    o->hbm_bytes = p.hbm_bytes;
    o->first_page = p.first_page;
    o->end_page = p.end_page;
    o->slot_begin = p.slot_begin;
    o->slot_count = p.slot_count;
    o->local_offset = p.local_offset;
    return COBS_GPU_OK;
}

uint64_t cobs_gpu_signature_size(const cobs_gpu_index* ix, size_t f, uint32_t page) {
    if (!ix || f >= ix->parts.size() || page >= ix->parts[f].meta.num_pages()) return 0;
    return ix->parts[f].meta.signature_sizes[page];
}

const char* cobs_gpu_doc_name(const cobs_gpu_index* ix, size_t f, uint64_t doc) {
    if (!ix || f >= ix->parts.size() || doc >= ix->parts[f].meta.doc_names.size()) return "";
    return ix->parts[f].meta.doc_names[doc].c_str();
}

uint64_t cobs_gpu_total_counts(const cobs_gpu_index* ix) { return ix ? ix->total_counts : 0; }
uint64_t cobs_gpu_local_counts(const cobs_gpu_index* ix) { return ix ? ix->local_counts : 0; }

// resident slice of file-level sub-index `page` (streamed chunks cannot be read back)
static cobs_gpu_status find_resident(const cobs_gpu_index* ix, size_t f, uint32_t page, const Chunk** c,
                                     const PageDev** pd) {
    if (!ix || f >= ix->parts.size()) return fail(COBS_GPU_ERR_ARG, "bad argument");
    const Part& p = ix->parts[f];
    if (p.streamed) return fail(COBS_GPU_ERR_UNSUPPORTED, "index is streamed, rows are not resident");
    for (const Chunk& ch : p.chunks)
        for (size_t i = 0; i < ch.vp.size(); ++i)
            if (ch.vp[i].fp == page) {
                *c = &ch;
                *pd = &ch.pages[i];
                return COBS_GPU_OK;
            }
    return fail(COBS_GPU_ERR_ARG, "sub-index not held by this shard");
}

cobs_gpu_status cobs_gpu_page_columns(const cobs_gpu_index* ix, size_t f, uint32_t page, uint64_t* col0,
                                      uint64_t* ncols) {
    if (!ix || f >= ix->parts.size() || !col0 || !ncols) return fail(COBS_GPU_ERR_ARG, "bad argument");
    *col0 = *ncols = 0;
    for (const VPage& v : ix->parts[f].held)
        if (v.fp == page) {
            *col0 = v.col0;
            *ncols = v.ncols;
            return COBS_GPU_OK;
        }
    return COBS_GPU_OK;      // not held: zero columns
}

cobs_gpu_status cobs_gpu_read_row(const cobs_gpu_index* ix, size_t f, uint32_t page, uint64_t row,
                                  uint8_t* out, size_t n) {
    const Chunk* c = nullptr;
    const PageDev* pd = nullptr;
    if (!out) return fail(COBS_GPU_ERR_ARG, "bad argument");
    cobs_gpu_status st = find_resident(ix, f, page, &c, &pd);
    if (st != COBS_GPU_OK) return st;
    if (row > pd->sig || n > c->pitch) return fail(COBS_GPU_ERR_ARG, "row or length out of range");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipMemcpy(out, c->d_data + pd->base + row * (uint64_t)c->pitch, n, hipMemcpyDeviceToHost));
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_read_rows(const cobs_gpu_index* ix, size_t f, uint32_t page, uint64_t row0,
                                   uint64_t nrows, uint8_t* out, size_t out_pitch) {
    const Chunk* c = nullptr;
    const PageDev* pd = nullptr;
    if (!out) return fail(COBS_GPU_ERR_ARG, "bad argument");
    cobs_gpu_status st = find_resident(ix, f, page, &c, &pd);
    if (st != COBS_GPU_OK) return st;
    if (row0 + nrows > pd->sig + 1 || out_pitch < pd->valid_bytes) return fail(COBS_GPU_ERR_ARG, "rows or pitch out of range");
    HIP_TRY(hipSetDevice(ix->device));
    if (nrows)
        HIP_TRY(hipMemcpy2D(out, out_pitch, c->d_data + pd->base + row0 * (uint64_t)c->pitch, c->pitch,
                            (size_t)pd->valid_bytes, (size_t)nrows, hipMemcpyDeviceToHost));
    return COBS_GPU_OK;
}

uint64_t cobs_gpu_graph_replays(const cobs_gpu_index* ix) { return ix ? ix->graph_replays : 0; }
uint64_t cobs_gpu_host_passes(const cobs_gpu_index* ix) { return ix ? ix->host_passes : 0; }

// True positives for the procedural index: see include/cobs_gpu_batch.h.
cobs_gpu_status cobs_gpu_plant(cobs_gpu_index* ix, size_t f, const char* text, size_t len, const uint32_t* docs,
                               const uint32_t* keep_permille, size_t ndocs, uint64_t salt) {
    if (!ix || f >= ix->parts.size() || (len && !text) || (ndocs && (!docs || !keep_permille)))
        return fail(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        Part& p = ix->parts[f];
        if (p.streamed) return fail(COBS_GPU_ERR_UNSUPPORTED, "the index is streamed: its rows are not resident");
        if (len > 0xFFFFFFF0ull || ndocs > (1u << 20)) return fail(COBS_GPU_ERR_ARG, "text or document list too long");
        if (len < p.meta.term_size || ndocs == 0) return COBS_GPU_OK;
        HIP_TRY(hipSetDevice(ix->device));
        const bool compact = p.meta.kind == IndexKind::Compact;
        const uint64_t page_docs = compact ? 8 * p.meta.header_page_size : ~0ull;
        std::vector<PlantDoc> h(ndocs);
        for (size_t i = 0; i < ndocs; ++i) {
            const uint64_t d = docs[i];
            if (d >= p.meta.doc_names.size()) return fail(COBS_GPU_ERR_ARG, "document " + std::to_string(d) + " does not exist");
            if (keep_permille[i] > 1000) return fail(COBS_GPU_ERR_ARG, "keep_permille beyond 1000");
            const uint32_t fp = compact ? (uint32_t)(d / page_docs) : 0u;
            const uint64_t byte = (compact ? d - (uint64_t)fp * page_docs : d) / 8;
            PlantDoc pd{};
            pd.col = nullptr;
            pd.doc = (uint32_t)d;
            pd.bit = (uint32_t)(d & 7u);
            pd.keep_permille = keep_permille[i];
            for (const Chunk& ch : p.chunks)
                for (size_t v = 0; v < ch.vp.size() && !pd.col; ++v)
                    if (ch.vp[v].fp == fp && byte >= ch.vp[v].col0 && byte < ch.vp[v].col0 + ch.vp[v].ncols && ch.d_data) {
                        pd.col = ch.d_data + ch.pages[v].base + (byte - ch.vp[v].col0);
                        pd.sig = ch.pages[v].sig;
                        pd.pitch = ch.pitch;
                    }
            h[i] = pd;      // (a document of another shard: col stays NULL, the rank that holds it plants it)
        }
        DevBuf<uint8_t> d_text;
        DevBuf<PlantDoc> d_docs;
        DevBuf<uint32_t> d_bad;
        HIP_TRY(d_text.reserve(len));
        HIP_TRY(d_docs.reserve(ndocs));
        HIP_TRY(d_bad.reserve(1));
        HIP_TRY(hipMemcpy(d_text.p, text, len, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_docs.p, h.data(), ndocs * sizeof(PlantDoc), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_bad.p, 0, 4));
        PlantArgs a{};
        a.text = d_text.p;
        a.docs = d_docs.p;
        a.salt = salt;
        a.len = (uint32_t)len;
        a.term_size = p.meta.term_size;
        a.canonicalize = p.meta.canonicalize;
        a.num_hashes = (uint32_t)p.meta.num_hashes;
        a.ndocs = (uint32_t)ndocs;
        a.bad = d_bad.p;
        HIP_TRY(launch_plant(a, nullptr));
        uint32_t bad = 0;
        HIP_TRY(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
        if (bad) return fail(COBS_GPU_ERR_INVALID_BASE, "the planted text holds a character outside ACGT");
        return COBS_GPU_OK;
    });
}

cobs_gpu_status cobs_gpu_stream_counters(const cobs_gpu_index* ix, uint64_t out[2]) {
    if (!ix || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    out[0] = ix->stream.fetched_chunks;
    out[1] = ix->stream.streamed_chunks;
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_stream_traffic(const cobs_gpu_index* ix, uint64_t out[4]) {
    if (!ix || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    out[0] = ix->stream.fetched_chunks;
    out[1] = ix->stream.streamed_chunks;
    // the fetched units' bytes: distinct looked-up rows x pitch, counted by the gather itself (a row looked up by several
    // terms of a batch crosses PCIe once since round 6)
    out[2] = 0;
    if (ix->stream.d_fetched.p) {
        unsigned long long v = 0;
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&v, ix->stream.d_fetched.p, sizeof v, hipMemcpyDeviceToHost));
        out[2] = v;
    }
    out[3] = ix->stream.streamed_bytes;
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_stream_plan(const cobs_gpu_index* ix, uint64_t out[4]) {
    if (!ix || !out) return fail(COBS_GPU_ERR_ARG, "NULL argument");
    uint64_t chunks = 0;
    for (const Part& p : ix->parts)
        if (p.streamed)
            for (const Chunk& c : p.chunks) chunks += c.resident ? 0 : 1;
    out[0] = ix->stream.cap;
    out[1] = ix->stream.resident_bytes;
    out[2] = ix->stream.pass_bytes;
    out[3] = chunks;
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_timers(cobs_gpu_index* ix, double out[5], int reset) {
    if (!ix) return fail(COBS_GPU_ERR_ARG, "NULL index");
    if (out) std::memcpy(out, ix->timers, sizeof ix->timers);
    if (reset) std::memset(ix->timers, 0, sizeof ix->timers);
    return COBS_GPU_OK;
}

}  // extern "C"

