// cobs_amd/csrc/engine.hpp -- internal types of libcobs_gpu.so shared by engine.cpp (index
// staging, batches, the search API), comm.cpp (RCCL exchange of the sharded layout) and
// build.cpp.  Nothing here is part of the C ABI (include/cobs_gpu.h, cobs_gpu_batch.h, cobs_gpu_diag.h).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/cobs_gpu_diag.h"      // (includes cobs_gpu_batch.h and cobs_gpu.h: the whole C ABI)
#include "device_types.hpp"
#include "index_file.hpp"
#include "kernels.hpp"

namespace cobs_amd {

// ---------------------------------------------------------------------------
// errors (thread-local text behind cobs_gpu_last_error)

cobs_gpu_status fail(cobs_gpu_status st, const std::string& msg);
cobs_gpu_status hip_fail(hipError_t e, const char* what);

#define HIP_TRY(expr)                                                  \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return ::cobs_amd::hip_fail(_e, #expr);  \
    } while (0)

// No C++ exception may cross the C ABI (an untrusted header can make a std::vector throw):
// every entry point that allocates runs its body through this.
template <typename F>
cobs_gpu_status guarded(F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(COBS_GPU_ERR_CAPACITY, "out of host memory");
    } catch (const std::exception& e) {
        return fail(COBS_GPU_ERR_ARG, std::string("unexpected exception: ") + e.what());
    } catch (...) {
        return fail(COBS_GPU_ERR_ARG, "unexpected exception");
    }
}

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }
double now_s();

template <typename T>
struct DevBuf {     // grow-only device allocation
    T* p = nullptr;
    size_t cap = 0;   // elements
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
};

template <typename T>
struct PinnedBuf {  // grow-only pinned host staging
    T* p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    PinnedBuf(PinnedBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipHostMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
};

// Tuning hooks.  The COBS_GPU_* environment variables are read ONCE, when an index is opened,
// into the handle (never on the launch path); cobs_gpu_set_tuning changes them per handle
// afterwards (the A/B scripts use that).  0 / -1 = automatic.
struct Tuning {
    uint32_t row_align = 0;     // COBS_GPU_ROW_ALIGN: row pitch alignment (16..4096, multiple of 16)
    uint32_t waves = 0;         // COBS_GPU_WAVES: waves per work-group (1, 2, 4)
    uint32_t tile_w = 0;        // COBS_GPU_TILE_W: 16-byte chunks per tile (4..64)
    int mq = -1;                // COBS_GPU_MQ: force the multi-query scan variant off / on
    uint64_t pass_bytes = 16ull << 30;   // COBS_GPU_PASS_BYTES: workspace limit of one device pass
    uint64_t pipe_chars = 4ull << 20;    // COBS_GPU_PIPE_CHARS: query text from which a call is pipelined
    bool no_pin = false;        // COBS_GPU_NO_PIN: never hipHostRegister the mapped file
    int graph = -1;             // COBS_GPU_GRAPH: captured-graph path for small batches off / on
    bool lds_staged = false;    // COBS_GPU_LDS_STAGED: the LDS-staged scan variant (A/B measurements only)
    uint32_t min_score_bytes = 0;   // 2 / 4: scores at least that wide (the reference's classic_search_disable_8bit / _16bit)
    int row_fetch = 1;          // streamed chunks are fetched row by row when a batch looks up few of their rows (0: always whole)
    uint32_t row_fetch_alpha = 1;   // ... i.e. when alpha x (looked-up bytes) <= the chunk's bytes.  Measured on MI355X: rows of 1568 bytes
                                    // fetched by the kernel cross PCIe at 50.8 GB/s, whole chunks at 52.3 GB/s (profiles/r03_c5_selective.txt): 1
    int tile_topk = 1;          // top-k passes without score rows select per tile in K2 (0: score rows + K3, A/B)
    int device_rank = 1;        // whole score rows are ranked on the device (0: by host threads, A/B and fallback)
    uint32_t rank_window_kib = 16u << 10;   // device-ranked records cross PCIe in pieces of this size (KiB)
    uint32_t stream_buf_kib = 256u << 10;   // COBS_GPU_STREAM_BUF_KIB: upper bound of ONE of the two stream buffers of an out-of-core handle;
                                            // what the budget leaves beyond them keeps slices of the streamed files resident (0: no bound)
    uint32_t row_range_min = 1024;  // ... unless a buffer holds fewer rows than this (then by columns); COBS_GPU_ROW_RANGE_MIN, tests
    uint32_t packed_width = 0;  // (set by the planner for a streamed part: chunks of this many columns keep that pitch)
    int stream_packed = 1;      // streamed chunks keep the file's row pitch (linear PCIe copies); 0: device pitch, 2-D copies (A/B)
    int row_ranges = 1;         // a streamed sub-index larger than a stream buffer is cut by ROWS (H = 1; 0: by columns, A/B and fallback)
    int compact_terms = 1;      // the scan of a row-range unit walks the terms the unit HOLDS (a compact second table), not every term (0: A/B)
    int rank_segments = 0;      // segments a single-pass ranking cuts a row into (0: by row length, 1: one work-group per query)
    int rank_slim = 1;          // full lists of the default call cross PCIe as slot streams + score counts (0: records, A/B)
    int rank_pack = 1;          // device-ranked records cross PCIe as one u32 where slot and score fit (0: always 8-byte pairs, A/B)
    int hash_stream = 0;        // K1 of a device-resident batch runs on the batch's own stream, K2 waits for it by event: the hashing
                                // of one (sub-)batch overlaps the scan / exchange of another (the sharded multi-GPU flow)
    uint32_t exp = 0;           // experimental kernel variants under A/B measurement (bit field, ScanArgs::exp)
    bool trace = false;         // COBS_GPU_TRACE: where the host side of a search call spends its time, on stderr
    uint32_t phase_slots = 0;   // tuning builds (make timing): work-groups of a scan launch that record phase stamps
    static Tuning from_env();
};

// A slice of one file-level sub-index: all of its rows, row bytes [col0, col0+ncols).
struct VPage {
    uint32_t fp = 0;
    uint64_t col0 = 0, ncols = 0;
    uint64_t row0 = 0, nrows = 0;    // nrows != 0: only rows [row0, row0 + nrows) (a row-range chunk of a streamed sub-index)
};

// A group of slices (equal width) that is in HBM at the same time and is scanned by one
// launch.  A resident part is one chunk per slice width (a shard cut inside a sub-index holds
// up to three: tail columns of its first sub-index, whole sub-indexes, head columns of its
// last); a part larger than the HBM budget is cut into chunks that are streamed through two
// device buffers, one scan pass per chunk (documents of different sub-indexes / column ranges
// never combine, so every chunk is an independent scan that fills its own score slots).
struct Chunk {
    std::vector<VPage> vp;
    std::vector<PageDev> pages;      // bases relative to the chunk's buffer
    PageDev* d_pages = nullptr;
    uint32_t pitch = 0, cpp = 0, total_chunks = 0;
    size_t bytes = 0;                // device bytes incl. zero rows
    size_t stage_bytes = 0;          // packed host bytes (rows x ncols)
    uint8_t* d_data = nullptr;       // resident chunk only
    // streamed chunk of a file whose mapping is registered: what the row-selective pass needs
    std::vector<uint64_t> src;       // [vp.size()] file offset of (first row, first held column) of every slice
    std::vector<std::pair<uint32_t, uint32_t>> cp;   // [vp.size()] (first, count) of the part's look-up counters that cover the page (count_rows_kernel)
    bool fetch_ok = false;           // ... and it may be fetched row by row (registered mapping, not resident)
    PageDev* d_pages2[2] = {nullptr, nullptr};   // the pages as a gathered buffer holds them (written per pass), per stream buffer
    // A ROW-RANGE chunk: one sub-index too large for a stream buffer, cut by rows (all columns, `nrows` rows from `row0`) --
    // whole rows cross PCIe at the link's best rate, column slices do not (plan.cpp: chunk_part).  Its scan counts only
    // the terms whose row falls into the range; every range but the first writes partial scores that are ADDED to the rows.
    bool resident = false;           // a chunk of a STREAMED part that stays in HBM (d_data): the budget had room for it beside the stream buffers
    bool row_range = false;
    uint32_t range_no = 0;           // 0 = the first range of its sub-index (writes the scores), > 0 = adds to them
    PageDev* d_pages_acc = nullptr;  // range_no > 0: the page with slot0 = 0 (partial scores go to a scratch matrix)
    uint32_t first_chunk = 0;        // a fetch group (Part::fetch_groups): index of the first chunk of its run in Part::chunks
};

// One index file as held by this device (possibly only a shard of it).
struct Part {
    IndexMeta meta;
    uint32_t first_page = 0, end_page = 0;   // file-level sub-indexes (partly) held here
    std::vector<VPage> held;                 // the slices of those sub-indexes this shard holds, in slot order
    std::vector<Chunk> chunks;
    // streamed file whose mapping is registered: the chunks of equal row pitch merged -- when a (small) batch fetches
    // every chunk of the file by rows, one fetch + one scan per pitch does the pass instead of one pair per chunk
    std::vector<Chunk> fetch_groups;
    std::vector<PageDev> tpages;             // sub-indexes [first_page, end_page): pages of the row-index table
    PageDev* d_tpages = nullptr;
    // streamed file with a registered mapping: one look-up counter per streamed piece of every sub-index (a whole slice,
    // or each of its row ranges) -- a pass counts, right after K1, how many rows the batch looks up in each (pass.cpp)
    std::vector<CountPage> cpages;           // [tpages]
    CountPage* d_cpages = nullptr;
    uint32_t ncounters = 0;
    bool streamed = false;
    bool has_row_ranges = false;             // some chunk is a row range: K2 cannot select on its partial counts (pass.cpp)
    bool idx64 = false;                      // a sub-index has >= 2^32 - 1 rows: 64-bit row-index table
    size_t hbm_bytes = 0;
    uint64_t resident_bytes = 0;             // what the held slices need when they stay in HBM
    // streaming (BASELINE config 5: index larger than the HBM budget)
    std::unique_ptr<MappedFile> file;        // source of the chunks
    bool file_pinned = false;                // the held pages of the mapping are registered with HIP: DMA straight from them
    uint8_t* pin_base = nullptr;             // ... first registered byte (for hipHostUnregister)
    const uint8_t* file_dev = nullptr;       // ... and this is its device-visible address (kernels read it over PCIe)
    bool synthetic = false;
    bool built = false;                      // rows are produced in place by index construction
    uint64_t synth_seed = 0;
    uint64_t doc_offset = 0;      // first global score slot of this file
    uint64_t slot_begin = 0;      // file-level score slots computed here
    uint64_t slot_count = 0;
    uint64_t local_offset = 0;    // position of those slots in a local count row

    uint32_t num_tpages() const { return end_page - first_page; }
    Part() = default;
    Part(Part&&) = default;
    Part(const Part&) = delete;
    ~Part();
};

// The two device buffers all streamed parts of a handle share (one pass at a time walks the
// parts in order, so the double buffering simply continues across files), with the copy
// stream and the events that order copies against scans.
struct StreamBufs {
    DevBuf<uint8_t> sbuf[2];
    PinnedBuf<uint8_t> stage[2];
    hipStream_t copy_stream = nullptr;
    // the slot assignment of a row-selective gather runs beside the previous unit's copy (that kernel waits for PCIe, this one
    // walks the row-index table): its own stream, and the event the copy of the same buffer waits for
    hipStream_t prep_stream = nullptr;
    hipEvent_t assigned[2] = {nullptr, nullptr};
    hipEvent_t copied[2] = {nullptr, nullptr}, scanned[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    uint64_t cap = 0;             // bytes per buffer
    // row-selective passes: the row-index table of the gathered rows in buffer i, and the event after K1
    // (the fetch kernel on the copy stream reads K1's table)
    DevBuf<uint8_t> table2[2];
    DevBuf<uint32_t> table3[2];               // a row-range unit's in-range terms per query (compact table) ...
    DevBuf<uint64_t> blk2[2];                 // ... its block offsets ...
    DevBuf<uint32_t> blkcnt[2];               // ... and the per-query block counts they are scanned from
    DevBuf<uint64_t> rowlist[2];              // source rows of the gathered rows in buffer i
    DevBuf<unsigned long long> cursor[2];     // distinct looked-up rows of the gather's pages
    DevBuf<uint32_t> bitmap[2], bprefix[2];   // row bitmaps of a unit's leader pages and the set bits in front of every word
    DevBuf<unsigned long long> d_fetched;     // bytes the gathers of all passes so far asked of PCIe (distinct rows x pitch)
    DevBuf<GatherPage> gpages[2];
    PinnedBuf<GatherPage> h_gpages[2];
    DevBuf<unsigned long long> d_counts;      // look-up counters of the file being scanned (count_rows_kernel)
    PinnedBuf<unsigned long long> h_counts;
    hipEvent_t hashed = nullptr;
    uint64_t fetched_chunks = 0, streamed_chunks = 0;   // diagnostics: how the chunks of all passes were brought in
    uint64_t lookup_bytes = 0, streamed_bytes = 0;      // ... looked-up rows x pitch of the fetched units (with repeats) / the rows of the chunks copied whole
    uint64_t resident_bytes = 0;  // the plan: bytes of the streamed files' chunks that stay resident ...
    uint64_t pass_bytes = 0;      // ... and row bytes a pass that copies every other chunk whole moves over PCIe
    size_t stage_need = 0;
    uint64_t seq = 0;             // chunks streamed so far: chunk goes to buffer seq % 2
    ~StreamBufs();
};

struct PartWork {    // per-file device workspace of a batch
    const uint64_t* blk_off = nullptr;   // inside the batch's upload buffer
    DevBuf<uint32_t> table;
    DevBuf<uint32_t> thr;
    std::vector<uint64_t> h_blk_off;
    uint64_t table_entries = 0;
};

// cobs_gpu_search_batch_view: results in memory the handle owns (anonymous mapping, huge pages where the host offers
// them on request; kept across calls, so its pages are faulted in once)
struct ResultArena {
    cobs_gpu_hit* p = nullptr;
    size_t cap = 0;                  // records
    std::vector<size_t> offs;
    cobs_gpu_status reserve(size_t n);                     // contents are not kept
    cobs_gpu_status grow_keep(size_t n, size_t used);      // the first `used` records are
    ~ResultArena();
};

struct Exchange;     // comm.cpp: buffers of the RCCL exchange bound to a batch
struct RankWork;     // rank.cpp: buffers of the on-device ranking of whole score rows

}  // namespace cobs_amd

struct cobs_gpu_index {
    int device = 0;
    uint32_t shard_rank = 0, shard_count = 1, shard_mode = 0;
    uint64_t hbm_budget = 0;      // 0 = everything resident
    uint32_t waves_per_group = 0; // 0 = by query length
    cobs_amd::Tuning tune;
    std::vector<cobs_amd::Part> parts;
    cobs_amd::StreamBufs stream;
    uint64_t total_counts = 0, local_counts = 0;
    double timers[5] = {0, 0, 0, 0, 0};
    cobs_amd::ResultArena arena;  // cobs_gpu_search_batch_view
    uint64_t host_passes = 0;     // device passes launched by the host-buffer calls
    uint64_t graph_replays = 0;   // small passes of the host API served by a captured hipGraph
    static constexpr int kScratch = 3;
    cobs_gpu_batch* scratch[kScratch] = {nullptr, nullptr, nullptr};   // workspaces of the host-buffer search API
    hipStream_t xchg_stream = nullptr;   // sharded search: the stream the ranks' agreements and exchanges of its passes run on (sharded.cpp)
    ~cobs_gpu_index();
};

struct cobs_gpu_batch {
    cobs_gpu_index* ix = nullptr;
    size_t max_queries = 0, max_len = 0;
    size_t nq = 0;
    std::vector<uint32_t> lens;
    std::vector<uint64_t> span_off;
    cobs_amd::DevBuf<uint8_t> text;
    // query text, span offsets, query lengths and the per-file block offsets live in ONE pinned
    // staging buffer / ONE device buffer (`text`): a batch is uploaded with a single async copy
    const uint64_t* d_span_off = nullptr;
    const uint32_t* d_qlen = nullptr;
    const uint8_t* d_text = nullptr;      // query characters (behind the tables in `text`)
    cobs_amd::PinnedBuf<uint8_t> h_text;
    cobs_amd::PinnedBuf<uint32_t> h_thr_stage;
    std::vector<cobs_amd::PartWork> work;
    cobs_amd::DevBuf<uint8_t> counts;
    cobs_amd::DevBuf<uint8_t> counts_part;   // partial scores of a row-range chunk (streamed index), added to `counts`
    cobs_amd::DevBuf<uint8_t> counts_acc;    // ... or, in a pass without score rows, to this: the scores of ONE sub-index, selected from after its last range
    uint32_t elem_bytes = 2;
    int planes = 0;
    uint64_t max_terms = 0;              // longest query of the batch, in terms
    cobs_amd::DevBuf<cobs_amd::HitDev> hits;
    cobs_amd::DevBuf<uint2> topk_out;    // K3 output [file][query][k], ordered (score desc, doc asc)
    cobs_amd::DevBuf<uint32_t> topk_cnt; // [file][query]
    std::vector<uint2> h_topk;
    std::vector<uint32_t> h_topk_cnt;
    uint32_t topk_k = 0;              // k of the last run (0 = K3 not run)
    bool topk_fetched = false;
    bool topk_sorted = false;         // K3 ordered the survivors of every (query, file) on the device
    bool topk_direct = false;         // K2 left every tile's k best in `cand`, K3 merged those: no score rows
    cobs_amd::DevBuf<uint2> cand;     // candidate pool [file][query][tiles x k (padded to 8)] of (document, score)
    // device flags: word 0 = first invalid query (2^32-1 - q, 0 = none), words 2..3 = 64-bit fill
    // of the hit pool (may exceed hit_cap: overflow)
    cobs_amd::DevBuf<uint32_t> flags;
    uint32_t hit_cap = 0;
    // last run
    bool ran = false, selected = false, synced = false;
    bool have_counts = false;         // the last run wrote the score rows
    double threshold = 0.0;
    uint32_t h_flags[4] = {0, 0, 0, 0};
    cobs_amd::PinnedBuf<uint32_t> h_flags_pin;   // host-buffer passes: the flag words land here right behind the pass's kernels ...
    bool flags_landing = false;                  // ... (queued: `done` follows them)
    uint64_t h_nhits() const { return (uint64_t)h_flags[3] << 32 | h_flags[2]; }
    std::vector<cobs_amd::HitDev> h_hits;       // pool copy, sorted by query
    std::vector<size_t> h_hit_off;
    bool pool_fetched = false;
    bool pool_sorted = false;         // h_hits holds every query's records in RESULT order (ordered on the device: order_pool)
    bool pool_pending = false;        // order_pool_launch queued the ordering of pool_n records and their copy home; order_pool_collect publishes them
    uint64_t pool_n = 0;
    cobs_amd::PinnedBuf<uint8_t> h_single;                  // staging of pool_single
    cobs_amd::DevBuf<uint32_t> pool_idx;                    // [3][nq + 1]: records per query | first record | scatter cursors
    cobs_amd::DevBuf<cobs_amd::HitDev> pool_tmp, pool_out;  // the pool bucketed by query / every bucket ordered
    cobs_amd::DevBuf<uint8_t> pool_single;                  // [nq] queries with a single hash in total (index order)
    cobs_amd::PinnedBuf<uint8_t> h_pool;                    // landing buffer: offsets, then the ordered records
    // host copy of a window of score rows [rows_q0, rows_q1) of the last run (raw elem_bytes)
    cobs_amd::PinnedBuf<uint8_t> h_rows;
    size_t rows_q0 = 0, rows_q1 = 0;
    std::vector<uint32_t> rank_hist;  // scratch of the counting sort in hits_host
    std::vector<cobs_gpu_hit> sel_scratch;
    // HIP events around K1 and K2 of the most recent runs (recorded on the launch stream)
    static constexpr int kRing = 64;
    hipEvent_t ev[kRing][4] = {};     // before K1 | after K1 | after the last K2 | (K1 on its own stream) before the first K2
    bool ev_split[kRing] = {};        // the run recorded ev[3]: its K2 time is ev[3]..ev[2]
    hipStream_t hash_stream = nullptr;   // tuning key hash_stream: K1's own stream ...
    hipEvent_t hashed = nullptr;         // ... and the event K2's stream waits for
    uint64_t run_seq = 0, read_seq = 0;
    uint64_t stats[4] = {0, 0, 0, 0};
    uint64_t algo_row_bytes = 0;      // gathered row bytes of the current queries (stats[0] adds the score bytes a run writes)
    hipEvent_t run_done = nullptr;    // after the last kernel of the most recent run
    // host-buffer API only: the stream this scratch batch lives on and the event after its pass
    hipStream_t own_stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t scan_end = nullptr;    // the event after the last K2 launch of the most recent run (one of its timing events)
    hipEvent_t scan_after = nullptr;  // set for one run: its K2 launches wait for this event (the previous pass), its K1 does not
    // captured graph of a small pass (single-query latency path)
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_key = 0, graph_candidate = 0;
    bool graph_run = false;           // the last run was a graph replay
    hipEvent_t graph_t0 = nullptr, graph_t1 = nullptr;   // ... bracketed by these (its device time goes to the "scan" timer as a whole)
    // results a replayed graph copies home by itself (pinned): flags | top-k counts | survivors | hit-pool prefix
    cobs_amd::PinnedBuf<uint8_t> h_res;
    size_t res_topk = 0, res_pool = 0, res_pool_n = 0;
    bool res_rows = false;            // ... and the score rows of an all-documents pass, into h_rows
    // shapes captured earlier (a service sees a handful of read lengths, not one): the current graph
    // above plus a few older ones, least recently used first to go; and the keys of shapes seen once
    // (the second sighting of a shape captures it, whatever ran in between)
    struct GraphEntry {
        hipGraphExec_t exec = nullptr;
        uint64_t key = 0, used = 0;
        size_t res_topk = 0, res_pool = 0, res_pool_n = 0;
        bool res_rows = false;
    };
    GraphEntry graph_more[3];
    uint64_t graph_recent[4] = {0, 0, 0, 0};
    uint64_t graph_clock = 0;
    hipStream_t graph_stream = nullptr;
    cobs_amd::DevBuf<uint64_t> phase;          // phase stamps of the last scan launch (tuning builds)
    cobs_amd::Exchange* xchg = nullptr;       // comm.cpp
    cobs_amd::RankWork* rank = nullptr;       // rank.cpp
    // set by an exchange, cleared by the next run: GLOBAL score rows (all shards' slices assembled
    // in global document order) of queries [g_q0, g_q0 + g_qn), owned by xchg
    const uint8_t* g_rows = nullptr;
    uint64_t g_q0 = 0, g_qn = 0;
    bool view_global = false;
    bool pool_global = false;         // h_hits holds the hit pools of ALL shards
    bool pool_owned = false;          // ... of the queries [own_q0, own_q0 + own_qn) only (owner-routed exchange)
    uint64_t own_q0 = 0, own_qn = 0;
    uint32_t topk_stride = 0;         // entries per (file, query) in h_topk (0 = topk_k; ranks * k after an exchange)
    ~cobs_gpu_batch();
};

namespace cobs_amd {

// ---- plan.cpp: host arithmetic of an opened index
cobs_gpu_status select_device(const cobs_gpu_options* o, int* device);
uint32_t pitch_for(uint64_t ncols, const Tuning& tune);
uint64_t slice_bytes(uint64_t sig, uint64_t ncols, const Tuning& tune);
// geometry of a scan launch: tile width (16-byte column chunks), waves per work-group, work-group variant
struct ScanGeom { uint32_t tile_w; int nwaves; bool multi_query; };
ScanGeom scan_geometry(const Chunk& c, uint64_t mean_blocks, uint64_t max_blocks, uint64_t num_hashes,
                       uint32_t forced_waves, int planes, bool idx64, const Tuning& tune);
std::vector<VPage> held_slices(const IndexMeta& m, uint32_t rank, uint32_t count, uint32_t mode);
cobs_gpu_status plan_part(Part& pt, const cobs_gpu_index* ix);
cobs_gpu_status chunk_part(Part& pt, uint64_t cap, const Tuning& tune, uint64_t keep_bytes = 0, uint64_t* kept = nullptr);
cobs_gpu_status plan_index(cobs_gpu_index* ix);
// a replayed small pass brings this many hit-pool entries home inside the graph
constexpr size_t kGraphPoolPrefix = 2048;
// K3 orders up to this many survivors per (query, file) on the device (8-byte keys in 64 KB of LDS)
constexpr size_t kTopkSortLimit = 8192;
// largest k for which K2 selects per tile (a tile holds 512 or more documents; the pool is queries x tiles x k entries)
constexpr size_t kTileTopkMax = 128;

// ---- stage.cpp: index data into HBM
cobs_gpu_status alloc_part(cobs_gpu_index* ix, Part& pt);
cobs_gpu_status upload_resident(Part& pt, const uint8_t* file);
SynthArgs synth_args(const Part& pt, const Chunk& c, uint8_t* data);
cobs_gpu_status stream_chunk_in(cobs_gpu_index* ix, Part& pt, const Chunk& c, int buf);

// ---- engine.cpp: the thread's error text (cobs_gpu_last_error)
std::string& last_error_text();

// ---- pass.cpp
uint64_t gathered_row_bytes(const Part& p);
uint64_t pass_shape_class(const cobs_gpu_batch* b);
void set_run_state(cobs_gpu_batch* b, double threshold, size_t topk, bool want_counts);
void stage_thresholds(cobs_gpu_batch* b, double threshold);

// ---- results.cpp
// `n` hit records at d_pool (the batch's own pool, or the pools of all shards after an exchange) -> b->h_hits in result
// order (query ascending, inside a query as counts_to_result orders it), b->h_hit_off; ordered on the device
cobs_gpu_status order_pool(cobs_gpu_batch* b, const HitDev* d_pool, uint64_t n, hipStream_t st);
// ... in two halves: everything queued on `st` (ERR_UNSUPPORTED, nothing queued: beyond the device ordering's indices) | waited for and published
cobs_gpu_status order_pool_launch(cobs_gpu_batch* b, const HitDev* d_pool, uint64_t n, hipStream_t st);
cobs_gpu_status order_pool_collect(cobs_gpu_batch* b, hipStream_t st, bool waited = false);     // waited: the caller saw an event behind the copy home
// the ordered lists of a whole pass into the caller's arrays in one sweep (false: they do not fit, nothing written)
bool hand_over_pool(cobs_gpu_batch* sb, size_t g0, size_t g1, cobs_gpu_hit** hits, size_t* cap, size_t* used, size_t* hit_offsets,
                    ResultArena* grow, size_t nq_call, cobs_gpu_status* status);
// ... and the ordered best-of lists of a limited pass over one file (false: not that case, nothing written)
bool hand_over_topk(cobs_gpu_batch* b, size_t g0, size_t g1, size_t num_results, cobs_gpu_hit* hits, size_t cap, size_t* used,
                    size_t* hit_offsets, cobs_gpu_status* status);
cobs_gpu_status fetch_counts(cobs_gpu_batch* b, size_t q, uint32_t* counts);
cobs_gpu_status rank_window(cobs_gpu_batch* b, size_t q0, size_t q1, size_t per_query, cobs_gpu_hit* hits);

// pass.cpp internals used by comm.cpp
cobs_gpu_status run_impl(cobs_gpu_batch* b, double threshold, size_t topk, void* hip_stream, bool want_counts = true);
cobs_gpu_status set_queries_on(cobs_gpu_batch* b, const char* const* queries, const size_t* lens, size_t nq,
                               hipStream_t up, bool wait, size_t* bad_query, size_t index_base = 0);
uint32_t threshold_for(double threshold, uint64_t terms);
uint64_t total_hashes(const cobs_gpu_batch* b, size_t q);
bool hit_before(const cobs_gpu_hit& a, const cobs_gpu_hit& b);
bool doc_before(const cobs_gpu_hit& a, const cobs_gpu_hit& b);
void destroy_exchange(Exchange* x);       // comm.cpp
// rank.cpp: counts_to_result over the score rows of the last run, on the device
void destroy_rank_work(RankWork* w);
bool rank_on_device_applies(const cobs_gpu_batch* b, size_t nq);
cobs_gpu_status rank_launch(cobs_gpu_batch* b, size_t q_first, size_t nq, size_t limit);
void rank_cancel(cobs_gpu_batch* b);
cobs_gpu_status rank_on_device(cobs_gpu_batch* b, size_t q_first, size_t nq, size_t limit, cobs_gpu_hit* hits, size_t cap,
                               size_t* used, size_t* hit_offsets, bool* overflow);
cobs_gpu_status open_zeroed(IndexMeta&& meta, const cobs_gpu_options* opts, cobs_gpu_index** out);

}  // namespace cobs_amd
