/*
 * include/cobs_gpu_diag.h -- diagnostics and test hooks of libcobs_gpu.so (nothing a caller of the search path
 * needs): the planners as host arithmetic (shards, exchanges -- what lets N ranks be checked without N GPUs), rows of
 * the resident matrix read back, per-launch bookkeeping for rooflines, counters of the engine's internal choices.
 */
#ifndef COBS_GPU_DIAG_H
#define COBS_GPU_DIAG_H

#include "cobs_gpu_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- planners (host only) ------------------------------------------------ */
/* Host only (no device needed): the score slots [slot_begin[r], slot_begin[r] + slot_count[r]) and
 * the index bytes shard r of shard_count would hold of the file at `path` (arrays of shard_count
 * entries; bytes may be NULL).  The slot ranges are disjoint, ascending and cover counts_size. */
cobs_gpu_status cobs_gpu_plan_shards(const char* path, uint32_t shard_count, uint32_t shard_mode,
                                     uint64_t* slot_begin, uint64_t* slot_count, uint64_t* bytes);

/* The exchange as a plan (host arithmetic only, no device, no communicator): what rank `rank` of
 * `nranks` sends to / receives from every peer and how the received slices are assembled, given all
 * ranks' slot layouts (slot_begin / slot_count: [nranks][nfiles], as cobs_gpu_info reports them).
 * cobs_gpu_batch_exchange_counts executes exactly this plan over RCCL; tests emulate N ranks with it. */
typedef struct cobs_gpu_xfer {
    uint64_t peer;
    uint64_t send_offset, send_bytes;   /* inside this rank's local count rows */
    uint64_t recv_offset, recv_bytes;   /* inside this rank's staging buffer */
} cobs_gpu_xfer;
typedef struct cobs_gpu_copy2d {        /* strided copy into the assembled rows (global document order) */
    uint64_t src_rank;
    uint64_t src_is_local;              /* 1: source is this rank's own count rows, 0: the staging buffer */
    uint64_t src_offset, src_pitch, dst_offset, dst_pitch, width, height;
} cobs_gpu_copy2d;
/* xfers: nranks entries; copies: *n_copies capacity in, count out (at most nranks * nfiles);
 * out = { q_begin, q_count, staging_bytes, assembled_bytes, uses_ncclAllGather, local_row_bytes } */
cobs_gpu_status cobs_gpu_exchange_plan(const uint64_t* slot_begin, const uint64_t* slot_count, const uint64_t* doc_offset,
                                       size_t nranks, size_t nfiles, uint64_t total_counts, size_t nq, uint32_t elem_bytes,
                                       uint32_t mode, size_t rank, cobs_gpu_xfer* xfers, cobs_gpu_copy2d* copies,
                                       size_t* n_copies, uint64_t out[6]);
/* ... as a plan (host arithmetic only): counts[r * nranks + j] = records rank r holds for the queries of rank j;
 * xfers[j] = what `rank` sends to / receives from rank j (bytes; send offsets inside its pool bucketed by owner,
 * receive offsets inside its staging buffer, rank after rank); out = { bytes received incl. its own bucket,
 * bytes of its pool }.  cobs_gpu_batch_exchange_hits_owned executes exactly this plan. */
cobs_gpu_status cobs_gpu_hit_exchange_plan(const uint64_t* counts, size_t nranks, size_t rank, cobs_gpu_xfer* xfers,
                                           uint64_t out[2]);
/* Diagnostics / tests: the device-side half of that exchange for any rank count, without a communicator: the hit
 * pool of the last synced thresholded run bucketed by owner as `nranks` ranks would bucket it -- counts[j] records
 * for rank j, the buckets back to back in `records` as (query, file, document, score) quadruples of uint32. */
cobs_gpu_status cobs_gpu_batch_bucketed_hits(cobs_gpu_batch* b, uint32_t nranks, uint64_t* counts, uint32_t* records,
                                             size_t cap_records, size_t* n_records);

/* ---- rows of the matrix as HBM holds them --------------------------------- */
/* copy `n` bytes of row `row` of sub-index `page` back from HBM (diagnostics/tests) */
cobs_gpu_status cobs_gpu_read_row(const cobs_gpu_index* ix, size_t file_no, uint32_t page,
                                  uint64_t row, uint8_t* out, size_t n);

/* row bytes [*col0, *col0 + *ncols) of sub-index `page` that this shard holds (0, 0 if none) */
cobs_gpu_status cobs_gpu_page_columns(const cobs_gpu_index* ix, size_t file_no, uint32_t page,
                                      uint64_t* col0, uint64_t* ncols);
/* the valid bytes of rows [row0, row0+nrows) of a held sub-index, out_pitch bytes apart */
cobs_gpu_status cobs_gpu_read_rows(const cobs_gpu_index* ix, size_t file_no, uint32_t page,
                                   uint64_t row0, uint64_t nrows, uint8_t* out, size_t out_pitch);

/* ---- bookkeeping ---------------------------------------------------------- */
/* Per-launch bookkeeping of the last cobs_gpu_batch_run (for rooflines):
 * out[0] = algorithmic bytes of the scan kernel(s): sum over queries of
 *          T * H * (row bytes gathered) + score bytes written
 * out[1] = number of scan-kernel launches, out[2] = k-mer lookups (sum T),
 * out[3] = bytes of row-index table written by K1 and read by K2.            */
cobs_gpu_status cobs_gpu_batch_stats(const cobs_gpu_batch* b, uint64_t out[4]);
/* HIP-event duration (ms) of the scan kernel(s) / hash kernel, averaged over the
 * runs since the previous call (at most the last 64); events are recorded on the
 * stream the kernels were launched on.  Call after cobs_gpu_batch_sync. */
cobs_gpu_status cobs_gpu_batch_kernel_ms(cobs_gpu_batch* b, float* scan_ms, float* hash_ms);

/* Diagnostics of tuning builds (libcobs_gpu_timing.so, `make -C cobs_amd/csrc timing`): s_memtime stamps
 * [work-group slot][wave 0..3][8 phases] of the work-groups sampled from the last scan launch after
 * cobs_gpu_set_tuning(ix, "phase_slots", n).  The production library records nothing (*n_words = 0). */
cobs_gpu_status cobs_gpu_batch_phase_stamps(cobs_gpu_batch* b, uint64_t* out, size_t cap_words, size_t* n_words);

/* Small calls of the host-buffer API (up to 16 queries) are captured into a hipGraph the second time
 * a pass of the same shape class (query count, score width, launch geometry -- not the exact query lengths --, same
 * parameters) comes along and replayed with one launch afterwards;
 * this counts the replays (diagnostics; tuning key "graph" = 0 turns the path off). */
uint64_t cobs_gpu_graph_replays(const cobs_gpu_index* ix);
/* Device passes the host-buffer calls (cobs_gpu_search, _search_batch, _search_batch_view, _counts) have launched on this
 * handle so far, replays included: a call is cut into passes (cobs_gpu.h), and a call that came back with
 * COBS_GPU_ERR_CAPACITY and was made again has paid for its passes twice -- the view call grows its arena instead. */
uint64_t cobs_gpu_host_passes(const cobs_gpu_index* ix);

/* Out-of-core handles (hbm_budget_bytes): how the chunks of all passes so far were brought into HBM.
 * out[0] = chunks whose looked-up rows were fetched one by one from the registered file mapping (a batch that
 * touches a fraction of the chunk's rows: the access pattern of the reference's mmap / AIO back-ends,
 * compact_index/mmap_search_file.cpp:34-67, aio_search_file.cpp:58-97), out[1] = chunks copied whole.
 * Tuning keys "row_fetch" (0 = always whole) and "row_fetch_alpha" (fetch when alpha x looked-up bytes <= the
 * chunk's bytes; default 1, 0 = whenever the rows fit a stream buffer) steer the choice. */
cobs_gpu_status cobs_gpu_stream_counters(const cobs_gpu_index* ix, uint64_t out[2]);
/* The same two counters in out[0] / out[1] plus out[2] / out[3] = the bytes those two ways asked of PCIe (the DISTINCT
 * looked-up rows x row pitch -- a row that several terms of a batch look up crosses once since round 6; counted by the
 * gather on the device, so this call waits for the device --; the rows of the whole chunks).  (Its own symbol: round 4 had widened cobs_gpu_stream_counters to
 * four words under the old name, which writes past the two-word buffer of a caller built against the older header.) */
cobs_gpu_status cobs_gpu_stream_traffic(const cobs_gpu_index* ix, uint64_t out[4]);
/* The plan of an out-of-core handle: out[0] = bytes of ONE of its two stream buffers, out[1] = bytes of the streamed
 * files' slices that stay RESIDENT beside them (the budget is spent per slice, not per file: what the buffers leave
 * over keeps the subset of whole slices with the most bytes in HBM), out[2] = row bytes a pass moves over PCIe when it
 * copies every other chunk whole, out[3] = number of those chunks.  COBS_GPU_STREAM_BUF_KIB (read when the index is
 * opened) bounds a stream buffer (default 256 MiB, 0 = half the budget as in rounds 1-4). */
cobs_gpu_status cobs_gpu_stream_plan(const cobs_gpu_index* ix, uint64_t out[4]);
/* The same plan as host arithmetic, without a device (what cobs_gpu_open would decide for `path` under the budget and
 * the shard options): out as above (for a file that fits: out[1] = its bytes, the rest 0); resident[i] (optional, `cap`
 * entries; *n_slices = how many there are) = 1 if the i-th held slice stays in HBM.  For tests of the planner. */
cobs_gpu_status cobs_gpu_plan_stream(const char* path, uint64_t hbm_budget_bytes, uint32_t shard_rank, uint32_t shard_count,
                                     uint32_t shard_mode, uint64_t out[4], uint8_t* resident, size_t cap, size_t* n_slices);

#ifdef __cplusplus
}
#endif
#endif /* COBS_GPU_DIAG_H */
