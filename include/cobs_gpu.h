/*
 * include/cobs_gpu.h -- C ABI of libcobs_gpu.so, the MI355X (gfx950) query
 * engine for COBS bit-sliced signature indexes (.cobs_classic / .cobs_compact).
 *
 * This is the drop-in boundary for ONE path of bingmann/cobs:
 * cobs::ClassicSearch::search (reference cobs/query/classic_search.cpp:403-505)
 * and everything it calls.  Each entry point names the reference interface it
 * replaces.  Plain pointers and sizes only; no C++ / torch types.  Functions
 * return a cobs_gpu_status; they never abort or exit (the reference terminates
 * the process on bad input, see INTEGRATION.md for the mapping).
 *
 * Threading: one handle / one batch = one in-flight call.  Different handles
 * may be used from different threads.
 */
#ifndef COBS_GPU_H
#define COBS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COBS_GPU_ABI_VERSION 2

typedef enum cobs_gpu_status {
    COBS_GPU_OK = 0,
    COBS_GPU_ERR_OPEN = 1,            /* index file cannot be opened / mapped   (util/query.cpp:24-31) */
    COBS_GPU_ERR_FORMAT = 2,          /* neither classic nor compact header     (classic_search.cpp:61-63) */
    COBS_GPU_ERR_QUERY_TOO_SHORT = 3, /* "query too short, needs to be at least k characters" (:431-433) */
    COBS_GPU_ERR_INVALID_BASE = 4,    /* "Invalid DNA base pair in query string" (:93-96) */
    COBS_GPU_ERR_QUERY_TOO_LONG = 5,  /* "query too long" (:323-327, :501-503) */
    COBS_GPU_ERR_HIP = 6,             /* HIP runtime error, text in cobs_gpu_last_error() */
    COBS_GPU_ERR_ARG = 7,             /* NULL / out-of-range argument */
    COBS_GPU_ERR_UNSUPPORTED = 8,     /* legal index the engine cannot hold (e.g. more than 2^32 score slots in one file) */
    COBS_GPU_ERR_CAPACITY = 9,        /* caller buffer too small; *n_out holds the needed size */
    COBS_GPU_ERR_NO_DEVICE = 10,      /* no HIP device (the library has no CPU fallback) */
    COBS_GPU_ERR_RCCL = 11            /* RCCL error in the multi-GPU exchange, text in cobs_gpu_last_error() */
} cobs_gpu_status;

/* Opaque handles. */
typedef struct cobs_gpu_index cobs_gpu_index;   /* >= 1 index files resident in HBM  (ClassicSearch::index_files_) */
typedef struct cobs_gpu_batch cobs_gpu_batch;   /* device workspace of one query batch */
typedef struct cobs_gpu_comm cobs_gpu_comm;     /* one rank of an RCCL communicator (multi-GPU exchange) */

typedef struct cobs_gpu_options {
    uint32_t struct_size;     /* sizeof(cobs_gpu_options) */
    int32_t device;           /* HIP device ordinal, -1 = current device */
    /* Multi-GPU sharding by sub-index block: this process keeps the
     * sub-indexes (compact) / row-byte columns (classic) of shard
     * shard_rank out of shard_count; counts of other shards' documents are 0.  */
    uint32_t shard_rank;
    uint32_t shard_count;     /* 0 or 1 = unsharded */
    uint32_t waves_per_group; /* waves that split one query's terms: 1, 2 or 4; 0 = chosen by query length */
    /* how a file is cut into shard_count shards (the unit is a 16-byte column chunk of one sub-index;
     * every shard holds a contiguous range of score slots):
     *   0 = equal bytes per shard; a cut may fall inside a sub-index (column range)
     *   1 = whole sub-indexes, equal count per shard (classic: 16-byte columns)            */
    uint32_t shard_mode;
    /* 0 = stage the whole (shard of the) index into HBM.  Otherwise the index may
     * use at most this many bytes of HBM: files that do not fit are cut into chunks
     * (whole sub-indexes, or pieces of one sub-index larger than a buffer: row ranges
     * when the file has one hash function, column ranges otherwise) that are streamed
     * from the mapped file through two device buffers with hipMemcpyAsync, one scan pass
     * per chunk overlapping the next chunk's copy (the successor of the reference's
     * mmap / AIO back-ends, util/query.cpp:38-88, compact_index/aio_search_file.cpp).
     * A handle with row-range chunks counts such a sub-index range by range (partial
     * scores are added up), so its passes always keep score rows: the hits-only and
     * top-k-only runs and every search call give the same results, selected from the
     * rows; cobs_gpu_batch_exchange_hits / _hits_owned are not available on it
     * (cobs_gpu_sharded_search_batch takes the row exchange by itself). */
    uint64_t hbm_budget_bytes;
} cobs_gpu_options;

/* Geometry of one opened index file (IndexSearchFile getters, query/index_file.hpp:19-35). */
typedef struct cobs_gpu_index_info {
    uint32_t kind;            /* 0 classic, 1 compact */
    uint32_t term_size;       /* k */
    uint32_t canonicalize;
    uint32_t num_pages;       /* sub-indexes; classic: 1 */
    uint64_t num_hashes;
    uint64_t page_size;       /* as IndexSearchFile::page_size(): classic reports 1 */
    uint64_t row_size;        /* bytes per full row: classic ceil(D/8); compact page_size * P */
    uint64_t counts_size;     /* 8 * row_size: score slots incl. padding documents */
    uint64_t num_docs;        /* file_names().size() */
    uint64_t doc_offset;      /* first score slot of this file in a concatenated count vector */
    uint64_t hbm_bytes;       /* bytes this file occupies in HBM on this process */
    uint32_t first_page;      /* sub-index range held by this shard: [first_page, end_page) */
    uint32_t end_page;
    uint64_t slot_begin;      /* score slots of this file computed by this shard: */
    uint64_t slot_count;      /*   [slot_begin, slot_begin + slot_count), multiples of 8 */
    uint64_t local_offset;    /* where those slots start in this process' local count vector */
} cobs_gpu_index_info;

/* One ranked hit (cobs::SearchResult, query/search.hpp:17-27, plus ids). */
typedef struct cobs_gpu_hit {
    uint32_t file_no;         /* which index file of the handle */
    uint32_t doc;             /* document id inside that file */
    uint32_t score;           /* number of matching k-mers */
} cobs_gpu_hit;

/* Parameters of a procedural (synthetic) index filled directly in HBM; the
 * benchmark-sized stand-in for `cobs classic-construct-random`
 * (construction/classic_index.cpp:661-725).  Bits are a pure function of
 * (seed, page, row, byte) with density ~0.297 so that any row can be recomputed
 * by a checker; documents >= num_docs have no bits. */
typedef struct cobs_gpu_synth {
    uint32_t kind;            /* 0 classic, 1 compact */
    uint32_t term_size;
    uint32_t canonicalize;
    uint32_t num_pages;       /* classic: 1 */
    uint64_t num_hashes;
    uint64_t page_size;       /* compact only */
    uint64_t num_docs;
    uint64_t seed;
    const uint64_t* signature_sizes;   /* num_pages entries */
} cobs_gpu_synth;

/* ---- library ----------------------------------------------------------- */
uint32_t cobs_gpu_abi_version(void);
/* thread-local text of the last error on this thread */
const char* cobs_gpu_last_error(void);
/* number of visible HIP devices (0 if none) */
int cobs_gpu_device_count(void);

/* ---- index ------------------------------------------------------------- */
/* ClassicSearch(std::string path) / ClassicSearch(vector<IndexSearchFile>)
 * (classic_search.cpp:41-64): sniff classic vs compact per file, stage every
 * sub-index into HBM (replaces initialize_mmap, util/query.cpp:38-88).        */
cobs_gpu_status cobs_gpu_open(const char* const* paths, size_t n_paths,
                              const cobs_gpu_options* opts, cobs_gpu_index** out);
cobs_gpu_status cobs_gpu_open_synthetic(const cobs_gpu_synth* desc,
                                        const cobs_gpu_options* opts, cobs_gpu_index** out);
void cobs_gpu_close(cobs_gpu_index* ix);
/* Per-handle tuning hooks of the scan launch (the COBS_GPU_* environment variables are read once,
 * by cobs_gpu_open*; this changes them afterwards).  key: "waves" (0, 1, 2, 4), "tile_w" (0, 4..64),
 * "mq" (-1 auto, 0, 1), "pass_bytes", "pipe_chars", "graph" (-1 auto, 0, 1), "lds_staged" (0, 1: the
 * measured LDS-staged variant of the scan, headline shape only), "device_rank" / "tile_topk" / "row_fetch" (0 turns the
 * on-device ranking of whole rows / the tile-level top-k / the row-selective out-of-core pass off: A/B and fallback),
 * "row_fetch_alpha", "min_score_bytes" (2 / 4: score rows at least that wide -- the reference's
 * classic_search_disable_8bit / _16bit switches, classic_search.cpp:207-209).  0 / -1 = automatic. */
cobs_gpu_status cobs_gpu_set_tuning(cobs_gpu_index* ix, const char* key, int64_t value);
/* Host only (no device needed): the score slots [slot_begin[r], slot_begin[r] + slot_count[r]) and
 * the index bytes shard r of shard_count would hold of the file at `path` (arrays of shard_count
 * entries; bytes may be NULL).  The slot ranges are disjoint, ascending and cover counts_size. */
cobs_gpu_status cobs_gpu_plan_shards(const char* path, uint32_t shard_count, uint32_t shard_mode,
                                     uint64_t* slot_begin, uint64_t* slot_count, uint64_t* bytes);

size_t cobs_gpu_num_files(const cobs_gpu_index* ix);
cobs_gpu_status cobs_gpu_info(const cobs_gpu_index* ix, size_t file_no, cobs_gpu_index_info* info);
/* signature_size of sub-index `page` of file `file_no` (0 if out of range) */
uint64_t cobs_gpu_signature_size(const cobs_gpu_index* ix, size_t file_no, uint32_t page);
/* file_names()[doc].c_str(): owned by the handle, valid until cobs_gpu_close */
const char* cobs_gpu_doc_name(const cobs_gpu_index* ix, size_t file_no, uint64_t doc);
/* sum of counts_size over all files = length of one query's count vector */
uint64_t cobs_gpu_total_counts(const cobs_gpu_index* ix);
/* score slots per query held by THIS shard (== cobs_gpu_total_counts when
 * unsharded); device count rows have this many elements */
uint64_t cobs_gpu_local_counts(const cobs_gpu_index* ix);
/* copy `n` bytes of row `row` of sub-index `page` back from HBM (diagnostics/tests) */
cobs_gpu_status cobs_gpu_read_row(const cobs_gpu_index* ix, size_t file_no, uint32_t page,
                                  uint64_t row, uint8_t* out, size_t n);

/* row bytes [*col0, *col0 + *ncols) of sub-index `page` that this shard holds (0, 0 if none) */
cobs_gpu_status cobs_gpu_page_columns(const cobs_gpu_index* ix, size_t file_no, uint32_t page,
                                      uint64_t* col0, uint64_t* ncols);
/* the valid bytes of rows [row0, row0+nrows) of a held sub-index, out_pitch bytes apart */
cobs_gpu_status cobs_gpu_read_rows(const cobs_gpu_index* ix, size_t file_no, uint32_t page,
                                   uint64_t row0, uint64_t nrows, uint8_t* out, size_t out_pitch);

/* The procedural index of cobs_gpu_open_synthetic written as a .cobs_classic / .cobs_compact FILE
 * (the generator tool of SURVEY 8f rank 2, cf. `cobs classic-construct-random`, src/cobs.cpp:243-291):
 * rows are produced on the device chunk by chunk and streamed to the file. */
cobs_gpu_status cobs_gpu_write_synthetic(const cobs_gpu_synth* desc, const char* out_path, int device);

/* ---- search (host buffers in, host buffers out) ------------------------ */
/* ClassicSearch::search (classic_search.cpp:403-505): hits ordered by score
 * descending, ties by (file_no, doc) ascending; no ordering when the query has a
 * single hash in total (max_counts <= 1, :136,:179).  num_results == 0 = all.  */
cobs_gpu_status cobs_gpu_search(cobs_gpu_index* ix, const char* query, size_t len,
                                double threshold, size_t num_results,
                                cobs_gpu_hit* hits, size_t cap, size_t* n_hits);

/* The same for nq queries (src/cobs.cpp:424-465 loops them one by one).  The call is
 * one device pass, or -- for 4 MiB of query text and more, or when the workspaces
 * would exceed 16 GiB -- several passes whose staging, upload and ranking overlap
 * the scans; the caller sees one call.  With threshold > 0 and num_results == 0 the
 * scan selects hits on the device and writes no score rows.  hit_offsets has nq+1
 * entries; hits of query i are hits[hit_offsets[i] .. hit_offsets[i+1]).  A query with
 * bad input fails the whole call; *bad_query (optional) receives its index.
 * If cap is too small the call returns COBS_GPU_ERR_CAPACITY and hit_offsets[nq]
 * holds the capacity a retry needs (hit_offsets stay valid, hits do not).      */
cobs_gpu_status cobs_gpu_search_batch(cobs_gpu_index* ix, const char* const* queries,
                                      const size_t* lens, size_t nq,
                                      double threshold, size_t num_results,
                                      cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets,
                                      size_t* bad_query);

/* Raw per-document counts of one query over all files (the reference's
 * score_list before counts_to_result, classic_search.cpp:456-467), widened to
 * u32; cap >= cobs_gpu_total_counts(). */
cobs_gpu_status cobs_gpu_counts(cobs_gpu_index* ix, const char* query, size_t len,
                                uint32_t* counts, size_t cap);

/* ---- device-resident batches (benchmark / multi-GPU plumbing) ---------- */
/* Workspace for up to max_queries queries of up to max_query_len characters. */
cobs_gpu_status cobs_gpu_batch_create(cobs_gpu_index* ix, size_t max_queries,
                                      size_t max_query_len, cobs_gpu_batch** out);
void cobs_gpu_batch_destroy(cobs_gpu_batch* b);
/* Copy query text to HBM (one H2D) and validate lengths.  After this call the
 * inputs of cobs_gpu_batch_run are resident in HBM.  A run of the batch that is
 * still in flight and was never synced is waited for first. */
cobs_gpu_status cobs_gpu_batch_set_queries(cobs_gpu_batch* b, const char* const* queries,
                                           const size_t* lens, size_t nq);
/* One pass of the hot path over the batch, asynchronously on `hip_stream`
 * (a hipStream_t, NULL = default stream): K1 canonicalise + XXH64 + row index
 * per sub-index (create_hashes, :66-107), K2 row gather + AND + bit-sliced
 * per-document count (read_from_disk / aggregate_rows / compute_counts,
 * :279-307, :643-1022), and, if threshold > 0, on-device selection of documents
 * with count >= ceil(threshold * T) (:127-132).  Counts stay in HBM.          */
cobs_gpu_status cobs_gpu_batch_run(cobs_gpu_batch* b, double threshold, void* hip_stream);
/* The same pass without score rows (threshold > 0 required): the comparison count >= ceil(threshold * T)
 * is done on the bit-sliced counters, only the selected (query, file, doc, score) records are
 * written.  cobs_gpu_batch_hits_host returns them; if the selection pool overflowed it fails
 * with COBS_GPU_ERR_ARG ("did not keep the score rows"): rerun with cobs_gpu_batch_run, as
 * cobs_gpu_search_batch does on its own. */
cobs_gpu_status cobs_gpu_batch_run_hits(cobs_gpu_batch* b, double threshold, void* hip_stream);
/* The same pass followed by K3: on-device selection of the num_results best
 * documents per query (score descending, ties by document ascending -- the set
 * std::partial_sort keeps, classic_search.cpp:134-145) among those with
 * count >= ceil(threshold * T), left in result order on the device (all score widths: 8, 16 and
 * 32 bit); cobs_gpu_batch_hits_host then moves only those.                      */
cobs_gpu_status cobs_gpu_batch_run_topk(cobs_gpu_batch* b, double threshold, size_t num_results,
                                        void* hip_stream);
/* The top-k pass WITHOUT score rows (the counterpart of cobs_gpu_batch_run_hits): counts_to_result keeps the
 * num_results best while it scans (classic_search.cpp:127-145) and needs no score matrix either.  K2 selects the
 * num_results best documents of every tile from its bit-sliced counters, K3 merges tiles x num_results
 * candidates per query; same result as cobs_gpu_batch_run_topk, cobs_gpu_batch_counts_* are not available
 * afterwards.  Where the tile-level selection does not apply (num_results > 128, a query with a single hash in
 * total, sub-indexes of 2^32 rows and more) the pass keeps score rows as cobs_gpu_batch_run_topk does. */
cobs_gpu_status cobs_gpu_batch_run_topk_only(cobs_gpu_batch* b, double threshold, size_t num_results,
                                             void* hip_stream);
/* wait for the stream and fetch device-side error flags (invalid bases, ...) */
cobs_gpu_status cobs_gpu_batch_sync(cobs_gpu_batch* b, void* hip_stream, size_t* bad_query);
/* Device pointer to the counts of the last run: row i (query i) starts at
 * ptr + i * row_stride_bytes and holds cobs_gpu_local_counts() elements of
 * elem_bytes each: 1 when no query of the batch has more than 255 terms, 2 up to
 * 65535, else 4 -- the Score widths of classic_search.cpp:453-504.  Valid until the
 * batch is destroyed. */
void* cobs_gpu_batch_counts_device(cobs_gpu_batch* b, uint32_t* elem_bytes, uint64_t* row_stride_bytes);
/* D2H of one query's counts widened to u32 */
cobs_gpu_status cobs_gpu_batch_counts_host(cobs_gpu_batch* b, size_t query_no, uint32_t* counts, size_t cap);
/* D2H + rank the hits of query `query_no` of the last run */
cobs_gpu_status cobs_gpu_batch_hits_host(cobs_gpu_batch* b, size_t query_no, size_t num_results,
                                         cobs_gpu_hit* hits, size_t cap, size_t* n_hits);

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

/* ---- multi-GPU: index sharded by sub-index block, one exchange per batch over RCCL / xGMI ----
 * (the shard boundary is the reference's own: sub-indexes cover disjoint document
 * ranges, compact_index/mmap_search_file.cpp:22-27, search_file.cpp:30-32.)  One rank = one GPU =
 * one cobs_gpu_index opened with shard_rank / shard_count = its rank / the communicator size.
 * The launcher (torch.distributed, MPI, threads of one process...) only has to hand the unique id
 * from rank 0 to the others.  All calls below are collective: every rank makes the same call.  */
#define COBS_GPU_UNIQUE_ID_BYTES 128
/* Side effect of the two calls below: RCCL prints a version banner to stdout when a process first initialises
 * it; while they run, file descriptor 1 of the PROCESS points at stderr (and is put back afterwards), so that the
 * caller's stdout stays clean -- output other threads write to stdout in that window lands on stderr. */
cobs_gpu_status cobs_gpu_comm_unique_id(uint8_t id[COBS_GPU_UNIQUE_ID_BYTES]);          /* ncclGetUniqueId */
cobs_gpu_status cobs_gpu_comm_create(const uint8_t id[COBS_GPU_UNIQUE_ID_BYTES], int rank, int nranks,
                                     int device /* -1 = current */, cobs_gpu_comm** out);  /* ncclCommInitRank */
void cobs_gpu_comm_destroy(cobs_gpu_comm* c);
int cobs_gpu_comm_rank(const cobs_gpu_comm* c);     /* ncclCommUserRank, -1 on error */
int cobs_gpu_comm_size(const cobs_gpu_comm* c);     /* ncclCommCount, 0 on error */

typedef enum cobs_gpu_exchange_mode {
    COBS_GPU_XCHG_ALLGATHER = 0,  /* every rank receives the count slices of all ranks for all queries
                                     (ncclAllGather when the slices have one size, else grouped send/recv) */
    COBS_GPU_XCHG_ALLTOALL = 1,   /* rank j receives the slices of the queries [nq*j/N, nq*(j+1)/N) only:
                                     every count crosses the fabric once (grouped ncclSend / ncclRecv)      */
    COBS_GPU_XCHG_REDUCE = 2      /* the counts "reduced over RCCL": every rank lays its slices into zeroed
                                     rows of global length, one ncclAllReduce(sum) over the bytes (disjoint
                                     slices: no byte has two non-zero addends, so the byte-wise sum is exact
                                     for every counter width).  The parity form; the gather forms move less */
} cobs_gpu_exchange_mode;
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
/* After cobs_gpu_batch_run on every rank: exchange the per-document counts of the shards on
 * `hip_stream` (asynchronous, ordered after the scan) and assemble rows in global document order. */
cobs_gpu_status cobs_gpu_batch_exchange_counts(cobs_gpu_batch* b, cobs_gpu_comm* c, uint32_t mode, void* hip_stream);
/* The assembled rows of queries [*q_begin, *q_begin + *q_count): cobs_gpu_total_counts() elements of
 * *elem_bytes each, *row_stride_bytes apart.  NULL before an exchange.  Valid until the next run. */
void* cobs_gpu_batch_global_counts_device(cobs_gpu_batch* b, uint64_t* q_begin, uint64_t* q_count,
                                          uint32_t* elem_bytes, uint64_t* row_stride_bytes);
/* bytes this rank received from other ranks in the last exchange */
uint64_t cobs_gpu_batch_exchange_bytes(const cobs_gpu_batch* b);
/* After a synced run with a threshold: gather the selected (query, file, doc, score) records of all
 * shards (sizes first, then the records); cobs_gpu_batch_hits_host then returns global results.
 * *overflow = 1 if a shard's pool overflowed (lists incomplete on every rank: rerun with score rows). */
cobs_gpu_status cobs_gpu_batch_exchange_hits(cobs_gpu_batch* b, cobs_gpu_comm* c, void* hip_stream, int* overflow);
/* The same exchange with every record sent ONCE, to the rank that owns its query: rank j owns the queries
 * [nq*j/N, nq*(j+1)/N) (as in COBS_GPU_XCHG_ALLTOALL) and ends with the hits of exactly those queries from every shard
 * (*q_begin / *q_count, optional); cobs_gpu_batch_hits_host then answers for them and refuses the others. */
cobs_gpu_status cobs_gpu_batch_exchange_hits_owned(cobs_gpu_batch* b, cobs_gpu_comm* c, void* hip_stream, int* overflow,
                                                   uint64_t* q_begin, uint64_t* q_count);
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
/* After a run with num_results > 0: all-gather the k best documents of every shard. */
cobs_gpu_status cobs_gpu_batch_exchange_topk(cobs_gpu_batch* b, cobs_gpu_comm* c, void* hip_stream);
/* cobs_gpu_search_batch over the sharded index: same arguments and result on every rank. */
cobs_gpu_status cobs_gpu_sharded_search_batch(cobs_gpu_index* ix, cobs_gpu_comm* c, const char* const* queries,
                                              const size_t* lens, size_t nq, double threshold, size_t num_results,
                                              cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets, size_t* bad_query);

/* The same call with the ranking SHARED by the ranks where that is possible: for the all-documents search (threshold <= 0
 * and no limit -- the reference's default call) every query yields one result per document, so every result's place in
 * `hits` is known up front; the count rows go all-to-all to query owners and rank j writes the results and offsets of the
 * queries [n*j/N, n*(j+1)/N) of every pass at their final places.  Ranks of ONE process pass the same arrays (together
 * they fill them, every entry written by exactly one rank: cobs_gpu_multi_search_batch does this); ranks in several
 * processes each get their part filled (hit_offsets[0] and, on ERR_CAPACITY, the needed sizes by rank 0 only).  All
 * ranks must pass the same cap.  Every other search behaves exactly like cobs_gpu_sharded_search_batch. */
cobs_gpu_status cobs_gpu_sharded_search_batch_split(cobs_gpu_index* ix, cobs_gpu_comm* c, const char* const* queries,
                                                    const size_t* lens, size_t nq, double threshold, size_t num_results,
                                                    cobs_gpu_hit* hits, size_t cap, size_t* hit_offsets, size_t* bad_query);

/* Diagnostics of tuning builds (libcobs_gpu_timing.so, `make -C cobs_amd/csrc timing`): s_memtime stamps
 * [work-group slot][wave 0..3][8 phases] of the work-groups sampled from the last scan launch after
 * cobs_gpu_set_tuning(ix, "phase_slots", n).  The production library records nothing (*n_words = 0). */
cobs_gpu_status cobs_gpu_batch_phase_stamps(cobs_gpu_batch* b, uint64_t* out, size_t cap_words, size_t* n_words);

/* Small calls of the host-buffer API (up to 16 queries) are captured into a hipGraph the second time
 * a pass of the same shape class (query count, score width, launch geometry -- not the exact query lengths --, same
 * parameters) comes along and replayed with one launch afterwards;
 * this counts the replays (diagnostics; tuning key "graph" = 0 turns the path off). */
uint64_t cobs_gpu_graph_replays(const cobs_gpu_index* ix);

/* Out-of-core handles (hbm_budget_bytes): how the chunks of all passes so far were brought into HBM.
 * out[0] = chunks whose looked-up rows were fetched one by one from the registered file mapping (a batch that
 * touches a fraction of the chunk's rows: the access pattern of the reference's mmap / AIO back-ends,
 * compact_index/mmap_search_file.cpp:34-67, aio_search_file.cpp:58-97), out[1] = chunks copied whole.
 * Tuning keys "row_fetch" (0 = always whole) and "row_fetch_alpha" (fetch when alpha x looked-up bytes <= the
 * chunk's bytes; default 1, 0 = whenever the rows fit a stream buffer) steer the choice. */
cobs_gpu_status cobs_gpu_stream_counters(const cobs_gpu_index* ix, uint64_t out[2]);

/* ---- multi-GPU, one process: a device list behind ONE handle -----------------------------------
 * cobs_gpu_multi_open shards the index over devices[0..n_devices) (opts: hbm_budget_bytes and
 * shard_mode are honoured per shard; device / shard_rank / shard_count are set by the library),
 * starts one worker thread per device and joins them in an RCCL communicator;
 * cobs_gpu_multi_search_batch is cobs_gpu_search_batch over all of them (same arguments, same
 * result, same error behaviour; one call in flight per handle). */
typedef struct cobs_gpu_multi cobs_gpu_multi;
cobs_gpu_status cobs_gpu_multi_open(const char* const* paths, size_t n_paths, const int* devices, size_t n_devices,
                                    const cobs_gpu_options* opts, cobs_gpu_multi** out);
void cobs_gpu_multi_close(cobs_gpu_multi* m);
size_t cobs_gpu_multi_size(const cobs_gpu_multi* m);                      /* ncclCommCount of its communicator */
/* the shard handle of one rank (geometry; document names via cobs_gpu_doc_name work on any rank) */
cobs_gpu_index* cobs_gpu_multi_index(const cobs_gpu_multi* m, size_t rank);
cobs_gpu_status cobs_gpu_multi_search_batch(cobs_gpu_multi* m, const char* const* queries, const size_t* lens, size_t nq,
                                            double threshold, size_t num_results, cobs_gpu_hit* hits, size_t cap,
                                            size_t* hit_offsets, size_t* bad_query);

/* phase timers of the host-buffer search API since the last reset, seconds:
 * out[0] hashes (K1), out[1] h2d, out[2] scan (K2), out[3] d2h, out[4] rank  */
cobs_gpu_status cobs_gpu_timers(cobs_gpu_index* ix, double out[5], int reset);

#ifdef __cplusplus
}
#endif
#endif /* COBS_GPU_H */
