/*
 * include/cobs_gpu.h -- C ABI of libcobs_gpu.so, the MI355X (gfx950) query
 * engine for COBS bit-sliced signature indexes (.cobs_classic / .cobs_compact).
 *
 * This is the drop-in boundary for ONE path of bingmann/cobs:
 * cobs::ClassicSearch::search (reference cobs/query/classic_search.cpp:403-505)
 * and everything it calls: open / geometry / search / timers, on one GPU or --
 * cobs_gpu_multi_* -- over a device list.  Each entry point names the reference
 * interface it replaces.  Two more headers declare what the benchmark, the
 * multi-process launchers and the tests use on top of it (all in libcobs_gpu.so):
 * cobs_gpu_batch.h (device-resident batches, RCCL exchange, the procedural
 * index) and cobs_gpu_diag.h (diagnostics, exchange plans, rows read back).  Plain pointers and sizes only; no C++ / torch types.  Functions
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
     *   0 = equal scan TIME per shard: a term looks up one row in every sub-index, so a shard's scan time follows the
     *       128-byte lines of a row (columns) it holds, not its bytes in HBM -- a line of a sub-index whose tile column
     *       stays in the Infinity Cache priced lower; a cut may fall inside a sub-index, on a whole line
     *   1 = whole sub-indexes, equal count per shard (classic: 16-byte columns)
     *   2 = equal bytes in HBM per shard (rows x columns): balances the footprint, not the scan time       */
    uint32_t shard_mode;
    /* 0 = stage the whole (shard of the) index into HBM.  Otherwise the index may
     * use at most this many bytes of HBM: files that do not fit are cut into chunks
     * (whole sub-indexes, or pieces of one sub-index larger than a buffer: row ranges
     * when the file has one hash function, column ranges otherwise) that are streamed
     * from the mapped file through two device buffers with hipMemcpyAsync, one scan pass
     * per chunk overlapping the next chunk's copy (the successor of the reference's
     * mmap / AIO back-ends, util/query.cpp:38-88, compact_index/aio_search_file.cpp).
     * A handle with row-range chunks counts such a sub-index range by range (partial
     * scores are added up); a pass without score rows (hits only, a limit only) adds them
     * up in a scratch matrix of the sub-index's own width and selects from that after the
     * sub-index's last range, so such a handle behaves like any other: same results, the
     * hit exchanges of cobs_gpu_batch.h included (round 6; it kept score rows of the whole
     * index until then). */
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
void cobs_gpu_close(cobs_gpu_index* ix);
/* Per-handle tuning hooks of the scan launch (the COBS_GPU_* environment variables are read once,
 * by cobs_gpu_open*; this changes them afterwards).  key: "waves" (0, 1, 2, 4), "tile_w" (0, 4..64),
 * "mq" (-1 auto, 0, 1), "pass_bytes", "pipe_chars", "graph" (-1 auto, 0, 1), "lds_staged" (0, 1: the
 * measured LDS-staged variant of the scan, headline shape only), "device_rank" / "tile_topk" / "row_fetch" (0 turns the
 * on-device ranking of whole rows / the tile-level top-k / the row-selective out-of-core pass off: A/B and fallback),
 * "row_fetch_alpha", "min_score_bytes" (2 / 4: score rows at least that wide -- the reference's
 * classic_search_disable_8bit / _16bit switches, classic_search.cpp:207-209); "rank_pack" / "rank_slim" / "rank_window_kib" / "rank_segments"
 * (how the ranked results of the default call cross PCIe: 4-byte records, slot streams, piece size; work-groups per row), "compact_terms",
 * "hash_stream": A/B switches named where DESIGN.md 3 describes what they switch.  0 / -1 = automatic. */
cobs_gpu_status cobs_gpu_set_tuning(cobs_gpu_index* ix, const char* key, int64_t value);
size_t cobs_gpu_num_files(const cobs_gpu_index* ix);
cobs_gpu_status cobs_gpu_info(const cobs_gpu_index* ix, size_t file_no, cobs_gpu_index_info* info);
/* signature_size of sub-index `page` of file `file_no` (0 if out of range) */
uint64_t cobs_gpu_signature_size(const cobs_gpu_index* ix, size_t file_no, uint32_t page);
/* file_names()[doc].c_str(): owned by the handle, valid until cobs_gpu_close */
const char* cobs_gpu_doc_name(const cobs_gpu_index* ix, size_t file_no, uint64_t doc);
/* sum of counts_size over all files = length of one query's count vector */
uint64_t cobs_gpu_total_counts(const cobs_gpu_index* ix);
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
 * holds the capacity a retry needs (hit_offsets stay valid, hits do not); the retry is
 * the whole search again -- size a thresholded call by the hits per query of earlier calls,
 * or use cobs_gpu_search_batch_view (cobs_gpu_batch.h), whose arena grows instead. */
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
 * out[0] hashes (K1), out[1] h2d, out[2] scan (K2; a small pass replayed from a captured graph is timed as a whole: its
 * device time -- hashing, scan, selection, its copies home -- goes here), out[3] d2h, out[4] rank  */
cobs_gpu_status cobs_gpu_timers(cobs_gpu_index* ix, double out[5], int reset);

#ifdef __cplusplus
}
#endif
#endif /* COBS_GPU_H */
