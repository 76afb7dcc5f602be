/*
 * include/cobs_gpu_batch.h -- the part of libcobs_gpu.so's C ABI that sits BESIDE the drop-in boundary
 * (include/cobs_gpu.h): device-resident query batches (inputs and counts stay in HBM: the benchmark's step, the
 * building blocks of the search calls), the one exchange step of the sub-index-sharded multi-GPU layout over RCCL
 * (one rank per process), cobs_gpu_search_batch over such a sharded index, and the procedural benchmark index.
 * Same conventions as cobs_gpu.h: plain pointers and sizes, cobs_gpu_status, nothing aborts.
 */
#ifndef COBS_GPU_BATCH_H
#define COBS_GPU_BATCH_H

#include "cobs_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cobs_gpu_batch cobs_gpu_batch;   /* device workspace of one query batch */
typedef struct cobs_gpu_comm cobs_gpu_comm;     /* one rank of an RCCL communicator (multi-GPU exchange) */

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

/* ---- procedural index ---------------------------------------------------- */
cobs_gpu_status cobs_gpu_open_synthetic(const cobs_gpu_synth* desc,
                                        const cobs_gpu_options* opts, cobs_gpu_index** out);
/* The procedural index of cobs_gpu_open_synthetic written as a .cobs_classic / .cobs_compact FILE
 * (the generator tool of SURVEY 8f rank 2, cf. `cobs classic-construct-random`, src/cobs.cpp:243-291):
 * rows are produced on the device chunk by chunk and streamed to the file. */
cobs_gpu_status cobs_gpu_write_synthetic(const cobs_gpu_synth* desc, const char* out_path, int device);

/* True positives for a RESIDENT index (the procedural one above, or any index that is not streamed): the documents
 * docs[0..ndocs) of file `file_no` additionally contain the terms of `text` -- document docs[i] holds term t (the
 * term_size characters from position t) iff mix64(salt ^ (uint64_t)docs[i] << 32 ^ t) % 1000 < keep_permille[i]
 * (mix64 = the splitmix64 finaliser the procedural bits use) -- and a held term sets, for each of the index's hash
 * functions, the bit of the document in row hash % S_p of its sub-index, exactly what index construction does for a
 * document's own terms (construction/classic_index.cpp:40-73).  Random bits alone give counts ~ Binomial(T, 0.3): no
 * query ever reaches the CLI's default threshold 0.8; with planted documents the thresholded paths --
 * selection, hit pool, D2H of hits, ranking, the hit exchange -- carry data at full size.  Documents a shard does not
 * hold are skipped (every rank plants what it holds).  The test suite's checker restates the rule. */
cobs_gpu_status cobs_gpu_plant(cobs_gpu_index* ix, size_t file_no, const char* text, size_t len, const uint32_t* docs,
                               const uint32_t* keep_permille, size_t ndocs, uint64_t salt);

/* cobs_gpu_search_batch with the results in memory the LIBRARY owns: *hits / *hit_offsets (nq + 1 entries) point into a
 * result arena kept on the handle -- grown on demand, its pages faulted in once and reused by every later call (huge pages
 * where the host offers them on request) -- and stay valid until the next search call on this handle.  The form for a
 * caller that would otherwise allocate a fresh result array per call: the reference's default call returns one record per
 * document and query (classic_search.cpp:450), 307 MB for 256 queries x 100 000 documents, and a fresh array costs
 * 75 000 first-touch page faults per call.  (The reference's own callers keep ONE result vector: src/cobs.cpp:618-626.)
 * It is also the form for a THRESHOLDED call whose number of hits the caller cannot guess: the arena grows while the
 * passes of the call come home, where cobs_gpu_search_batch can only report COBS_GPU_ERR_CAPACITY after the whole search
 * has run and be called a second time. */
cobs_gpu_status cobs_gpu_search_batch_view(cobs_gpu_index* ix, const char* const* queries, const size_t* lens, size_t nq,
                                           double threshold, size_t num_results, const cobs_gpu_hit** hits,
                                           const size_t** hit_offsets, size_t* bad_query);

/* score slots per query held by THIS shard (== cobs_gpu_total_counts when
 * unsharded); device count rows have this many elements */
uint64_t cobs_gpu_local_counts(const cobs_gpu_index* ix);

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
/* `cobs benchmark-fpr --dist` (src/cobs.cpp:627-632: counts[r.score]++ over every result of every query): after a run
 * that kept score rows, hist[s] += the (query, real document) pairs of the batch with score s (s >= nbins: the last
 * bin); tallied on the device.  On a shard: of the documents it holds. */
cobs_gpu_status cobs_gpu_batch_score_histogram(cobs_gpu_batch* b, uint64_t* hist, size_t nbins);
/* D2H + rank the hits of query `query_no` of the last run */
cobs_gpu_status cobs_gpu_batch_hits_host(cobs_gpu_batch* b, size_t query_no, size_t num_results,
                                         cobs_gpu_hit* hits, size_t cap, size_t* n_hits);

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
/* Failure behaviour of a communicator (the reference has no distributed code; this is the contract of the layer added
 * here): an RCCL call that fails marks the communicator BROKEN -- an open ncclGroupStart is closed first, a dead
 * communicator is aborted (ncclCommAbort) -- and every later call on it fails at once with COBS_GPU_ERR_RCCL on this
 * rank, before any collective.  A collective that a peer never enters does not fail, it waits: */
/* ... with a time limit, the stream waits the library itself performs around collectives (layout and size exchanges,
 * status agreements, the row exchanges of cobs_gpu_sharded_search_batch) give up after timeout_ms, abort the
 * communicator and return COBS_GPU_ERR_RCCL.  0 = wait for ever (the default, unless COBS_GPU_COMM_TIMEOUT_MS is set in the
 * environment when the communicator is created: the limit of communicators the library makes itself, cobs_gpu_multi_open). */
void cobs_gpu_comm_set_timeout(cobs_gpu_comm* c, uint32_t timeout_ms);
/* ... and what this rank entered last, as one line of text (RCCL calls entered / returned, the last call, whether its
 * stream is idle) -- callable from ANOTHER thread while the owner sits in a call: what a caller's watchdog prints
 * when a step does not come back.  -> characters written (NUL-terminated). */
size_t cobs_gpu_comm_state(const cobs_gpu_comm* c, char* buf, size_t cap);
/* Collective.  The first bytes a new communicator moves, each step under `timeout_ms` and every received byte checked:
 * one grouped ncclSend / ncclRecv all-to-all with a different size for every (sender, receiver) pair, one
 * ncclAllGather, ncclAllReduce(max, sum) -- the operations a batch exchange uses; big_bytes > 0 adds a timed all-to-all
 * of that many bytes per pair.  out = all-to-all bytes received | its microseconds | all-gather us | all-reduce us |
 * large all-to-all bytes received | its us (second round) | 0 | 0.  A failure leaves the communicator broken. */
cobs_gpu_status cobs_gpu_comm_preflight(cobs_gpu_comm* c, uint32_t timeout_ms, uint64_t big_bytes, uint64_t out[8]);

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
/* After a run with num_results > 0: all-gather the k best documents of every shard and merge them on the device into
 * the global k best of every (file, query) (K3 over the gathered lists; k > 8192: merged per query on the host);
 * cobs_gpu_batch_hits_host then answers from them.  Not for a query with a single hash in total: the reference does not order
 * such a result by score (max_counts <= 1, classic_search.cpp:134,177), it is the first documents in index order,
 * which per-shard best-of lists do not determine -- exchange the score rows for such a batch
 * (cobs_gpu_batch_exchange_counts; cobs_gpu_sharded_search_batch does). */
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

/* How both calls above run (round 6): the call is cut into passes that overlap inside the library, as the reference
 * parallelises inside search() (parallel_for over document batches, classic_search.cpp:355-400): upload + hashing of
 * pass i+1 | scan of pass i | the ranks' agreement, the exchange over RCCL and the ordering of the results of pass i-1,
 * on their own streams tied by events.  After a scan the ranks agree through ONE all-gathered record per pass (status,
 * first invalid query, hit-pool fill; written by the device, read once per pass while the next pass scans).  All ranks
 * must have set the tuning keys that cut passes (pass_bytes, pipe_chars) alike. */

/* ---- the device-resident form of the sharded search: what `bench.py --gpus N` times ----
 * ONE batch of queries, uploaded once; a step scans it on every rank against that rank's shard and leaves the count rows
 * on the device in global document order (cobs_gpu_batch_global_counts_device of every sub-batch).  The batch is cut
 * into `sub_batches` sub-batches over the queries -- the reference's own loop is per batch of documents,
 * classic_search.cpp:355-400 -- whose hashing (the sub-batch's own stream), scan (scan stream) and exchange (exchange
 * stream) overlap, also across steps: hash(i+1) | scan(i) | exchange(i-1).  Collective: every rank makes the same calls.
 * More than one sub-batch sets the handle's tuning key hash_stream. */
typedef struct cobs_gpu_sharded_batch cobs_gpu_sharded_batch;
cobs_gpu_status cobs_gpu_sharded_batch_create(cobs_gpu_index* ix, cobs_gpu_comm* c, uint32_t sub_batches,
                                              cobs_gpu_sharded_batch** out);
void cobs_gpu_sharded_batch_destroy(cobs_gpu_sharded_batch* sb);
/* sub-batch i holds the queries [nq*i/S, nq*(i+1)/S); synchronous (one upload per sub-batch) */
cobs_gpu_status cobs_gpu_sharded_batch_set_queries(cobs_gpu_sharded_batch* sb, const char* const* queries, const size_t* lens,
                                                   size_t nq);
/* one step, asynchronous: every sub-batch hashed, scanned, its count rows exchanged (mode: cobs_gpu_exchange_mode) */
cobs_gpu_status cobs_gpu_sharded_batch_step(cobs_gpu_sharded_batch* sb, double threshold, uint32_t mode);
/* waits for everything queued (the exchange stream under the communicator's time limit); invalid queries are reported
 * here, *bad_query = index in the batch */
cobs_gpu_status cobs_gpu_sharded_batch_sync(cobs_gpu_sharded_batch* sb, size_t* bad_query);
size_t cobs_gpu_sharded_batch_subs(const cobs_gpu_sharded_batch* sb);
/* sub-batch i (owned by sb) and its queries: for the batch-level accessors (global / local count rows, stats) */
cobs_gpu_batch* cobs_gpu_sharded_batch_sub(cobs_gpu_sharded_batch* sb, size_t i, size_t* q_begin, size_t* q_count);
/* After a sync; per step, summed over the sub-batches, averaged over the steps since the previous call (at most 64):
 * out = scan ms | hash ms | exchange ms | algorithmic bytes | bytes received from other ranks |
 * scan launches | steps averaged over | 0 -- HIP events on the streams the kernels and collectives ran on. */
cobs_gpu_status cobs_gpu_sharded_batch_times(cobs_gpu_sharded_batch* sb, double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* COBS_GPU_BATCH_H */
