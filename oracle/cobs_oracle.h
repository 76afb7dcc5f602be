/*
 * oracle/cobs_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the bingmann/cobs query path
 * (cobs::ClassicSearch::search and everything below it).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; nothing under cobs_amd/ links, imports or calls it.
 *
 * Parity status: PINNED against the reference's own known answers
 * (tests/test_oracle_pins.py): canonicalize_kmer KATs of
 * reference tests/util.cpp:38-60, the exact-score assertions of
 * python/tests/test_cobs_index.py:22-61 on the reference's tests/data/fasta
 * corpus, the "== 1" / ">= truth" assertions of tests/classic_index_query.cpp
 * and tests/compact_index_query.cpp, and XXH64 (third-party xxHash, un-vendored
 * submodule extlib/xxhash; algorithm restated from the public XXH64 spec) against
 * the published XXH64 known answers and python-xxhash 3.8.1 (libxxhash 0.8.2).
 * The reference itself is NOT buildable in this image (every translation unit
 * includes the empty extlib/tlx submodule), so there is no oracle/_ref.
 */
#ifndef COBS_ORACLE_H
#define COBS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_index oracle_index;

enum {
    ORACLE_OK = 0,
    ORACLE_ERR_OPEN = 1,
    ORACLE_ERR_FORMAT = 2,
    ORACLE_ERR_QUERY_TOO_SHORT = 3,
    ORACLE_ERR_INVALID_BASE = 4,
    ORACLE_ERR_QUERY_TOO_LONG = 5,
    ORACLE_ERR_GEOMETRY = 6,
    ORACLE_ERR_ARG = 7
};

/* -- hot-path pieces --------------------------------------------------- */
uint64_t oracle_xxh64(const void* data, size_t len, uint64_t seed);
/* returns 1 if all characters were ACGT, else 0 (invalid chars become \0) */
int oracle_canonicalize_kmer(const char* in, char* out, size_t k);
/* std::default_random_engine (minstd_rand0) % 4 -> "ACGT" */
void oracle_random_sequence(char* out, size_t size, uint64_t seed);
/* std::mt19937 stream -> "ACGT"; state carried across calls via *state_io
 * (opaque 624+1 uint32 words, zero-initialise and pass seed on first call) */
void oracle_mt19937_sequence(char* out, size_t size, uint32_t* state_io,
                             uint32_t seed, int first);

/* all (len-k+1) term hashes of a sequence exactly as create_hashes does:
 * out[i*H+j] = XXH64(canon(seq+i), k, j); good[i] = 0 where a term held non-ACGT */
void oracle_term_hashes(const char* seq, size_t len, uint32_t k, int canonicalize,
                        uint64_t num_hashes, uint64_t* out, uint8_t* good);

/* -- index access ------------------------------------------------------ */
int oracle_open(const char* path, oracle_index** out);
/* index over caller-owned memory; kind 0 = classic, 1 = compact.
 * page_data[p] points at sub-index p (classic: one pointer, row stride row_size) */
int oracle_from_memory(int kind, uint32_t term_size, uint8_t canonicalize,
                       uint64_t num_hashes, uint64_t page_size, uint32_t num_pages,
                       const uint64_t* signature_sizes, uint32_t num_docs,
                       const uint8_t* const* page_data, oracle_index** out);
/* procedural synthetic compact/classic index: rows are generated on demand by
 * oracle_synth_fill(); the same generator is implemented in the HIP library. */
int oracle_synthetic(int kind, uint32_t term_size, uint8_t canonicalize,
                     uint64_t num_hashes, uint64_t page_size, uint32_t num_pages,
                     const uint64_t* signature_sizes, uint32_t num_docs,
                     uint64_t seed, oracle_index** out);
void oracle_close(oracle_index* ix);

uint32_t oracle_term_size(const oracle_index* ix);
uint32_t oracle_canonicalize(const oracle_index* ix);
uint64_t oracle_num_hashes(const oracle_index* ix);
uint64_t oracle_page_size(const oracle_index* ix);    /* classic: 1 (as the reference) */
uint64_t oracle_row_size(const oracle_index* ix);
uint64_t oracle_counts_size(const oracle_index* ix);
uint32_t oracle_num_pages(const oracle_index* ix);
uint64_t oracle_signature_size(const oracle_index* ix, uint32_t page);
uint32_t oracle_num_docs(const oracle_index* ix);
const char* oracle_doc_name(const oracle_index* ix, uint32_t doc);
uint64_t oracle_data_offset(const oracle_index* ix);

/* bytes [byte_begin, byte_begin+n) of row `row` of sub-index `page` of the
 * procedural index (kind/page_size/num_docs decide the zero padding). */
void oracle_synth_fill(int kind, uint64_t seed, uint64_t page_size, uint32_t num_pages,
                       uint32_t num_docs, uint32_t page, uint64_t row,
                       uint64_t byte_begin, uint64_t n, uint8_t* out);
/* True positives for a procedural index (oracle_synthetic): documents docs[0..ndocs) additionally hold the terms of
 * `text`, document docs[i] the terms t with mix64(salt ^ (uint64_t)docs[i] << 32 ^ t) % 1000 < keep_permille[i] --
 * the checker's restatement of cobs_gpu_plant (include/cobs_gpu_batch.h); not part of the reference. */
int oracle_plant(oracle_index* ix, const char* text, size_t len, const uint32_t* docs,
                 const uint32_t* keep_permille, size_t ndocs, uint64_t salt);

/* -- search ------------------------------------------------------------ */
/* raw per-document counts of ONE index, length counts_size (incl. padding docs).
 * score_width_out (optional) receives 1/2/4 = the reference's Score type. */
int oracle_counts(oracle_index* ix, const char* query, size_t len, int threads,
                  uint32_t* counts, int* score_width_out);

/* ClassicSearch::search over n indexes.  Results: index number, document id
 * inside that index, score.  cap must be >= sum of counts_size. */
int oracle_search(oracle_index* const* ixs, size_t n, const char* query, size_t len,
                  double threshold, size_t num_results, int threads,
                  uint32_t* out_index, uint32_t* out_doc, uint32_t* out_score,
                  size_t cap, size_t* n_out);

/* CPU-baseline timing loop (all in C): queries q at text[offsets[q]..offsets[q+1]);
 * threshold < 0 = per-document counts only (no threshold filter / ranking) */
size_t oracle_search_many(oracle_index* const* ixs, size_t n, const char* text,
                          const uint64_t* offsets, size_t nq, double threshold,
                          size_t num_results, int threads, double seconds,
                          double* elapsed_out, uint64_t* checksum_out);

/* accumulated phase seconds since last reset: hashes, io, and, add, sort */
void oracle_timers(double out[5], int reset);

const char* oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
