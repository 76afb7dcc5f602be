// cobs_amd/csrc/device_types.hpp -- structures shared by host code and HIP kernels.
#pragma once
#include <cstdint>

struct cobs_gpu_hit;

namespace cobs_amd {

// One sub-index ("page" in the reference's compact index; a classic index is a
// single page) as laid out in HBM.  Rows are `pitch` bytes apart (multiple of
// 16), row `sig` (one past the last signature row) is all zero and is what
// padded query terms point at.
struct PageDev {
    uint64_t base;         // byte offset of row 0 from the file's HBM blob
    uint64_t sig;          // signature_size S_p (row index = hash % S_p)
    uint64_t magic;        // floor((2^64 - 1) / S_p) for the exact fast modulo
    uint64_t row0;         // streamed row-range chunk: `sig` rows starting at row0 of the sub-index are in the buffer (else 0)
    uint32_t slot0;        // first local score slot of this page (multiple of 8)
    uint32_t doc0;         // file-level document id of the page's first document
    uint32_t valid_bytes;  // row bytes that map to score slots (<= pitch)
    uint32_t tpage;        // index of this sub-index in the part-wide row-index table (K1 output)
};

// Arguments of the hashing kernel K1 for one index file.
struct HashArgs {
    const uint8_t* text;        // query characters, query q at text[span_off[q] ..]
    const uint64_t* span_off;   // nq + 1 prefix sums of per-query thread spans
    const uint32_t* q_len;      // characters per query
    const uint64_t* blk_off;    // nq + 1 prefix sums of 8-term blocks per query
    const PageDev* pages;       // the file-level sub-indexes this part holds (table pages): sig / magic are used
    void* table;                // row indices (u32, or u64 when idx64): [q][page][block (nblk + 1 padding block)][hash][8]
    uint32_t* err_query;        // atomicMax of (2^32-1 - q) over queries q holding a non-ACGT base; 0 = none
    uint32_t nq;
    uint32_t npages;
    uint32_t term_size;
    uint32_t canonicalize;
    uint32_t num_hashes;
    uint32_t idx64;             // some sub-index has >= 2^32 - 1 rows: 64-bit table entries
};

// One selected (query, document) pair of the on-device threshold pass.
struct HitDev {
    uint32_t query;
    uint32_t part;
    uint32_t doc;
    uint32_t score;
};

// Arguments of the scan kernel K2 for one index file.
struct ScanArgs {
    const uint8_t* blob;        // HBM blob of the file
    const PageDev* pages;
    const void* table;          // from K1 (u32 entries, u64 when idx64)
    const uint64_t* blk_off;    // nq + 1
    void* counts;               // u8, u16 or u32 [nq][counts_stride]
    const uint32_t* thresholds; // per query (this file) or nullptr = no selection
    HitDev* hits;               // selection pool
    unsigned long long* hit_count;   // pool fill (may exceed hit_cap: overflow; 64-bit so that it cannot wrap)
    uint64_t counts_stride;     // elements per query row
    uint64_t counts_offset;     // local slot offset of this file inside a row
    uint32_t hit_cap;
    uint32_t nq;
    uint32_t npages;
    uint32_t table_npages;      // sub-indexes in the row-index table (its page stride); PageDev::tpage selects one
    uint32_t pitch;             // bytes between rows
    uint32_t cpp;               // 16-byte chunks per row (pitch / 16)
    uint32_t total_chunks;      // npages * cpp
    uint32_t num_hashes;
    uint32_t num_docs;          // documents of the file (ids >= this are padding)
    uint32_t part;              // file number
    uint32_t write_counts;      // 0: selection only
    uint32_t tile_w;            // 16-byte chunks per tile: 4, 8, 16, 32 or 64
    uint32_t chunk_begin;       // this launch covers column chunks [chunk_begin, chunk_end)
    uint32_t chunk_end;
    uint32_t idx64;             // 64-bit row indices in the table
    uint32_t lds_staged;        // measured variant: rows travel HBM -> LDS -> VGPR (global_load_lds)
    uint32_t exp;               // bit field of experimental variants under A/B measurement (tuning key "exp")
    // run_topk without score rows: every tile leaves its topk_k best (document, score) candidates at
    // cand[query * cand_stride + (tile_base + tile) * topk_k ..]; nullptr = the other epilogues
    uint2* cand;
    uint32_t topk_k, cand_stride, tile_base;
    // tuning builds (COBS_SCAN_TIMING) only: s_memtime stamps [slot][wave 0..3][8] of every dbg_every-th work-group
    uint64_t* dbg;
    uint32_t dbg_every, dbg_slots;
};

// Owner-routed hit exchange (xchg_kernels.hip): bucket the hit pool by the rank that owns each record's query.
struct BucketArgs {
    const HitDev* hits;
    uint64_t n;
    unsigned long long* cursor;  // [nranks]: counting pass: zeroed, receives the counts; scatter pass: the buckets' start positions
    HitDev* out;                 // scatter pass: the bucketed pool
    uint32_t nq, nranks;
};

// The hit pool of a thresholded pass put into RESULT order on the device (xchg_kernels.hip, results.cpp: order_pool):
// records bucketed by query (count, scan, scatter), every query's bucket ordered as counts_to_result orders it
// (score descending, then (file, document) ascending: classic_search.cpp:136-145; index order for a query with a single
// hash in total, :134) -- the host then copies finished lists instead of sorting one query at a time.
struct PoolArgs {
    const HitDev* in;            // the pool, in the order the scan's atomics appended it
    HitDev* tmp;                 // bucketed by query
    HitDev* out;                 // ... and every bucket ordered
    uint32_t* cnt;               // [nq + 1] records per query (zeroed before the count)
    uint32_t* off;               // [nq + 1] first record of every query (exclusive scan of cnt)
    uint32_t* cur;               // [nq] scatter cursors (zeroed)
    const uint8_t* single;       // [nq] 1 = the query has a single hash in total (index order), or nullptr: none has
    uint32_t n, nq;
    uint32_t seg_max;            // buckets beyond this many records are left in pool order (the host orders those)
};

// Row-selective access to a chunk that is not resident (fetch_kernels.hip): fetch the rows K1's table names
// from the registered file mapping into a gathered buffer shaped like a resident chunk.
// Row-range chunks of a streamed sub-index (fetch_kernels.hip): K1's row indices of one sub-index rewritten for a
// buffer that holds rows [row0, row0 + nrows) only, and the partial scores of such a chunk added to the score rows.
struct RemapArgs {
    const void* table;           // K1's row indices (u32, or u64 when idx64)
    void* table2;                // same layout: index - row0 inside the range, else nrows (the buffer's zero row)
    const uint64_t* blk_off;     // nq + 1
    uint64_t row0, nrows;
    uint32_t nq, tpage, table_npages, num_hashes;
};
// The terms a row-range unit actually HOLDS, per query (compact_*_kernel): a range's scan walks every term of every query
// while 99 % of them name the zero row (their row lies in another range of the sub-index) -- 4.4 ms per unit for 10 000
// queries x 1000 terms, issue-bound, where the in-range terms alone are a 12-term "read".  A second table with the same
// layout but each query's in-range entries only, and its own block offsets.  One hash function, 32-bit indices.
struct CompactArgs {
    const uint32_t* table2;      // the unit's row-index table (entries naming no row == zero_idx)
    uint32_t* table3;            // compact: [query][sub-index][blocks2 + 1][8] (only sub-index `tpage` is written)
    const uint64_t* blk_off;     // nq + 1: blocks per query of table2
    uint64_t* blk2;              // nq + 1: ... of table3 (exclusive scan of `cnt`)
    uint32_t* cnt;               // [nq + 1] blocks of in-range terms per query
    uint32_t nq, tpage, table_npages, zero_idx;
};
struct AddScoresArgs {
    void* dst;                   // score rows [nq][dst_stride] (elements of elem_bytes)
    const void* src;             // partial scores [nq][nslots]
    uint64_t dst_stride, dst_offset;   // elements
    uint32_t nslots, nq, elem_bytes;
};
// counts_to_result's filter (reference classic_search.cpp:127-132) over the ACCUMULATED scores of one sub-index whose row
// ranges were scanned one after the other (select_rows_kernel): what K2's epilogue does for a sub-index it sees whole.
struct SelectRowsArgs {
    const void* scores;          // [nq][stride] scores of ONE sub-index slice (elements of elem_bytes), slot 0 = document doc0
    const uint32_t* thresholds;  // [nq] ceil(threshold * T)
    HitDev* hits;                // the batch's hit pool ...
    unsigned long long* hit_count;   // ... and its 64-bit fill
    uint64_t stride;             // elements between the queries' rows (a multiple of 8)
    uint32_t nslots;             // slots that belong to the slice (valid row bytes x 8)
    uint32_t nq, elem_bytes;
    uint32_t doc0, num_docs;     // file-level document of slot 0 | real documents of the file
    uint32_t part, hit_cap;
};

// How many rows a batch LOOKS UP in every streamed piece of a file (count_rows_kernel): counter first + min(row / per,
// n - 1) of the entry's sub-index (per == 0: one counter for the sub-index); first == 0xFFFFFFFF: not counted (resident).
struct CountPage { uint64_t per; uint32_t first, n; };
struct CountArgs {
    const void* table;           // K1's row indices (u32, or u64 when idx64)
    const uint64_t* blk_off;     // nq + 1
    const PageDev* tpages;       // the part's sub-indexes (sig: a row index >= sig is K1's padding)
    const CountPage* cpages;     // [table_npages]
    unsigned long long* counts;  // [ncounters], zeroed
    uint32_t nq, table_npages, num_hashes, ncounters;
};

// The row-selective gather (gather_assign_kernel, gather_copy_kernel): exactly the looked-up rows of a unit's pages, packed.
// A page = one slice of a sub-index (all of its rows, or a row range); its gathered rows sit at rows [slot0, slot0 + count)
// of the buffer, its zero row at slot0 + count.
struct GatherPage {
    uint64_t src;                // file offset of (first row of the page, first held column)
    uint64_t row0, nrows;        // rows of the sub-index the page covers
    uint64_t slot0;              // first gathered row (rows of `pitch` bytes)
    uint32_t count;              // looked-up rows in [row0, row0 + nrows): exact (count_rows_kernel)
    uint32_t tpage;              // sub-index in the row-index table
    uint32_t leader;             // the page of this unit whose row list this page uses (column slices of one sub-index share it)
    uint32_t valid_bytes;        // row bytes of the slice (the rest of the pitch reads as zero)
    uint64_t bm_off;             // leader pages: first word of the page's row bitmap (one bit per row of [row0, row0 + nrows)) ...
    uint32_t bm_words;           // ... and its length in 32-bit words
    uint32_t reserved;
};
struct GatherArgs {
    const uint8_t* file;         // device-visible address of the mapped index file
    const void* table;           // K1's row indices
    void* table2;                // same layout: the gathered row of every entry (page-relative; count = the page's zero row)
    const uint64_t* blk_off;     // nq + 1
    const GatherPage* pages;
    const PageDev* pages_in;     // the unit's pages as a resident chunk would hold them (slot0, doc0, ... are copied)
    PageDev* pages2;             // ... as the gathered buffer holds them (written here)
    uint8_t* dst;                // the gathered rows
    uint64_t* rowlist;           // [total slots] source row (relative to the page's row0) of every gathered row
    unsigned long long* cursor;  // [npages]: DISTINCT looked-up rows of every leader page = slots in use (gather_rank_kernel)
    uint32_t* bitmap;            // the leader pages' row bitmaps (zeroed): bit r = some entry looks up row r (gather_mark_kernel)
    uint32_t* bprefix;           // same shape: set bits in front of every word -- a row's slot is bprefix + the bits below it
    unsigned long long* fetched_bytes;   // accounting: += (distinct rows x pitch) of every page (what the unit asks of PCIe)
    uint64_t entries;            // table entries per sub-index: (blk_off[nq] + nq) * 8 * num_hashes
    uint64_t total_rows;         // gathered rows incl. the pages' zero rows
    uint64_t src_pitch;          // bytes between rows in the file
    uint32_t nq, npages, table_npages, num_hashes, pitch;
    uint32_t exp;                // experimental variants of the copy under A/B measurement (COBS_GPU_GATHER_EXP: 1 = non-temporal loads, 2 = two pieces per thread in flight)
};

// Arguments of the top-k selection kernel K3 for one index file.
struct TopkArgs {
    const void* counts;          // [nq][counts_stride] scores of score_bytes (1 or 2) bytes
    const uint32_t* thresholds;  // per query or nullptr (= 0)
    uint2* out;                  // [nq][k] (doc, score); ordered (score desc, doc asc) when at most sort_limit survive
    uint32_t* out_count;         // [nq] entries written (<= k)
    uint64_t counts_stride;
    uint64_t counts_offset;      // first local slot of this file in a row
    uint32_t nslots;             // local slots of this file
    uint32_t doc_base;           // file-level document id of local slot 0
    uint32_t num_docs;           // real documents of the file
    uint32_t k;
    uint32_t nq;
    uint32_t score_bits;         // scores are < 2^score_bits (the scan kernel's plane count, <= 32)
    uint32_t level_bits;         // histogram bits per radix level (<= 12)
    uint32_t levels;             // ceil(score_bits / level_bits): 1, 2 or 3
    uint32_t score_bytes;        // 1 (planes <= 8), 2 (<= 16) or 4
    uint32_t sort_limit;         // order the survivors on the device when there are at most this many (0 = never)
    uint32_t out_stride;         // entries between the queries' outputs (0 = k)
    uint32_t pad_out;            // != 0: the entries of `out` beyond the survivors are marked unused (document 0xFFFFFFFF) --
                                 // the output is ONE TILE of K2's candidate pool (a sub-index counted in row ranges)
    uint32_t from_pool;          // counts = candidate pool of K2 ([nq][counts_stride] (document, score) entries, the first
                                 // nslots in use; counts_stride a multiple of 8) instead of score rows
};

// Ranking of all documents of a query on the device (rank_kernels.hip): one index file as the
// ranked score row holds it.
struct RankPart {
    const uint32_t* thr;         // [batch queries] thresholds of this file, or nullptr (= 0)
    uint32_t slot0;              // first slot of the file's slice in the row
    uint32_t doc_first;          // file-level id of the document at that slot
    uint32_t ndocs;              // real documents among the slice's slots
    uint32_t file_no;
};

// One pass of the stable radix sort by score (most passes: one).  Block b ranks query q0 + b.
struct RankArgs {
    const void* rows;            // score rows [batch queries][row_stride] of score_bytes each
    uint64_t row_stride;         // elements
    const RankPart* parts;       // nparts files, slices back to back in slot order
    const uint8_t* by_score;     // [batch queries] 0: a single hash in total, the result keeps index order
    const uint2* src;            // (score, slot) pairs of the previous pass [nq][pair_stride]
    uint2* dst;                  // ... of this pass, unless it is the last
    uint32_t* npass;             // [nq] passing documents: written by a first pass that is not the last, read by later ones
    void* out;                   // last pass: results [nq][out_stride], at most `limit` per query: (slot of the ranked row, score) as
                                 // uint2, or -- pack_bits > 0 -- as ONE u32 = score << pack_bits | slot (slot < 2^pack_bits)
    uint32_t* out_count;         // last pass: [nq] results written = min(limit, passing documents)
    uint64_t pair_stride, out_stride;
    uint32_t nparts, nslots;     // slots of a row that are ranked (the files' slices)
    uint32_t q0, nq;
    uint32_t row_q0;             // batch query whose row is rows[0] (0 for a batch's own rows; the first owned query of exchanged rows)
    uint32_t shift, bits;        // digit of this pass = (score >> shift) & (2^bits - 1), bits <= 12
    uint32_t limit;
    uint32_t score_bytes;
    uint32_t pack_bits;          // 0: 8-byte records
    uint32_t* seg_hist;          // a single pass cut into segments: [nq][4 * nseg][2^bits] (rank_kernels.hip: SEG), else null
    uint32_t nseg;               // segments per row (<= 1: one work-group per query)
    uint32_t* bin_count;         // a single pass (first and last) only, may be null: [nq][2^bits] records per bin, bin =
                                 // 2^bits - 1 - score -- with them the ORDER of the slots alone tells every record's score
};

// The ranked records of full lists as a bit stream of slots (rank_kernels.hip: pack_slots_kernel): record p of query q --
// in[q * in_stride + p] = score << slot_bits | slot -- becomes bits [p * slot_bits, (p + 1) * slot_bits) of the query's
// `words` dwords; the scores travel as RankArgs::bin_count.  C3: 17 bits per result over PCIe instead of 32.
struct SlotPackArgs {
    const uint32_t* in;
    uint32_t* out;               // [nq][words]
    uint64_t in_stride;
    uint32_t n;                  // records per query
    uint32_t words;              // dwords per query: >= ceil(n * slot_bits / 32)
    uint32_t slot_bits;          // 1 .. 31
    uint32_t nq;
};

// The distribution of the scores of a pass (rank_kernels.hip: score_hist_kernel) -- what `cobs benchmark-fpr --dist`
// tallies result by result on the host (src/cobs.cpp:627-632): hist[s] += (query, real document) pairs with score s.
struct HistRange { uint32_t begin, end; };       // local slots of one file's REAL documents (padding slots excluded)
struct HistArgs {
    const void* rows;            // score rows [nq][row_stride]
    uint64_t row_stride;         // elements
    const HistRange* ranges;
    unsigned long long* hist;    // [nbins]; scores >= nbins fall into the last bin
    uint32_t nranges, nq, nbins, score_bytes;
};

// Arguments of the construction kernel: set the signature bits of documents.
// a stretch of term text whose k-grams are all terms (newlines included); otherwise a stretch is
// sequences each followed by '\n' and no term holds a '\n'
constexpr uint32_t kBuildRawStretch = 0x80000000u;
// a stretch that holds no term text at all (the unused rest of a document's span of the staging buffer)
constexpr uint32_t kBuildGapStretch = 0xFFFFFFFFu;

struct BuildArgs {
    const uint8_t* text;        // stretches of term text back to back (documents.hpp); the buffer is
                                // padded by 8 readable bytes
    const uint64_t* seg_off;    // nsegs + 1 offsets into text
    const uint32_t* seg_col;    // nsegs: the document's column (slot) in the matrix | kBuildRawStretch
    uint32_t* matrix;           // signature_size rows of row_bytes (multiple of 4) bytes, as words
    uint64_t signature_size;
    uint64_t magic;             // floor((2^64-1) / signature_size)
    uint64_t row_bytes;
    uint32_t nsegs;
    uint32_t term_size;
    uint32_t canonicalize;
    uint32_t num_hashes;
    // byte-map mode (bytemap != nullptr): instead of an atomicOr into the matrix a term stores the
    // byte 1 at bytemap[(column - col_base) * bm_stride + row] -- one plane of signature_size bytes
    // per document of the launch; pack_bytemap_kernel folds the planes into the matrix afterwards
    uint8_t* bytemap;
    uint64_t bm_stride;
    uint32_t col_base;
};

// byte-map planes of one build launch -> bits of the matrix
struct PackArgs {
    const uint8_t* bytemap;     // ndocs planes of bm_stride bytes (0 / 1), plane d = column col_base + d
    uint64_t bm_stride;         // multiple of 256
    uint32_t* matrix;           // rows of row_bytes (multiple of 4) bytes, as words
    uint64_t row_bytes;
    uint64_t rows;              // signature_size
    uint32_t col_base, ndocs;
};

// classic_construct_random: documents of random 31-mers
struct RandomBuildArgs {
    uint32_t* matrix;           // signature_size rows of row_bytes (multiple of 4) bytes, as words
    uint64_t signature_size, magic, row_bytes;
    uint64_t doc0;              // first document (column) of this launch
    uint64_t document_size;     // 31-mers per document
    uint64_t seed;
    uint32_t num_hashes;
};

// classic_combine: bit-level concatenation of the rows of several matrices
struct CombineArgs {
    const uint8_t* const* src;      // nsrc row blocks (device pointers, device array)
    const uint64_t* src_row_bytes;  // nsrc
    const uint64_t* bit_off;        // nsrc + 1 prefix sums of the inputs' document counts
    uint8_t* dst;
    uint64_t dst_row_bytes;
    uint64_t rows;
    uint32_t nsrc;
};

// procedural index fill
struct SynthArgs {
    uint8_t* blob;
    const PageDev* pages;
    uint64_t seed;
    uint64_t row_bytes;         // file-level bytes per row of a sub-index
    uint64_t col0;              // first file-level row byte held locally (classic column shards)
    uint64_t num_docs;
    uint64_t page_docs;         // documents per sub-index (compact: 8*page_size, classic: all)
    uint32_t npages;
    uint32_t first_page;        // file-level number of local page 0
    uint32_t pitch;
};

// True positives planted into a resident procedural index (cobs_gpu_plant, include/cobs_gpu_batch.h): the terms of one
// text become terms of a list of documents, as index construction would set them (classic_index.cpp:40-73).
struct PlantDoc {
    uint8_t* col;               // address of (row 0, the byte that holds the document's bit); nullptr: not held by this shard
    uint64_t sig;               // rows of the document's sub-index
    uint32_t pitch;             // bytes between its rows
    uint32_t bit;               // doc % 8
    uint32_t doc;               // file-level document number (enters the keep rule)
    uint32_t keep_permille;     // share of the text's terms this document holds
};
struct PlantArgs {
    const uint8_t* text;
    const PlantDoc* docs;
    uint64_t salt;
    uint32_t len, term_size, canonicalize, num_hashes, ndocs;
    uint32_t* bad;              // set to 1 if the text holds a character outside ACGT (canonicalize != 0): nothing is planted for such terms
};

// rows of one sub-index of the procedural index, for the file writer
struct SynthRowsArgs {
    uint8_t* dst;               // row r at dst + r * pitch
    uint64_t seed;
    uint64_t row0, nrows;
    uint64_t row_bytes;         // valid bytes per row (rest of the pitch is zero)
    uint64_t live_docs;         // real documents of this sub-index (later bits are zero)
    uint32_t page;              // file-level sub-index
    uint32_t pitch;             // multiple of 8
};

struct RepitchArgs {
    const uint8_t* src;         // staged raw rows: row r at src + r * src_pitch
    uint8_t* dst;               // dst row r at dst + r * dst_pitch
    uint64_t rows;
    uint32_t src_pitch;
    uint32_t dst_pitch;
    uint32_t copy_bytes;        // bytes copied per row (rest of dst row zeroed)
    uint32_t src_col0;          // first source byte of each row
};

}  // namespace cobs_amd
