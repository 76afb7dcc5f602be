// cobs_amd/csrc/kernels.hpp -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "device_types.hpp"

namespace cobs_amd {

// K1: one thread per query position (total_threads = span_off[nq]).
hipError_t launch_hash(const HashArgs& a, uint64_t total_threads, hipStream_t stream);

// Smallest instantiated plane count that can hold counts up to max_terms
// (4, 8 -> u8 scores as the reference's T < 255 path; 10, 12, 16 -> u16; 20, 24, 32 -> u32); -1 if none.
int scan_planes_for(uint64_t max_terms);
inline uint32_t scan_score_bytes(int planes) { return planes <= 8 ? 1u : planes <= 16 ? 2u : 4u; }

// K2: ntiles * nq work-groups of nw (1, 2 or 4) waves; multi_query: ntiles * ceil(nq / (64 / tile_w))
// work-groups whose lane groups serve different queries (short queries; see scan_has_multi_query).
hipError_t launch_scan(const ScanArgs& a, uint32_t ntiles, int planes, int nw, bool multi_query,
                       hipStream_t stream);
bool scan_has_multi_query(int planes, uint32_t num_hashes, uint32_t tile_w);
// run_topk without score rows (ScanArgs::cand): the tile_topk epilogue exists for files with 32-bit row indices
bool scan_has_tile_topk(uint32_t num_hashes, bool idx64);
// the LDS-staged A/B variant exists for the headline shape only (H = 1, 10 planes, 2 or 4 waves)
bool scan_has_lds_staged(int planes, uint32_t num_hashes, int nw);

// K3: per query the k best (score desc, doc asc) documents with score >= threshold.
hipError_t launch_topk(const TopkArgs& a, hipStream_t stream);

// Ranking of every document (rank_kernels.hip): one pass of the stable radix sort by score; a.nq work-groups.
hipError_t launch_rank(const RankArgs& a, bool first, bool last, hipStream_t stream);
hipError_t launch_pack_slots(const SlotPackArgs& a, hipStream_t stream);
// Distribution of the scores of a pass over its real documents (rank_kernels.hip): nq x nranges work-groups.
hipError_t launch_score_hist(const HistArgs& a, hipStream_t stream);

// Row-selective out-of-core access (fetch_kernels.hip): one thread per (looked-up row, 16-byte piece).
// count_rows: how many rows the batch looks up in every streamed piece of a part (one counter per piece);
// gather: slots for exactly the DISTINCT looked-up rows of a unit's pages, ascending (gather_mark / _rank / _assign / _list_kernel;
// a.bitmap zeroed), then the rows themselves (gather_copy_kernel).
hipError_t launch_count_rows(const CountArgs& a, uint64_t total_entries, bool idx64, hipStream_t stream);
hipError_t launch_gather_assign(const GatherArgs& a, bool idx64, uint32_t max_words, hipStream_t stream);
// grid_limit: work-groups of the copy (a grid-stride loop): few when other kernels of the pass run beside it, see pass.cpp
hipError_t launch_gather_copy(const GatherArgs& a, uint32_t grid_limit, hipStream_t stream);
// Row-range chunks of a streamed sub-index (fetch_kernels.hip): rewrite the `entries` row indices of one sub-index for
// the rows a chunk holds; add a chunk's partial scores to the score rows.
hipError_t launch_remap_rows(const RemapArgs& a, uint64_t entries, bool idx64, hipStream_t stream);
hipError_t launch_add_scores(const AddScoresArgs& a, hipStream_t stream);
// ... and the threshold filter over the scores a sub-index's row ranges added up (kernels.hip: select_rows_kernel)
hipError_t launch_select_rows(const SelectRowsArgs& a, hipStream_t stream);
// a row-range unit's in-range terms per query: a compact second table + its block offsets (count, scan, write)
hipError_t launch_compact_terms(const CompactArgs& a, hipStream_t stream);
hipError_t launch_clear_flags(uint32_t* flags, hipStream_t stream);

// Owner-routed hit exchange (xchg_kernels.hip): count == true -> records per owner into a.cursor, else scatter.
hipError_t launch_bucket_hits(const BucketArgs& a, bool count, hipStream_t stream);
// The hit pool in result order (xchg_kernels.hip): count per query, scan, scatter, one wave per query orders its bucket.
// a.cnt ([nq + 1]) and a.cur ([nq]) must be zero.
hipError_t launch_order_pool(const PoolArgs& a, hipStream_t stream);
constexpr uint32_t kPoolSegMax = 1024;      // buckets the device orders (larger ones: the host)
// The all-gathered best-of lists of the shards ([rank][lists][k] + counts) side by side per list for K3's pool mode
// ([lists][stride], stride >= nranks * k and a multiple of 8; unused entries marked).
hipError_t launch_merge_lists(const uint2* all, const uint32_t* cnt, uint2* out, uint32_t nranks, uint64_t lists, uint32_t k,
                              uint32_t stride, hipStream_t stream);
// One rank's record of a sharded pass (sharded.cpp): status | first invalid query word | hit-pool fill | extra, 4 x u64.
hipError_t launch_pass_meta(const uint32_t* flags, uint64_t* rec, uint64_t status, uint64_t extra, hipStream_t stream);

// Index construction: one thread per text position hashes its term and sets the bits.
hipError_t launch_build(const BuildArgs& a, uint64_t total_bytes, hipStream_t stream);
hipError_t launch_pack_bytemap(const PackArgs& a, hipStream_t stream);

hipError_t launch_random_build(const RandomBuildArgs& a, uint64_t ndocs, hipStream_t stream);
hipError_t launch_combine(const CombineArgs& a, hipStream_t stream);
hipError_t launch_synth(const SynthArgs& a, hipStream_t stream);
hipError_t launch_synth_rows(const SynthRowsArgs& a, hipStream_t stream);
hipError_t launch_plant(const PlantArgs& a, hipStream_t stream);
hipError_t launch_repitch(const RepitchArgs& a, hipStream_t stream);

}  // namespace cobs_amd
