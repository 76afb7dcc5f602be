"""cobs_amd -- MI355X (gfx950) query engine for COBS bit-sliced signature indexes.

Drop-in for ONE path of bingmann/cobs: `cobs_index.Search(path).search(query,
threshold, num_results)` (reference python/module.cpp:367-386) /
`cobs::ClassicSearch::search` (reference cobs/query/classic_search.cpp:403-505).
All compute happens in libcobs_gpu.so (HIP, C ABI in include/cobs_gpu.h).
"""
from ._capi import CobsGpuError  # noqa: F401
from .construct import (ClassicIndexParameters, CompactIndexParameters, DocumentList,  # noqa: F401
                        classic_construct, classic_construct_list, compact_construct,
                        compact_construct_list, disable_cache, write_synthetic, build_search,
                        classic_combine, compact_combine, classic_construct_random, DocumentEntry, FileType)
from .search import Batch, MultiSearch, Search, SearchResult, ShardedBatch  # noqa: F401

__version__ = "0.2.0"
__all__ = ["Search", "MultiSearch", "SearchResult", "Batch", "ShardedBatch", "CobsGpuError", "DocumentList", "DocumentEntry", "FileType",
           "ClassicIndexParameters",
           "CompactIndexParameters", "classic_construct", "classic_construct_list", "compact_construct",
           "compact_construct_list", "disable_cache", "write_synthetic", "build_search", "classic_combine", "compact_combine",
           "classic_construct_random", "__version__"]
