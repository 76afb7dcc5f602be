"""`import cobs_index` -- the reference's Python module name (python/module.cpp:31,
`PYBIND11_MODULE(cobs_index, m)`) served by the MI355X engine: a script written against
bingmann/cobs' Python API runs unchanged with this directory on its path.  Everything is
re-exported from cobs_amd; compute happens in libcobs_gpu.so (HIP, no CPU fallback)."""
from cobs_amd import (ClassicIndexParameters, CompactIndexParameters, DocumentList, Search,  # noqa: F401
                      SearchResult, __version__, classic_construct, classic_construct_list,
                      compact_construct, compact_construct_list, disable_cache)

__all__ = ["disable_cache", "DocumentList", "ClassicIndexParameters", "classic_construct",
           "classic_construct_list", "CompactIndexParameters", "compact_construct",
           "compact_construct_list", "SearchResult", "Search", "__version__"]
