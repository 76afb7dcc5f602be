"""`import cobs_index` -- the reference's Python module name (python/module.cpp:31,
`PYBIND11_MODULE(cobs_index, m)`) served by the MI355X engine: a script written against
bingmann/cobs' Python API runs unchanged with this directory on its path.  Everything is
re-exported from cobs_amd; compute happens in libcobs_gpu.so (HIP, no CPU fallback)."""
from cobs_amd import (ClassicIndexParameters, CompactIndexParameters, DocumentEntry, DocumentList,  # noqa: F401
                      FileType, Search, SearchResult, __version__, classic_construct, classic_construct_list,
                      compact_construct, compact_construct_list, disable_cache)
from cobs_amd.construct import (Any, Cortex, Fasta, FastaMulti, Fastq, FastqMulti, KMerBuffer,  # noqa: F401
                                Text)      # py::enum_<FileType>::export_values() (module.cpp:110-127)

__all__ = ["disable_cache", "FileType", "DocumentEntry", "DocumentList", "ClassicIndexParameters", "classic_construct",
           "classic_construct_list", "CompactIndexParameters", "compact_construct",
           "compact_construct_list", "SearchResult", "Search", "__version__"]
