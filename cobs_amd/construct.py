"""Host-side mirror of the reference's Python construction API (python/module.cpp:106-348):
FileType, DocumentEntry, DocumentList, ClassicIndexParameters, CompactIndexParameters,
classic_construct, compact_construct, their *_list variants, disable_cache -- same names,
arguments and defaults.  The document list, the file readers (text, McCortex, .cobs_doc, FASTA,
FASTQ, multi-FASTA; .gz where the reference reads it) and the ordering / sizing rules of
classic_construct / compact_construct live in libcobs_gpu.so (cobs_amd/csrc/documents.cpp,
build.cpp): host threads parse files while the GPU hashes the terms and sets the signature bits.
This module only binds those entry points.
"""
import ctypes as C
import enum
import os

from . import _capi
from ._capi import BuildParams, check


class FileType(enum.IntEnum):
    """cobs::FileType (document_list.hpp:35-54), values of COBS_GPU_FILETYPE_*"""
    Any = 0
    Text = 1
    Cortex = 2
    KMerBuffer = 3
    Fasta = 4
    Fastq = 5
    FastaMulti = 6
    FastqMulti = 7
    List = 8
    Memory = 9


Any, Text, Cortex, KMerBuffer, Fasta, Fastq, FastaMulti, FastqMulti = (
    FileType.Any, FileType.Text, FileType.Cortex, FileType.KMerBuffer, FileType.Fasta, FileType.Fastq,
    FileType.FastaMulti, FileType.FastqMulti)            # py::enum_::export_values()


def _filetype(ft):
    if isinstance(ft, str):                                  # StringToFileType (document_list.cpp:15-32)
        out = C.c_uint32(0)
        check(_capi.load().cobs_gpu_filetype_from_string(ft.encode(), C.byref(out)))
        return int(out.value)
    return int(ft)


def disable_cache(disable=True):
    """The reference caches document statistics in .cobs_cache files; this engine never writes
    caches, so this is a no-op kept for API compatibility (module.cpp:100-105)."""
    return None


class DocumentEntry:
    """cobs::DocumentEntry (document_list.hpp:62-76): a view of one entry of a DocumentList"""

    def __init__(self, owner, index):
        e = _capi.DocEntry()
        check(owner._lib.cobs_gpu_doclist_entry(owner._h, index, C.byref(e)))
        self._owner, self._index = owner, index
        self.path = os.fsdecode(e.path)
        self.name = e.name.decode("latin-1")
        self.type = FileType(e.type)
        self.size = int(e.size)
        self.subdoc_index = int(e.subdoc_index)
        self.term_size = int(e.term_size)
        self.term_count = int(e.term_count)

    def num_terms(self, k=31):
        """DocumentEntry::num_terms(k) (:85-112): what sizes a signature"""
        n = C.c_uint64(0)
        check(self._owner._lib.cobs_gpu_doclist_num_terms(self._owner._h, self._index, k, C.byref(n)))
        return int(n.value)

    def terms(self, k=31):
        """DocumentEntry::process_terms(k, callback) (:116-151): the terms in callback order"""
        lib, h = self._owner._lib, self._owner._h
        n = C.c_uint64(0)
        check(lib.cobs_gpu_doclist_terms(h, self._index, k, None, 0, C.byref(n)))
        buf = C.create_string_buffer(max(1, int(n.value) * k))
        check(lib.cobs_gpu_doclist_terms(h, self._index, k, buf, int(n.value) * k, C.byref(n)))
        raw = buf.raw
        return [raw[i * k:(i + 1) * k] for i in range(int(n.value))]

    def __repr__(self):
        return "DocumentEntry(%r, %s, size=%d)" % (self.name, self.type.name, self.size)


class DocumentList:
    """cobs::DocumentList (document_list.hpp:154-411): a directory scan (recursive, sorted by path),
    a .list file, single files, or in-memory documents"""

    def __init__(self, root=None, filter=FileType.Any, file_type=None):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        check(self._lib.cobs_gpu_doclist_create(C.byref(self._h)))
        if root is not None:
            self.add_recursive(root, file_type if file_type is not None else filter)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.cobs_gpu_doclist_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def size(self):
        return int(self._lib.cobs_gpu_doclist_size(self._h))

    __len__ = size

    def __getitem__(self, i):
        if not 0 <= i < self.size():
            raise IndexError(i)
        return DocumentEntry(self, i)

    def __iter__(self):
        return (DocumentEntry(self, i) for i in range(self.size()))

    def add(self, path):
        check(self._lib.cobs_gpu_doclist_add(self._h, os.fsencode(path)))

    def add_recursive(self, path, filter=FileType.Any):
        check(self._lib.cobs_gpu_doclist_add_recursive(self._h, os.fsencode(path), _filetype(filter)))

    def add_document(self, name, sequences):
        """in-memory document: a name and its sequences (bytes); no reference counterpart"""
        text = b"\n".join(sequences)
        check(self._lib.cobs_gpu_doclist_add_memory(self._h, name.encode("latin-1"), text, len(text)))

    def sort_by_path(self):
        check(self._lib.cobs_gpu_doclist_sort(self._h, 0))

    def sort_by_size(self):
        check(self._lib.cobs_gpu_doclist_sort(self._h, 1))


class ClassicIndexParameters:
    """construction/classic_index.hpp:29-53 (the fields that affect the result)"""

    def __init__(self):
        self.term_size = 31
        self.canonicalize = 1
        self.num_hashes = 1
        self.false_positive_rate = 0.3
        self.signature_size = 0
        self.mem_bytes = 0
        self.num_threads = 0
        self.clobber = False
        self.continue_ = False
        self.keep_temporary = False


class CompactIndexParameters(ClassicIndexParameters):
    """construction/compact_index.hpp:24-45"""

    def __init__(self):
        super().__init__()
        self.page_size = 0


def _params(p, device):
    b = BuildParams()
    b.struct_size = C.sizeof(BuildParams)
    b.term_size, b.canonicalize, b.num_hashes = p.term_size, p.canonicalize, p.num_hashes
    b.false_positive_rate = p.false_positive_rate
    b.signature_size = getattr(p, "signature_size", 0)
    b.page_size = getattr(p, "page_size", 0)
    b.device = device
    b.text_batch_bytes = int(getattr(p, "text_batch_bytes", 0))
    b.set_bits_mode = int(getattr(p, "set_bits_mode", 0))
    return b


def _as_list(inp, file_type):
    if isinstance(inp, DocumentList):
        return inp
    return DocumentList(inp, file_type)


def _check_output(out_file, ext, params):
    if not out_file.endswith(ext):
        raise ValueError("Error: COBS index file must end with " + ext)
    if os.path.exists(out_file) and not (params.clobber or params.continue_):
        raise FileExistsError("Output file exists, will not overwrite without --clobber")


def classic_construct(input=None, out_file=None, index_params=None, file_type="any", tmp_path="",
                      list=None, device=-1):
    """cobs_index.classic_construct (module.cpp:235-270): documents in list order, one signature
    size from the largest document, file written in the reference's format"""
    params = index_params or ClassicIndexParameters()
    docs = _as_list(list if list is not None else input, file_type)
    _check_output(out_file, ".cobs_classic", params)
    b = _params(params, device)
    check(_capi.load().cobs_gpu_build_classic_list(docs._h, C.byref(b), os.fsencode(out_file)))


def compact_construct(input=None, out_file=None, index_params=None, file_type="any", tmp_path="",
                      list=None, device=-1):
    """cobs_index.compact_construct (module.cpp:314-348): documents sorted by (size, path),
    groups of 8 * page_size documents (path order inside a group)"""
    params = index_params or CompactIndexParameters()
    docs = _as_list(list if list is not None else input, file_type)
    _check_output(out_file, ".cobs_compact", params)
    b = _params(params, device)
    check(_capi.load().cobs_gpu_build_compact_list(docs._h, C.byref(b), os.fsencode(out_file)))


def classic_construct_list(list, out_file, index_params=None, tmp_path="", device=-1):
    """cobs_index.classic_construct_list (module.cpp:253-268): same, from a populated DocumentList"""
    classic_construct(None, out_file, index_params, "any", tmp_path, list=list, device=device)


def compact_construct_list(list, out_file, index_params=None, tmp_path="", device=-1):
    """cobs_index.compact_construct_list (module.cpp:332-347)"""
    compact_construct(None, out_file, index_params, "any", tmp_path, list=list, device=device)


def build_search(input=None, index_params=None, kind="classic", file_type="any", list=None, device=-1):
    """classic_construct / compact_construct straight into a query handle (cobs_gpu_build_index_list):
    the matrix is built in HBM at the engine's row pitch and searched where it lies -- no index
    file in between.  -> cobs_amd.Search"""
    from ._capi import Options
    from .search import Search
    compact = kind in (1, "compact")
    params = index_params or (CompactIndexParameters() if compact else ClassicIndexParameters())
    docs = _as_list(list if list is not None else input, file_type)
    b = _params(params, device)
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.device = device
    h = C.c_void_p()
    check(_capi.load().cobs_gpu_build_index_list(1 if compact else 0, docs._h, C.byref(b), C.byref(o), C.byref(h)))
    return Search(None, _handle=h)


def release_build_buffers():
    """free the staging memory the builders keep between builds (cobs_gpu_build_release_buffers)"""
    _capi.load().cobs_gpu_build_release_buffers()


def classic_combine(in_files, out_file, mem_bytes=0, device=-1):
    """classic_combine (construction/classic_index.cpp:195-327): several classic indexes with the
    same parameters -> one, rows concatenated at bit granularity on the GPU"""
    lib = _capi.load()
    arr = (C.c_char_p * len(in_files))(*[os.fsencode(p) for p in in_files])
    check(lib.cobs_gpu_combine_classic(arr, len(in_files), os.fsencode(out_file), int(mem_bytes), device))


def compact_combine(in_files, out_file, page_size):
    """compact_combine_into_compact (construction/compact_index.cpp:51-169): classic indexes -> the
    sub-indexes of one compact index, rows padded to page_size bytes"""
    lib = _capi.load()
    arr = (C.c_char_p * len(in_files))(*[os.fsencode(p) for p in in_files])
    check(lib.cobs_gpu_combine_compact(arr, len(in_files), os.fsencode(out_file), int(page_size)))


def classic_construct_random(out_file, signature_size=2 * 1024 * 1024, num_documents=10000, document_size=1000000,
                             num_hashes=1, seed=1, device=-1):
    """`cobs classic-construct-random` (src/cobs.cpp:243-291, classic_index.cpp:661-725), defaults
    included: documents of random 31-mers hashed into a classic index on the GPU"""
    check(_capi.load().cobs_gpu_construct_random(os.fsencode(out_file), int(signature_size), int(num_documents),
                                                 int(document_size), int(num_hashes), int(seed), device))


def write_synthetic(out_file, kind, signature_sizes, num_docs, page_size=0, term_size=31, canonicalize=1,
                    num_hashes=1, seed=1, device=-1):
    """The procedural benchmark index (Search.synthetic) as a .cobs_classic / .cobs_compact FILE:
    the generator tool next to `cobs classic-construct-random` (reference src/cobs.cpp:243-291)."""
    from ._capi import Synth
    lib = _capi.load()
    sigs = (C.c_uint64 * len(signature_sizes))(*[int(s) for s in signature_sizes])
    d = Synth()
    d.kind = 1 if kind in (1, "compact") else 0
    d.term_size, d.canonicalize, d.num_pages = term_size, canonicalize, len(signature_sizes)
    d.num_hashes, d.page_size, d.num_docs, d.seed = num_hashes, page_size, num_docs, seed
    d.signature_sizes = C.cast(sigs, C.POINTER(C.c_uint64))
    check(lib.cobs_gpu_write_synthetic(C.byref(d), os.fsencode(out_file), device))


__all__ = ["write_synthetic", "build_search", "release_build_buffers", "classic_combine", "compact_combine", "classic_construct_random", "DocumentList", "DocumentEntry", "FileType", "ClassicIndexParameters", "CompactIndexParameters",
           "classic_construct", "classic_construct_list", "compact_construct", "compact_construct_list",
           "disable_cache"]
