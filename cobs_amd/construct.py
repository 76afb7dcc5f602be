"""Host-side mirror of the reference's Python construction API (python/module.cpp:149-348):
DocumentList, ClassicIndexParameters, CompactIndexParameters, classic_construct,
compact_construct, disable_cache -- same names, arguments and defaults.  Parsing of
FASTA input (plain or .gz) happens here on the host; hashing the terms and setting the
signature bits happens on the GPU (cobs_gpu_build_classic / cobs_gpu_build_compact).

Input scope: FASTA documents (.fa/.fasta/.fna/.ffn/.faa/.frn, optionally .gz) and
in-memory documents; the reference's other parsers (FASTQ, McCortex, multi-FASTA, text,
.cobs_doc) are out of scope (component 8).
"""
import ctypes as C
import gzip
import os

from . import _capi
from ._capi import BuildParams, check

_FASTA_EXT = (".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn")


def disable_cache(disable=True):
    """The reference caches FASTA statistics in .cobs_cache files; this mirror never
    writes caches, so this is a no-op kept for API compatibility (module.cpp:100-105)."""
    return None


def _is_fasta(path):
    p = path[:-3] if path.endswith(".gz") else path
    return p.endswith(_FASTA_EXT)


def _base_name(path):
    """cobs::base_name: file name cut at the first '.' (reference cobs/util/file.hpp:69-76)"""
    return os.path.basename(path).split(".")[0]


def _read_fasta(path):
    """-> (text with sequences joined by newlines, size).  Terms continue across line
    breaks inside a sequence; comment ('>' ';') and empty lines end a sequence
    (reference cobs/fasta_file.hpp:53-91,155-182).  size = FastaFile::size()."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    size = sum(len(ln) + 1 for ln in lines)
    seqs, cur = [], []
    for ln in lines:
        if len(ln) == 0 or ln[:1] in (b">", b";"):
            if cur:
                seqs.append(b"".join(cur))
            cur = []
        else:
            cur.append(ln)
    if cur:
        seqs.append(b"".join(cur))
    return b"\n".join(seqs), size


class DocumentEntry:
    """cobs::DocumentEntry (document_list.hpp:62-76) as far as construction needs it"""

    def __init__(self, path, name, size, text=None):
        self.path = path
        self.name = name
        self.size = size
        self.type = "fasta"
        self._text = text

    def text(self):
        if self._text is None:
            self._text = _read_fasta(self.path)[0]
        return self._text


class DocumentList:
    """cobs::DocumentList: a directory scan (recursive, sorted by path) or an explicit list"""

    def __init__(self, root=None, file_type="any"):
        self._list = []
        if root is not None:
            self.add_recursive(root, file_type)

    def size(self):
        return len(self._list)

    __len__ = size

    def __getitem__(self, i):
        return self._list[i]

    def __iter__(self):
        return iter(self._list)

    def add(self, path):
        text, size = _read_fasta(path)
        self._list.append(DocumentEntry(path, _base_name(path), size, text))

    def add_document(self, name, sequences):
        """in-memory document: a name and its sequences (bytes)"""
        text = b"\n".join(sequences)
        self._list.append(DocumentEntry(name, name, len(text) + 1, text))

    def add_recursive(self, root, file_type="any"):
        if os.path.isfile(root):
            self.add(root)
        else:
            for dirpath, _, files in os.walk(root):
                for fn in files:
                    if _is_fasta(fn):
                        self.add(os.path.join(dirpath, fn))
        self.sort_by_path()

    def sort_by_path(self):
        self._list.sort(key=lambda d: d.path)

    def sort_by_size(self):
        self._list.sort(key=lambda d: (d.size, d.path))


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


def _build(fn, docs, params, out_file, device):
    lib = _capi.load()
    names = (C.c_char_p * len(docs))(*[d.name.encode() for d in docs])
    texts = [d.text() for d in docs]
    tarr = (C.c_char_p * len(docs))(*texts)
    lens = (C.c_size_t * len(docs))(*[len(t) for t in texts])
    b = _params(params, device)
    check(getattr(lib, fn)(names, tarr, lens, len(docs), C.byref(b), os.fsencode(out_file)))


def classic_construct(input=None, out_file=None, index_params=None, file_type="any", tmp_path="",
                      list=None, device=-1):
    """cobs_index.classic_construct (module.cpp:235-270): documents in path order, one
    signature size from the largest document, file written in the reference's format"""
    params = index_params or ClassicIndexParameters()
    docs = _as_list(list if list is not None else input, file_type)
    _check_output(out_file, ".cobs_classic", params)
    ordered = sorted(docs, key=lambda d: d.path)
    _build("cobs_gpu_build_classic", ordered, params, out_file, device)


def compact_construct(input=None, out_file=None, index_params=None, file_type="any", tmp_path="",
                      list=None, device=-1):
    """cobs_index.compact_construct (module.cpp:314-348): documents sorted by (size, path),
    groups of 8 * page_size documents (path order inside a group)"""
    params = index_params or CompactIndexParameters()
    docs = _as_list(list if list is not None else input, file_type)
    _check_output(out_file, ".cobs_compact", params)
    ordered = sorted(docs, key=lambda d: (d.size, d.path))
    page_size = params.page_size
    if page_size == 0:      # compact_index.cpp:184-189
        v = int((len(ordered) // 8) ** 0.5)
        p2 = 1
        while p2 < v:
            p2 *= 2
        page_size = min(max(p2 if v else 0, 8), 4096)
    group = 8 * page_size
    final = []
    for g in range(0, len(ordered), group):
        final.extend(sorted(ordered[g:g + group], key=lambda d: d.path))
    fixed = CompactIndexParameters()
    fixed.__dict__.update(params.__dict__)
    fixed.page_size = page_size
    _build("cobs_gpu_build_compact", final, fixed, out_file, device)


def classic_construct_list(list, out_file, index_params=None, tmp_path="", device=-1):
    """cobs_index.classic_construct_list (module.cpp:253-268): same, from a populated DocumentList"""
    classic_construct(None, out_file, index_params, "any", tmp_path, list=list, device=device)


def compact_construct_list(list, out_file, index_params=None, tmp_path="", device=-1):
    """cobs_index.compact_construct_list (module.cpp:332-347)"""
    compact_construct(None, out_file, index_params, "any", tmp_path, list=list, device=device)


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


__all__ = ["write_synthetic", "DocumentList", "DocumentEntry", "ClassicIndexParameters", "CompactIndexParameters",
           "classic_construct", "classic_construct_list", "compact_construct", "compact_construct_list",
           "disable_cache"]
