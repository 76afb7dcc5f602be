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


def _fasta_runs(lines, k):
    """The character runs whose k-grams FastaFile::process_terms hashes (reference
    cobs/fasta_file.hpp:155-182), including what its line buffer does at the edges: a run ends at
    a comment ('>' ';') or empty line -- but the test looks at index `pos` of the buffer, which is
    0 while the run so far is shorter than k (then a comment line is swallowed INTO the run) and
    keeps its old value k-1 after a clear (then a line of exactly k-1 characters counts as empty).
    Reads past the end of the buffer (undefined in the reference) are taken as sequence."""
    runs, cur = [], bytearray()
    held, pos = 0, 0             # held = characters of the run the reference still has in its buffer
    for ln in lines:
        size = held + len(ln)
        if size == pos:
            comment = True
        elif pos < size:
            ch = cur[len(cur) - held + pos] if pos < held else ln[pos - held]
            comment = ch in b">;"
        else:
            comment = False
        if comment:
            if cur:
                runs.append(bytes(cur))
            cur, held = bytearray(), 0
            continue
        cur += ln
        if size > k - 1:
            held, pos = k - 1, k - 1
        else:
            held, pos = size, 0
    if cur:
        runs.append(bytes(cur))
    return runs


def _fasta_lines(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return lines


def _read_fasta(path, k=31):
    """-> (runs joined by newlines, size, number of terms).  The bits of a document come from
    process_terms (the runs above); size = FastaFile::size() and the term count that sizes a
    signature come from the file's index (compute_index / num_terms, fasta_file.hpp:53-91,
    147-153): maximal runs of non-comment, non-empty lines."""
    lines = _fasta_lines(path)
    size = sum(len(ln) + 1 for ln in lines)
    num_terms, run = 0, 0
    for ln in lines[1:] + [b""]:
        if len(ln) == 0 or ln[:1] in (b">", b";"):
            num_terms += max(run - k + 1, 0)
            run = 0
        else:
            run += len(ln)
    return b"\n".join(_fasta_runs(lines, k)), size, num_terms


class DocumentEntry:
    """cobs::DocumentEntry (document_list.hpp:62-76) as far as construction needs it"""

    def __init__(self, path, name, size, text=None):
        self.path = path
        self.name = name
        self.size = size
        self.type = "fasta"
        self._text = text            # in-memory documents: sequences joined by newlines
        self._by_k = {}

    def text(self, k=31):
        """the character runs whose k-grams are hashed, joined by newlines"""
        if self._text is not None:
            return self._text
        if k not in self._by_k:
            t, _, n = _read_fasta(self.path, k)
            self._by_k[k] = (t, n)
        return self._by_k[k][0]

    def num_terms(self, k=31):
        """FastaFile::num_terms(k): what sizes a signature (fasta_file.hpp:147-153)"""
        if self._text is not None:
            return sum(max(len(sq) - k + 1, 0) for sq in self._text.split(b"\n"))
        self.text(k)
        return self._by_k[k][1]


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
        size = sum(len(ln) + 1 for ln in _fasta_lines(path))
        self._list.append(DocumentEntry(path, _base_name(path), size))

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
    b.text_batch_bytes = int(getattr(p, "text_batch_bytes", 0))
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
    k = params.term_size
    texts = [d.text(k) for d in docs]
    tarr = (C.c_char_p * len(docs))(*texts)
    lens = (C.c_size_t * len(docs))(*[len(t) for t in texts])
    b = _params(params, device)
    # the term counts that size the signatures (the reference takes them from the documents'
    # indexes, not from what process_terms later hashes)
    terms = (C.c_uint64 * len(docs))(*[d.num_terms(k) for d in docs])
    b.doc_terms = C.cast(terms, C.POINTER(C.c_uint64))
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


def build_search(input=None, index_params=None, kind="classic", file_type="any", list=None, device=-1):
    """classic_construct / compact_construct straight into a query handle (cobs_gpu_build_index):
    the matrix is built in HBM at the engine's row pitch and searched where it lies -- no index
    file in between.  -> cobs_amd.Search"""
    from ._capi import Options
    from .search import Search
    compact = kind in (1, "compact")
    params = index_params or (CompactIndexParameters() if compact else ClassicIndexParameters())
    docs = _as_list(list if list is not None else input, file_type)
    if compact:
        ordered = sorted(docs, key=lambda d: (d.size, d.path))
        page_size = getattr(params, "page_size", 0)
        if page_size == 0:
            v = int((len(ordered) // 8) ** 0.5)
            p2 = 1
            while p2 < v:
                p2 *= 2
            page_size = min(max(p2 if v else 0, 8), 4096)
        final = []
        for g in range(0, len(ordered), 8 * page_size):
            final.extend(sorted(ordered[g:g + 8 * page_size], key=lambda d: d.path))
        fixed = CompactIndexParameters()
        fixed.__dict__.update(params.__dict__)
        fixed.page_size = page_size
        docs, params = final, fixed
    else:
        docs = sorted(docs, key=lambda d: d.path)
    lib = _capi.load()
    k = params.term_size
    names = (C.c_char_p * len(docs))(*[d.name.encode() for d in docs])
    texts = [d.text(k) for d in docs]
    tarr = (C.c_char_p * len(docs))(*texts)
    lens = (C.c_size_t * len(docs))(*[len(t) for t in texts])
    b = _params(params, device)
    terms = (C.c_uint64 * len(docs))(*[d.num_terms(k) for d in docs])
    b.doc_terms = C.cast(terms, C.POINTER(C.c_uint64))
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.device = device
    h = C.c_void_p()
    check(lib.cobs_gpu_build_index(1 if compact else 0, names, tarr, lens, len(docs), C.byref(b), C.byref(o), C.byref(h)))
    return Search(None, _handle=h)


def classic_combine(in_files, out_file, mem_bytes=0, device=-1):
    """classic_combine (construction/classic_index.cpp:195-327): several classic indexes with the
    same parameters -> one, rows concatenated at bit granularity on the GPU"""
    lib = _capi.load()
    arr = (C.c_char_p * len(in_files))(*[os.fsencode(p) for p in in_files])
    check(lib.cobs_gpu_combine_classic(arr, len(in_files), os.fsencode(out_file), int(mem_bytes), device))


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


__all__ = ["write_synthetic", "build_search", "classic_combine", "classic_construct_random", "DocumentList", "DocumentEntry", "ClassicIndexParameters", "CompactIndexParameters",
           "classic_construct", "classic_construct_list", "compact_construct", "compact_construct_list",
           "disable_cache"]
