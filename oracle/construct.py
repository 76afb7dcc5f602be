"""Index construction restated for small inputs -- TEST INFRASTRUCTURE ONLY.

The query path under test reads .cobs_classic / .cobs_compact files.  To pin the
oracle against the reference's own tests (which construct an index and then
assert on query scores) this module restates, with numpy, just enough of the
reference's construction side:

  * FASTA term extraction           cobs/fasta_file.hpp:53-91,155-182
  * signature sizing                cobs/util/calc_signature_size.cpp:17-33
  * bit layout                      cobs/construction/classic_index.cpp:40-73
  * classic / compact geometry      cobs/construction/classic_index.cpp:565-575,
                                    cobs/construction/compact_index.cpp:171-340,51-169
  * header serialisation            cobs/file/classic_index_header.cpp:26-37,
                                    cobs/file/compact_index_header.cpp:20-43
  * test corpus generators          tests/test_util.hpp:44-84

  * classic_combine                 cobs/construction/classic_index.cpp:195-327
  * classic_construct_random        cobs/construction/classic_index.cpp:661-725

It is also the checker of the product's own GPU construction (SURVEY.md section 8f rank 4:
cobs_amd/csrc/build.cpp must write these bytes); nothing under cobs_amd/ imports this file.
"""
import gzip
import math
import os
import struct

import numpy as np

from . import oracle as _o


class Doc:
    """One input document: a name, the sort key 'path', its 'size' (sort key of
    compact construction) and the full 64-bit hashes [n_terms, H] of its terms."""

    def __init__(self, name, path, size, num_terms, hashes):
        self.name = name
        self.path = path
        self.size = size
        self.num_terms = num_terms
        self.hashes = hashes


# ---------------------------------------------------------------------------
# FASTA documents


def _fasta_lines(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()          # std::getline does not yield a final empty line
    return lines


def fasta_sequences(path):
    """Maximal runs of sequence lines (fasta_file.hpp:155-182: comment or empty
    lines restart the term buffer; terms continue across line breaks)."""
    seqs, cur = [], []
    for ln in _fasta_lines(path):
        if len(ln) == 0 or ln[:1] in (b">", b";"):
            if cur:
                seqs.append(b"".join(cur))
            cur = []
        else:
            cur.append(ln)
    if cur:
        seqs.append(b"".join(cur))
    return seqs


def fasta_size(path):
    """FastaFile::size(): sum of line lengths + 1 (fasta_file.hpp:53-91)"""
    return sum(len(ln) + 1 for ln in _fasta_lines(path))


def base_name(path):
    """cobs::base_name: file name cut at the first '.' (util/file.hpp:69-76)"""
    return os.path.basename(path).split(".")[0]


def fasta_term_buffers(path, k):
    """FastaFile::process_terms (fasta_file.hpp:155-182) followed literally, std::string and all:
    yields every buffer whose k-windows the reference hands to the callback.  Its quirks are part
    of what an index holds: after a run no longer than k-1 characters the buffer is kept and `pos`
    is reset to 0, so a following '>' / ';' line is APPENDED and hashed as sequence; `pos` survives
    a clear, so after a comment line the next line is tested at index k-1 instead of 0 (a line of
    exactly k-1 characters is dropped as "empty").  Where the reference indexes past the end of
    the string (undefined behaviour) the line is taken as sequence."""
    line = bytearray()
    pos = 0
    for ln in _fasta_lines(path):
        line += ln                                       # tlx::appendline
        if len(line) == pos:
            first = None                                 # "empty line"
        elif pos < len(line):
            first = line[pos:pos + 1]
        else:
            first = b"A"                                 # out-of-range read in the reference
        if first is None or first in (b">", b";"):
            line = bytearray()                           # line.clear(); pos keeps its value
            continue
        if len(line) >= k:
            yield bytes(line)
        if len(line) > k - 1:
            line = line[len(line) - (k - 1):]
            pos = len(line)
        else:
            pos = 0


def fasta_doc(path, k, canonicalize, num_hashes, rel_path=None):
    # bits come from process_terms (the buffers above), the term COUNT that sizes the signature
    # from the cached index (compute_index / num_terms, fasta_file.hpp:53-91,147-153)
    hs = [_o.term_hashes(b, k, canonicalize, num_hashes)[0] for b in fasta_term_buffers(path, k)]
    hashes = np.concatenate(hs) if hs else np.zeros((0, num_hashes), dtype=np.uint64)
    num_terms = sum(max(len(s) - k + 1, 0) for s in fasta_sequences(path))
    return Doc(base_name(path), rel_path or path, fasta_size(path), num_terms, hashes)


def fasta_dir_docs(directory, k=31, canonicalize=1, num_hashes=1):
    """DocumentList(dir): recursive scan, sorted by path (document_list.hpp:169-172)"""
    paths = []
    for root, _, files in os.walk(directory):
        for fn in files:
            if fn.endswith((".fasta", ".fasta.gz", ".fa", ".fa.gz")):
                paths.append(os.path.join(root, fn))
    paths.sort()
    return [fasta_doc(p, k, canonicalize, num_hashes) for p in paths]


# ---------------------------------------------------------------------------
# the reference tests' synthetic corpora (tests/test_util.hpp)


def _kmerbuffer_doc(index, term_idx, all_hashes, prefix=""):
    name = "%sdocument_%06u" % (prefix, index)
    hashes = all_hashes[term_idx]
    # size_ = file size of the .cobs_doc: constant header + 8 bytes per 31-mer
    return Doc(name, name + ".cobs_doc", 64 + len(name) + 8 * len(term_idx), len(term_idx), hashes)


def generate_documents_all(query, num_documents=33, num_terms=1000000, num_hashes=3, prefix=""):
    """tests/test_util.hpp:44-62: document j holds term i iff j % (i % (n-1) + 1) == 0;
    terms i < min(num_terms, len(query) - 31)."""
    k = 31
    all_hashes, good = _o.term_hashes(query, k, 1, num_hashes)
    assert good.all()
    n = min(num_terms, len(query) - 31)
    members = [[] for _ in range(num_documents)]
    for i in range(n):
        m = i % (num_documents - 1) + 1
        for j in range(0, num_documents, m):
            members[j].append(i)
    return [_kmerbuffer_doc(j, np.asarray(members[j], dtype=np.int64), all_hashes, prefix)
            for j in range(num_documents)]


def generate_documents_one(query, num_documents=33, num_hashes=3, prefix=""):
    """tests/test_util.hpp:68-84: document i holds the first term (10 i + 1) times"""
    k = 31
    all_hashes, good = _o.term_hashes(query[:k], k, 1, num_hashes)
    assert good.all()
    return [_kmerbuffer_doc(i, np.zeros(10 * i + 1, dtype=np.int64), all_hashes, prefix)
            for i in range(num_documents)]


# ---------------------------------------------------------------------------
# sizing, bit matrix, file writers


def calc_signature_size(num_elements, num_hashes, false_positive_rate):
    """util/calc_signature_size.cpp:17-33 (all arithmetic in double)"""
    ratio = -float(num_hashes) / math.log(1.0 - math.pow(false_positive_rate, 1.0 / float(num_hashes)))
    assert ratio > 0
    return int(math.ceil(float(num_elements) * ratio))


def build_matrix(docs, signature_size, row_size):
    """classic_index.cpp:40-43: bit d%8 of byte d/8 of row (hash % S)"""
    m = np.zeros((signature_size, row_size), dtype=np.uint8)
    for d, doc in enumerate(docs):
        if doc.hashes.size == 0:
            continue
        rows = (doc.hashes.reshape(-1) % np.uint64(signature_size)).astype(np.int64)
        rows = np.unique(rows)
        m[rows, d // 8] |= np.uint8(1 << (d % 8))
    return m


def classic_header(term_size, canonicalize, names, signature_size, num_hashes):
    b = b"COBS:" + b"CLASSIC_INDEX" + struct.pack("<I", 1)
    b += struct.pack("<IBIQQ", term_size, canonicalize, len(names), signature_size, num_hashes)
    for n in names:
        b += n.encode() + b"\n"
    return b + b"CLASSIC_INDEX"


def compact_header(term_size, canonicalize, page_size, params, names):
    b = b"COBS:" + b"COMPACT_INDEX" + struct.pack("<I", 1)
    b += struct.pack("<IBIIQ", term_size, canonicalize, len(params), len(names), page_size)
    for s, h in params:
        b += struct.pack("<QQ", s, h)
    for n in names:
        b += n.encode() + b"\n"
    pad = (page_size - ((len(b) + len(b"COMPACT_INDEX")) % page_size)) % page_size
    return b + b"\0" * pad + b"COMPACT_INDEX"


def write_classic(path, term_size, canonicalize, names, signature_size, num_hashes, matrix):
    assert matrix.shape == (signature_size, (len(names) + 7) // 8)
    with open(path, "wb") as f:
        f.write(classic_header(term_size, canonicalize, names, signature_size, num_hashes))
        f.write(np.ascontiguousarray(matrix).tobytes())


def write_compact(path, term_size, canonicalize, page_size, params, names, matrices):
    """matrices[p]: uint8 [S_p, page_size] (last group already zero-padded)"""
    with open(path, "wb") as f:
        f.write(compact_header(term_size, canonicalize, page_size, params, names))
        for (s, _), m in zip(params, matrices):
            assert m.shape == (s, page_size)
            f.write(np.ascontiguousarray(m).tobytes())


def classic_construct(docs, out_path, term_size=31, canonicalize=1, num_hashes=1,
                      false_positive_rate=0.3, signature_size=0):
    """classic_construct (classic_index.cpp:565-659): documents in DocumentList order (path,
    sub-document), one matrix of ceil(D/8)-byte rows; S from the num_terms of the LARGEST document
    by (size, path) -- get_max_file_size, :521-563, std::max_element: the first of equals --
    unless given."""
    docs = sorted(docs, key=lambda d: (d.path, getattr(d, "subdoc", 0)))
    if signature_size == 0:
        big = docs[0]
        for d in docs[1:]:
            if (big.size, big.path) < (d.size, d.path):
                big = d
        signature_size = calc_signature_size(big.num_terms, num_hashes, false_positive_rate)
    row_size = (len(docs) + 7) // 8
    m = build_matrix(docs, signature_size, row_size)
    write_classic(out_path, term_size, canonicalize, [d.name for d in docs], signature_size,
                  num_hashes, m)
    return signature_size


def compact_construct(docs, out_path, term_size=31, canonicalize=1, num_hashes=1,
                      false_positive_rate=0.3, page_size=0):
    """compact_construct (compact_index.cpp:171-340) + compact_combine_into_compact
    (:51-169): sort by (size, path), groups of 8*page_size documents (path order
    inside a group), own signature size per group, rows padded to page_size."""
    docs = sorted(docs, key=lambda d: (d.size, d.path))
    if page_size == 0:
        v = int(math.sqrt(len(docs) // 8))
        p = 1
        while p < v:
            p *= 2
        page_size = min(max(p, 8), 4096)
    group = 8 * page_size
    params, mats, names = [], [], []
    for g in range(0, len(docs), group):
        part = sorted(docs[g:g + group], key=lambda d: (d.path, getattr(d, "subdoc", 0)))   # DocumentList(files), :315
        max_terms = max(d.num_terms for d in part)
        s = calc_signature_size(max_terms, num_hashes, false_positive_rate)
        if max_terms == 0:
            continue
        m = build_matrix(part, s, page_size)
        params.append((s, num_hashes))
        mats.append(m)
        names.extend(d.name for d in part)
    write_compact(out_path, term_size, canonicalize, page_size, params, names, mats)
    return page_size, params


# ---------------------------------------------------------------------------
# classic_combine and classic_construct_random (restated; checkers of the GPU versions)


def read_classic(path):
    """-> (term_size, canonicalize, names, signature_size, num_hashes, matrix uint8 [S, ceil(D/8)])"""
    raw = open(path, "rb").read()
    assert raw[:18] == b"COBS:CLASSIC_INDEX"
    ver, k, canon, ndocs, sig, nh = struct.unpack_from("<IIBIQQ", raw, 18)
    pos = 18 + struct.calcsize("<IIBIQQ")
    names = []
    for _ in range(ndocs):
        e = raw.index(b"\n", pos)
        names.append(raw[pos:e].decode())
        pos = e + 1
    assert raw[pos:pos + 13] == b"CLASSIC_INDEX"
    pos += 13
    row = (ndocs + 7) // 8
    m = np.frombuffer(raw, dtype=np.uint8, count=sig * row, offset=pos).reshape(sig, row)
    return k, canon, names, sig, nh, m


def classic_combine(in_paths, out_path):
    """classic_combine_streams (classic_index.cpp:195-327): output row = the inputs' rows
    concatenated at bit granularity (input i contributes its number of documents), names in order"""
    parts, names, head = [], [], None
    for p in in_paths:
        k, canon, nm, sig, nh, m = read_classic(p)
        assert head in (None, (k, canon, sig, nh))
        head = (k, canon, sig, nh)
        parts.append(np.unpackbits(m, axis=1, bitorder="little")[:, :len(nm)])
        names += nm
    bits = np.concatenate(parts, axis=1)
    pad = (-bits.shape[1]) % 8
    if pad:
        bits = np.concatenate([bits, np.zeros((bits.shape[0], pad), dtype=np.uint8)], axis=1)
    k, canon, sig, nh = head
    write_classic(out_path, k, canon, names, sig, nh, np.packbits(bits, axis=1, bitorder="little"))


def _mix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return z ^ (z >> np.uint64(31))


def classic_construct_random(out_path, signature_size, num_documents, document_size, num_hashes, seed):
    """classic_construct_random (classic_index.cpp:661-725) with the engine's counter generator:
    k-mer j of document d = the low 62 bits of mix64(mix64(seed ^ d) + j), two bits per base
    (A C G T), first base lowest; canonicalised, hashed, bit set"""
    old = np.seterr(over="ignore")
    try:
        row = (num_documents + 7) // 8
        m = np.zeros((signature_size, row), dtype=np.uint8)
        base = np.frombuffer(b"ACGT", dtype=np.uint8)
        for d in range(num_documents):
            key = _mix64(np.array([np.uint64(seed) ^ np.uint64(d)], dtype=np.uint64))[0]
            bits = _mix64((key + np.arange(document_size, dtype=np.uint64)).astype(np.uint64))
            codes = (bits[:, None] >> (np.uint64(2) * np.arange(31, dtype=np.uint64))[None, :]) & np.uint64(3)
            kmers = base[codes.astype(np.int64)]
            for j in range(document_size):
                hs, good = _o.term_hashes(kmers[j].tobytes(), 31, 1, num_hashes)
                rows = (hs.reshape(-1) % np.uint64(signature_size)).astype(np.int64)
                m[rows, d // 8] |= np.uint8(1 << (d % 8))
    finally:
        np.seterr(**old)
    write_classic(out_path, 31, 1, ["file_%06d" % i for i in range(num_documents)], signature_size, num_hashes, m)
