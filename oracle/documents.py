"""The reference's document readers and DocumentList restated -- TEST INFRASTRUCTURE ONLY.

Checker of cobs_amd/csrc/documents.cpp (the readers in front of the GPU construction): every
reader below follows its reference counterpart's loop literally -- the same buffers, the same
string operations, the same stale variables -- and yields the BUFFERS whose k-windows the
reference hands to its callback, in order.  Nothing under cobs_amd/ imports this file.

  TextFile::process_terms          cobs/text_file.hpp:44-66
  CortexFile                       cobs/cortex_file.hpp:29-158
  KMer<N>::init / to_string        cobs/kmer.hpp:54-99, cobs/kmer.cpp:148-213
  KMerBufferHeader, KMerBuffer     cobs/file/kmer_buffer_header.cpp:20-37, cobs/kmer_buffer.hpp:49-73
  FastqFile                        cobs/fastq_file.hpp:53-198
  FastaMultifile, FastaSubfile     cobs/fasta_multifile.hpp:38-63, 134-180
  DocumentList, DocumentEntry      cobs/document_list.hpp:62-411
  (FastaFile is in oracle/construct.py)

Pinned by tests/test_oracle_pins.py against the reference's own fixtures and expectations
(tests/cortex_file.cpp, fastq_file.cpp, fasta_multifile.cpp, text_file.cpp and their data files,
copied to tests/golden/documents/).
"""
import gzip
import os
import struct

import numpy as np

from . import construct as _c
from . import oracle as _o

ANY, TEXT, CORTEX, KMER_BUFFER, FASTA, FASTQ, FASTA_MULTI, FASTQ_MULTI, LIST = range(9)
M64 = (1 << 64) - 1


def _read(path, gunzip=False):
    if gunzip and path.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def _getlines(data, pos=0):
    """std::getline until it fails: yields (line, offset after the line, stream still good)"""
    n = len(data)
    while pos < n:
        e = data.find(b"\n", pos)
        if e < 0:
            yield data[pos:], n, False          # the line ran into the end of the stream: eofbit
            return
        yield data[pos:e], e + 1, True
        pos = e + 1


def windows(buffers, k):
    """the terms: every k-window of every buffer, in order"""
    out = []
    for b in buffers:
        out.extend(b[i:i + k] for i in range(len(b) - k + 1))
    return out


# ---------------------------------------------------------------------------
# TextFile


def text_term_buffers(path, k):
    """text_file.hpp:44-66 with its 64 KiB buffer: after each read the last k-1 characters are
    copied to the front FROM OFFSET wb - k + 1 (wb = bytes just read), not from the buffer's end."""
    data = _read(path)
    size = 64 * 1024
    buffer = bytearray(size)
    pos, off, eof = 0, 0, False
    while not eof:
        want = size - pos
        chunk = data[off:off + want]
        wb = len(chunk)
        off += wb
        buffer[pos:pos + wb] = chunk
        if wb < want:
            eof = True                          # istream::read came up short
        if pos + wb >= k:
            yield bytes(buffer[:pos + wb])
        if wb + 1 < k:
            break
        buffer[0:k - 1] = buffer[wb - k + 1:wb]
        pos = k - 1


# ---------------------------------------------------------------------------
# packed k-mers, McCortex, .cobs_doc

_BYTE_TO_BASES = ["".join("ACGT"[(b >> s) & 3] for s in (6, 4, 2, 0)).encode() for b in range(256)]   # kmer.cpp:148-213


def kmer_to_string(packed, kmer_size):
    """KMer<N>::to_string (kmer.hpp:85-99) / cortex_file.hpp:136-147"""
    nbytes = (kmer_size + 3) // 4
    s = b""
    for i in range(nbytes):
        t = _BYTE_TO_BASES[packed[nbytes - 1 - i]]
        if i == 0 and kmer_size % 4 != 0:
            t = t[4 - kmer_size % 4:]
        s += t
    return s


def kmer_pack(chars):
    """KMer<N>::init (kmer.hpp:54-67): the last four characters go to byte 0, ..., the leading
    N % 4 characters, left-padded with 'A', to the last byte"""
    n = len(chars)
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    padded = b"A" * ((4 - n % 4) % 4) + bytes(chars)
    out = bytearray(len(padded) // 4)
    for j in range(len(out)):
        v = 0
        for c in padded[4 * j:4 * j + 4]:
            v = (v << 2) | code[c]
        out[len(out) - 1 - j] = v
    return bytes(out)


def cortex_header(path):
    """-> dict(version, kmer_size, words, colors, name, data_begin, num_kmers) (cortex_file.hpp:55-107)"""
    b = _read(path)
    if b[:6] != b"CORTEX":
        raise ValueError("CortexFile: magic number not found @ " + path)
    p = 6
    version, kmer_size, words, colors = struct.unpack_from("<4I", b, p)
    p += 16
    if version != 6:
        raise ValueError("Invalid .ctx file version")
    if colors != 1:
        raise ValueError("Invalid number of colors")
    p += 12 * colors
    name = b""
    for _ in range(colors):
        n, = struct.unpack_from("<I", b, p)
        name = b[p + 4:p + 4 + n]
        p += 4 + n
    p += 16 * colors
    for _ in range(colors):
        p += 12
        n, = struct.unpack_from("<I", b, p)
        p += 4 + n
    if b[p:p + 6] != b"CORTEX":
        raise ValueError("CortexFile: magic number not found @ " + path)
    p += 6
    return dict(version=version, kmer_size=kmer_size, words=words, colors=colors, name=name.decode("latin-1"),
                data_begin=p, num_kmers=(len(b) - p) // (8 * words + 5 * colors), file_size=len(b))


def cortex_term_buffers(path, k):
    """cortex_file.hpp:118-152: every record's k-mer string is a buffer of its own.  (The packed
    size is a function-local static there, i.e. taken from the first file a process reads; this
    follows the reference's test vectors, which are per file.)"""
    h = cortex_header(path)
    b = _read(path)
    rec = 8 * h["words"] + 5 * h["colors"]
    if k > h["kmer_size"]:
        return
    for r in range(h["num_kmers"]):
        at = h["data_begin"] + r * rec
        yield kmer_to_string(b[at:at + 8 * h["words"]], h["kmer_size"])


def kmer_buffer_header(path):
    b = _read(path)
    if b[:13] != b"COBS:DOCUMENT" or struct.unpack_from("<I", b, 13)[0] != 1:
        raise ValueError("invalid .cobs_doc " + path)
    kmer_size, = struct.unpack_from("<I", b, 17)
    e = b.index(b"\0", 21)
    name = b[21:e]
    if b[e + 1:e + 9] != b"DOCUMENT":
        raise ValueError("invalid .cobs_doc " + path)
    return dict(kmer_size=kmer_size, name=name.decode("latin-1"), data_begin=e + 9, file_size=len(b))


def kmer_buffer_term_buffers(path, k):
    """document_list.hpp:116-129: KMerBuffer<31> only"""
    h = kmer_buffer_header(path)
    assert k == 31 and h["kmer_size"] == 31
    b = _read(path)
    for at in range(h["data_begin"], len(b) - 7, 8):
        yield kmer_to_string(b[at:at + 8], 31)


def write_kmer_buffer(path, name, kmers):
    """KMerBuffer<31>::serialize (kmer_buffer.hpp:49-55): what the reference's tests write with
    generate_documents_* (tests/test_util.hpp:44-84)"""
    with open(path, "wb") as f:
        f.write(b"COBS:DOCUMENT" + struct.pack("<II", 1, 31) + name.encode() + b"\0" + b"DOCUMENT")
        for km in kmers:
            assert len(km) == 31
            f.write(kmer_pack(km))


# ---------------------------------------------------------------------------
# FASTQ


def fastq_index(path):
    """compute_index (fastq_file.hpp:53-86) -> (size, {read length: count})"""
    size, hist = 0, {}
    for num, (line, _, _) in enumerate(_getlines(_read(path, True))):
        size += len(line) + 1
        if num % 4 == 0 and line[:1] != b"@":
            raise ValueError("FastqFile: line %d does not start with @ - %s" % (num, path))
        if num % 4 == 2 and line[:1] != b"+":
            raise ValueError("FastqFile: line %d does not start with + - %s" % (num, path))
        if num % 4 == 1:
            hist[len(line)] = hist.get(len(line), 0) + 1
    return size, hist


def fastq_term_buffers(path, k):
    """process_terms (fastq_file.hpp:163-182): the second line of every four"""
    for num, (line, _, _) in enumerate(_getlines(_read(path, True))):
        if num % 4 == 1 and len(line) >= k:
            yield line


# ---------------------------------------------------------------------------
# multi-FASTA


def mfasta_index(path):
    """compute_index (fasta_multifile.hpp:134-180) -> [(pos_begin, size)]: the do / while
    (is.good()) loop over getline, statement by statement"""
    data = _read(path)
    if data[:1] not in (b">", b";"):
        raise ValueError("FastaMultifile: file does not start with > or ; - " + path)
    lines = _getlines(data)
    out = []
    line, pos, good = next(lines)                # "read first line"
    while True:
        if line[:1] == b">":
            if not good:
                break                            # a header at the very end, no newline: tellg() is -1 there
            pos_begin, size, broke = pos, 0, False
            for line, pos, good in lines:        # while (std::getline(is, line))
                if line[:1] in (b">", b";"):
                    broke = True
                    break
                size += len(line)
            out.append((pos_begin, size))
            if not broke:
                break                            # the inner getline failed: the stream is not good
        else:                                    # ';' comment, empty line, '\r', "invalid line": next line
            nxt = next(lines, None)
            if nxt is None:
                break
            line, pos, good = nxt
        if not good:
            break                                # `while (is.good())`: that line ran into the end of the file
    return out


def mfasta_term_buffers(path, pos_begin, k):
    """FastaSubfile::process_terms (fasta_multifile.hpp:38-63), std::string arithmetic included"""
    data = bytearray()
    for line, _, _ in _getlines(_read(path), pos_begin):
        if line[:1] in (b">", b";"):
            break
        data += line
        if len(data) == 0:
            continue
        if len(data) >= k:
            yield bytes(data)
        n = (len(data) - k + 1) & M64           # data.erase(0, data.size() - term_size + 1) in size_t
        del data[:min(n, len(data))]


# ---------------------------------------------------------------------------
# DocumentList


class Entry:
    def __init__(self, path, ftype, name, size, subdoc_index=0, term_size=0, term_count=0, extra=None):
        self.path, self.type, self.name, self.size = path, ftype, name, size
        self.subdoc_index, self.term_size, self.term_count, self.extra = subdoc_index, term_size, term_count, extra

    def num_terms(self, k):
        """DocumentEntry::num_terms (document_list.hpp:85-112)"""
        if self.type in (TEXT, FASTA_MULTI):
            return 0 if self.size < k else self.size - k + 1
        if self.type in (CORTEX, KMER_BUFFER):
            return self.term_count * (self.term_size - k + 1) if self.term_size >= k else 0
        if self.type == FASTQ:
            return sum(c * (0 if n < k else n - k + 1) for n, c in self.extra.items())
        if self.type == FASTA:
            return sum(max(len(s) - k + 1, 0) for s in _c.fasta_sequences(self.path))
        raise ValueError("DocumentEntry: unknown file type")

    def term_buffers(self, k):
        """DocumentEntry::process_terms (:116-151)"""
        if self.type == TEXT:
            return text_term_buffers(self.path, k)
        if self.type == CORTEX:
            return cortex_term_buffers(self.path, k)
        if self.type == KMER_BUFFER:
            return kmer_buffer_term_buffers(self.path, k)
        if self.type == FASTA:
            return _c.fasta_term_buffers(self.path, k)
        if self.type == FASTQ:
            return fastq_term_buffers(self.path, k)
        if self.type == FASTA_MULTI:
            return mfasta_term_buffers(self.path, self.extra, k)
        raise ValueError("DocumentEntry: unknown file type")

    def terms(self, k):
        return windows(self.term_buffers(k), k)


_EXT = [((".txt",), TEXT), ((".ctx", ".cortex"), CORTEX), ((".cobs_doc",), KMER_BUFFER),
        (tuple(e + z for e in (".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn") for z in ("", ".gz")), FASTA),
        (tuple(e + z for e in (".fq", ".fastq") for z in ("", ".gz")), FASTQ),
        ((".mfasta",), FASTA_MULTI), ((".mfastq",), FASTQ_MULTI), ((".list",), LIST)]


def identify_filetype(path):
    """document_list.hpp:199-243"""
    for exts, t in _EXT:
        if path.endswith(exts):
            return t
    return ANY


def load(path):
    """DocumentList::load (:246-334)"""
    t = identify_filetype(path)
    if t == TEXT:
        return [Entry(path, t, _c.base_name(path), os.path.getsize(path))]
    if t == CORTEX:
        h = cortex_header(path)
        return [Entry(path, t, h["name"], h["file_size"], 0, h["kmer_size"], h["num_kmers"])]
    if t == KMER_BUFFER:
        h = kmer_buffer_header(path)
        return [Entry(path, t, h["name"], h["file_size"], 0, h["kmer_size"],
                      (h["file_size"] - h["data_begin"]) // ((h["kmer_size"] + 3) // 4))]
    if t == FASTA:
        return [Entry(path, t, _c.base_name(path), fasta_size(path))]
    if t == FASTQ:
        size, hist = fastq_index(path)
        return [Entry(path, t, _c.base_name(path), size, extra=hist)]
    if t == FASTA_MULTI:
        return [Entry(path, t, "%s_%06u" % (_c.base_name(path), i), size, i, extra=pos)
                for i, (pos, size) in enumerate(mfasta_index(path))]
    raise ValueError("DocumentList: unknown document file to add: " + path)


def fasta_size(path):
    """FastaFile::compute_index (fasta_file.hpp:53-90): nothing is counted when the first getline
    already hits the end of the stream"""
    data = _read(path, True)
    if b"\n" not in data:
        return 0
    return sum(len(line) + 1 for line, _, _ in _getlines(data))


def document_list(root, filter=ANY):
    """DocumentList(root, filter) (:160-163, 345-411): recursive scan / .list file / one file,
    entries sorted by (path, subdoc_index)"""
    paths = []
    if os.path.isdir(root):
        for d, _, files in os.walk(root):
            for fn in files:
                p = os.path.join(d, fn)
                t = identify_filetype(p)
                ok = t in (TEXT, CORTEX, KMER_BUFFER, FASTA, FASTQ, FASTA_MULTI, FASTQ_MULTI) if filter == ANY else t == filter
                if ok:
                    paths.append(p)
    elif root.endswith(".list") or filter == LIST:
        for line, _, _ in _getlines(_read(root)):
            if len(line) == 0 or line[:1] == b"#":
                continue
            p = line.decode()
            paths.append(p if os.path.isabs(p) else os.path.join(os.path.dirname(root), p))
    elif os.path.isfile(root):
        paths.append(root)
    out = []
    for p in sorted(paths):
        try:
            out.extend(load(p))
        except ValueError:
            pass                                 # "EXCEPTION: ..." is logged, the scan goes on (:393-402)
    out.sort(key=lambda e: (e.path, e.subdoc_index))
    return out


def docs(entries, k=31, canonicalize=1, num_hashes=1):
    """entries -> construct.Doc objects (hashes of every term) for classic_/compact_construct"""
    out = []
    for e in entries:
        hs = [_o.term_hashes(b, k, canonicalize, num_hashes)[0] for b in e.term_buffers(k)]
        hashes = np.concatenate(hs) if hs else np.zeros((0, num_hashes), dtype=np.uint64)
        d = _c.Doc(e.name, e.path, e.size, e.num_terms(k), hashes)
        d.subdoc = e.subdoc_index
        out.append(d)
    return out
