"""Pins of the oracle (oracle/cobs_oracle.c, the CPU restatement) against every
known answer the reference's own tests hold for this path, plus the
facts recorded from a run of the real reference during the survey.  CPU only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import cases

Q50 = b"AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"


# --- XXH64: third-party xxHash (extlib/xxhash, un-vendored); call sites
# classic_search.cpp:84,99, util/misc.hpp:69 --------------------------------------

def test_xxh64_known_answers(oracle):
    assert oracle.xxh64(b"", 0) == 0xEF46DB3751D8E999            # published XXH64 test vector
    kmer = b"AGGAAAGTCTTTTACGCTGGGGTAAGAGTGA"
    assert oracle.xxh64(kmer, 0) == 0xC8D2277D16E89C15
    assert oracle.xxh64(kmer, 1) == 0x5121D4E40ED1BE5B
    assert oracle.xxh64(kmer, 2) == 0x5DCEECCDAD06CA97


def test_xxh64_against_python_xxhash(oracle):
    xxhash = pytest.importorskip("xxhash")       # python-xxhash 3.8.1 / libxxhash 0.8.2 in this image
    rng = np.random.default_rng(0)
    for n in list(range(0, 100)) + [127, 128, 129, 1000]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 2, 7, 2 ** 63 + 5):
            assert oracle.xxh64(data, seed) == xxhash.xxh64_intdigest(data, seed=seed)


# --- canonicalize_kmer: reference tests/util.cpp:38-60 (all seven vectors) ------------

KATS = [
    (b"AGGAAAGTCTTTTACGCTGGGGTAAGAGTGA", b"AGGAAAGTCTTTTACGCTGGGGTAAGAGTGA", True),
    (b"TGGAAAGTCTTTTACGCTGGGGTAAGAGTGA", b"TCACTCTTACCCCAGCGTAAAAGACTTTCCA", True),
    (b"TTTTTTGTCTTTTACGCTGGGGTTTAAAAAA", b"TTTTTTAAACCCCAGCGTAAAAGACAAAAAA", True),
    (b"AAAAAAAAAAAAAAAATTTTTTTTTTTTTTT", b"AAAAAAAAAAAAAAAATTTTTTTTTTTTTTT", True),
    (b"AGGAAAGTCTTTTACGCTGGGXXXAGAGTGA", b"AGGAAAGTCTTTTACGCTGGG\0\0\0AGAGTGA", False),
    (b"TGGAAAGTCTTTTACGCTGGGXXXAGAGTGA", b"TCACTCT\0\0\0CCCAGCGTAAAAGACTTTCCA", False),
    (b"AAAAAAAAAAAAAAAXTTTTTTTTTTTTTTT", b"AAAAAAAAAAAAAAA\0TTTTTTTTTTTTTTT", False),
]


def test_canonicalize_reference_kats(oracle):
    for inp, want, good in KATS:
        got, g = oracle.canonicalize_kmer(inp)
        assert got == want and g == good


def test_canonicalize_tie_rule(oracle):
    """the middle base of an odd k is never compared"""
    for kmer in (b"AAAAAAAAAAAAAAATTTTTTTTTTTTTTTT", b"AAAAAAAAAAAAAAAGTTTTTTTTTTTTTTT", b"CTG", b"AGT"):
        assert oracle.canonicalize_kmer(kmer) == (kmer, True)


def test_canonicalize_is_min_of_kmer_and_revcomp(oracle):
    """reference tests/parameters.cpp:107-122 on random 31-mers"""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    seq = oracle.random_sequence(10000, 1)
    for i in range(len(seq) - 31):
        k = seq[i:i + 31]
        rc = k.translate(comp)[::-1]
        assert oracle.canonicalize_kmer(k)[0] == min(k, rc)


# --- file formats: reference tests/file.cpp:37-142 ---------------------------------

def test_golden_files_are_reproducible(golden_dir, tmp_path):
    """tests/golden/make_golden.py regenerates the committed fixtures byte for byte"""
    import shutil
    work = tmp_path / "golden"
    shutil.copytree(golden_dir, work)
    for n in ("c1.cobs_classic", "c1.cobs_compact", "expected.json"):
        os.remove(work / n)
    root = os.path.dirname(os.path.dirname(golden_dir))
    env = dict(os.environ, PYTHONPATH=root)
    subprocess.check_call([sys.executable, str(work / "make_golden.py")], env=env, cwd=root)
    for n in ("c1.cobs_classic", "c1.cobs_compact", "expected.json"):
        assert (work / n).read_bytes() == open(os.path.join(golden_dir, n), "rb").read(), n


def test_header_geometry(oracle, golden_dir):
    c = oracle.Index.open(os.path.join(golden_dir, "c1.cobs_classic"))
    assert (c.term_size, c.canonicalize, c.num_hashes, c.num_docs) == (31, 1, 1, 7)
    assert c.signature_size(0) == 8748 and c.data_offset == 116          # survey-probed reference build
    assert c.page_size == 1 and c.row_size == 1 and c.counts_size == 8    # classic_index/search_file.hpp:26
    assert os.path.getsize(os.path.join(golden_dir, "c1.cobs_classic")) == 116 + 8748
    k = oracle.Index.open(os.path.join(golden_dir, "c1.cobs_compact"))
    assert (k.page_size, k.num_pages, k.data_offset, k.counts_size) == (8, 1, 128, 64)
    assert k.data_offset % k.page_size == 0                               # tests/file.cpp:122-142
    assert [k.doc_name(i) for i in range(7)] == ["sample%d" % i for i in range(1, 8)]


def test_compact_data_is_page_aligned(oracle, tmp_path):
    for ps, ndocs in ((2, 33), (16, 200), (24, 500), (4096, 40000)):
        pages = (ndocs + 8 * ps - 1) // (8 * ps)
        p = cases.make_compact(cases.tmp(tmp_path, "a%d.cobs_compact" % ps), ndocs, ps, [50 + i for i in range(pages)])
        ix = oracle.Index.open(p)
        assert ix.data_offset % ps == 0 and ix.num_pages == pages
        assert os.path.getsize(p) == ix.data_offset + sum((50 + i) * ps for i in range(pages))


def test_not_an_index(oracle, golden_dir):
    with pytest.raises(oracle.OracleError):
        oracle.Index.open(os.path.join(golden_dir, "fasta", "sample1.fasta"))


# --- exact scores: python/tests/test_cobs_index.py:22-61 + survey-probed outputs -----

def test_python_test_known_answer(oracle, golden_dir):
    exp = json.load(open(os.path.join(golden_dir, "expected.json")))
    for key, name in (("classic", "c1.cobs_classic"), ("compact", "c1.cobs_compact")):
        ix = oracle.Index.open(os.path.join(golden_dir, name))
        r = oracle.search(ix, Q50)
        assert len(r) == 7 and r[0][2] == "sample1" and r[0][3] == 20      # the reference's assertions
        # full vector recorded from the real reference
        assert [(n, s) for (_, _, n, s) in r] == [("sample1", 20), ("sample7", 3), ("sample2", 1),
                                                   ("sample4", 1), ("sample6", 1), ("sample3", 0), ("sample5", 0)]
        assert list(ix.counts(Q50)[:8]) == [20, 1, 0, 1, 0, 1, 3, 0]
        assert [int(x) for x in ix.counts(Q50)] == exp[key]["counts"]
        # threshold = ceil(threshold * T) in double (SURVEY App. D, T = 20)
        assert len(oracle.search(ix, Q50, 0.05)) == 5
        assert len(oracle.search(ix, Q50, 0.051)) == 2
        assert [(n, s) for (_, _, n, s) in oracle.search(ix, Q50, 0.15)] == [("sample1", 20), ("sample7", 3)]
        assert len(oracle.search(ix, Q50, 0.1500001)) == 1
        # max_counts <= 1: unsorted, document order (App. D)
        assert [(n, s) for (_, _, n, s) in oracle.search(ix, Q50[5:36])] == [
            ("sample1", 1), ("sample2", 0), ("sample3", 0), ("sample4", 0), ("sample5", 0),
            ("sample6", 0), ("sample7", 1)]
        # num_results truncation keeps the best
        assert [(n, s) for (_, _, n, s) in oracle.search(ix, Q50, 0.0, 2)] == [("sample1", 20), ("sample7", 3)]
    a = oracle.Index.open(os.path.join(golden_dir, "c1.cobs_compact"))
    b = oracle.Index.open(os.path.join(golden_dir, "c1.cobs_classic"))
    assert [(n, s) for (_, _, n, s) in oracle.search([a, b], Q50)] == [
        ("sample1", 20), ("sample1", 20), ("sample7", 3), ("sample7", 3), ("sample2", 1), ("sample4", 1),
        ("sample6", 1), ("sample2", 1), ("sample4", 1), ("sample6", 1), ("sample3", 0), ("sample5", 0),
        ("sample3", 0), ("sample5", 0)]


def test_every_document_kmer_is_found(oracle, construct, golden_dir, tmp_path):
    """reference tests/fasta_file.cpp:55-94: index with canonicalize=0, 3 hashes, fpr 0.1;
    every 31-mer of a document (protein letters and N included) scores >= 1 in it"""
    fasta = os.path.join(golden_dir, "fasta")
    docs = construct.fasta_dir_docs(fasta, 31, 0, 3)
    assert len(docs) == 7
    p = cases.tmp(tmp_path, "raw.cobs_classic")
    construct.classic_construct(docs, p, canonicalize=0, num_hashes=3, false_positive_rate=0.1)
    ix = oracle.Index.open(p)
    assert ix.canonicalize == 0 and ix.num_hashes == 3
    for d, fn in enumerate(sorted(os.listdir(fasta))):
        for seq in construct.fasta_sequences(os.path.join(fasta, fn)):
            if len(seq) >= 31:
                assert ix.counts(seq)[d] == len(seq) - 30
                res = oracle.search(ix, seq[:31])
                assert len(res) == 7


# --- the reference's synthetic-corpus tests, run through the restatement -----------

def test_classic_index_query_suite(oracle, construct, tmp_path):
    """tests/classic_index_query.cpp:36-154 (query shortened from 50000 to 8000 characters)"""
    query = oracle.random_sequence(8000, 2)
    docs = construct.generate_documents_all(query)
    p = cases.tmp(tmp_path, "all.cobs_classic")
    construct.classic_construct(docs, p, num_hashes=3, false_positive_rate=0.1)
    ix = oracle.Index.open(p)
    res = oracle.search(ix, query)
    assert len(res) == len(docs)
    for (_, d, name, score) in res:
        assert score >= docs[int(name[-2:])].num_terms         # all_included: lower bound
    # false_positive: random 31-mers score 0 or 1, per-document total bounded (scaled from 10000 queries)
    totals = np.zeros(ix.counts_size, dtype=np.int64)
    n = 2000
    for i in range(n):
        c = ix.counts(oracle.random_sequence(31, i))
        assert c.max() <= 1
        totals += c
    assert totals.max() <= 1070 * n / 10000 * 1.25
    for ndocs in (33, 2000):
        one = construct.generate_documents_one(query, ndocs)
        po = cases.tmp(tmp_path, "one%d.cobs_classic" % ndocs)
        construct.classic_construct(one, po, num_hashes=3, false_positive_rate=0.1)
        res = oracle.search(oracle.Index.open(po), query)
        assert len(res) == ndocs and all(r[3] == 1 for r in res)


def test_compact_index_query_suite_all_score_widths(oracle, construct, tmp_path):
    """tests/compact_index_query.cpp:36-181: u8 path (160 chars) and u16 path agree with truth"""
    for length in (160, 6000):
        query = oracle.random_sequence(length, 1)
        docs = construct.generate_documents_all(query)
        p = cases.tmp(tmp_path, "all%d.cobs_compact" % length)
        construct.compact_construct(docs, p, num_hashes=3, false_positive_rate=0.1, page_size=2)
        ix = oracle.Index.open(p)
        assert ix.page_size == 2 and ix.num_pages == 3
        c, width = ix.counts(query, want_width=True)
        assert width == (1 if length == 160 else 2)
        res = oracle.search(ix, query)
        assert len(res) == len(docs)
        for (_, d, name, score) in res:
            assert score >= docs[int(name[-2:])].num_terms
    one = construct.generate_documents_one(query, 2000)
    po = cases.tmp(tmp_path, "one.cobs_compact")
    construct.compact_construct(one, po, num_hashes=3, false_positive_rate=0.1, page_size=2)
    ix = oracle.Index.open(po)
    assert ix.num_pages == 125
    res = oracle.search(ix, query)
    assert len(res) == 2000 and all(r[3] == 1 for r in res)


def test_score_width_paths_agree(oracle, tmp_path):
    """u8 (T<=255), u16, and multi-threaded batches give the same counts as a direct bit count"""
    q = oracle.random_sequence(400, 5)
    p = cases.make_compact(cases.tmp(tmp_path, "w.cobs_compact"), 500, 16, [700, 900, 1100, 1300], 2, 31, 1, 0.3, 3)
    ix = oracle.Index.open(p)
    raw = open(p, "rb").read()
    off = ix.data_offset
    mats = []
    for s in (700, 900, 1100, 1300):
        mats.append(np.frombuffer(raw, dtype=np.uint8, count=s * 16, offset=off).reshape(s, 16))
        off += s * 16
    for length in (31, 200, 285, 286, 400):
        qq = q[:length]
        hashes, _ = oracle.term_hashes(qq, 31, 1, 2)
        want = np.zeros(ix.counts_size, dtype=np.uint32)
        for pg, (s, m) in enumerate(zip((700, 900, 1100, 1300), mats)):
            rows = m[(hashes % np.uint64(s)).astype(np.int64)]          # [T, H, 16]
            anded = np.bitwise_and.reduce(rows, axis=1)
            bits = np.unpackbits(anded, axis=1, bitorder="little")       # [T, 128]
            want[pg * 128:(pg + 1) * 128] = bits.sum(axis=0)
        for threads in (1, 3):
            assert np.array_equal(ix.counts(qq, threads=threads), want)


def test_error_conditions(oracle, golden_dir):
    ix = oracle.Index.open(os.path.join(golden_dir, "c1.cobs_classic"))
    with pytest.raises(oracle.OracleError) as e:
        ix.counts(b"ACGT" * 7)
    assert e.value.code == 3
    with pytest.raises(oracle.OracleError) as e:
        ix.counts(Q50[:20] + b"N" + Q50[21:])
    assert e.value.code == 4


def test_query_generators(oracle):
    """cobs::random_sequence = minstd_rand0 % 4; benchmark queries = one mt19937 stream"""
    s = oracle.random_sequence(12, 1)
    x, want = 1, []
    for _ in range(12):
        x = x * 16807 % 2147483647
        want.append(b"ACGT"[x % 4])
    assert s == bytes(want)
    g = oracle.Mt19937Queries(5489)               # default std::mt19937 seed: first output 3499211612
    assert g.next(1) == b"ACGT"[3499211612 % 4:3499211612 % 4 + 1]


def test_survey_probe_hand_built_20_documents(oracle, construct, tmp_path):
    """SURVEY 8c [probed]: known answer of the REAL reference (see cases.survey_probe_files)"""
    pc, pk, q, names, want = cases.survey_probe_files(oracle, construct, tmp_path)
    for p in (pc, pk):
        ix = oracle.Index.open(p)
        res = oracle.search(ix, q)
        assert [(n, s) for (_, _, n, s) in res] == sorted(zip(names, want), key=lambda t: (-t[1], t[0]))
        assert ix.counts(q)[:20].tolist() == want


def test_golden_fixture_hashes(golden_dir):
    """the committed index fixtures are the exact bytes tests/golden/make_golden.py pins"""
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    for name, want in mg.SHA256.items():
        assert hashlib.sha256(open(os.path.join(golden_dir, name), "rb").read()).hexdigest() == want, name


def test_header_writers_follow_the_reference_file_tests(oracle, construct, tmp_path):
    """/root/reference/tests/file.cpp:37-142 restated: what the header writers serialize, the
    readers (the oracle's and, for geometry, libcobs_gpu's host-side parser) give back; the byte
    layout is the one of cobs/file/classic_index_header.cpp:26-37 and
    compact_index_header.cpp:20-43, spelled out here independently."""
    import struct
    # file.cpp:37-60 classic_index_header, :62-90 classic_index (payload bytes all 7)
    names = ["n1", "n2", "n3", "n4"]
    p = str(tmp_path / "h.cobs_classic")
    m = np.full((123, 1), 7, dtype=np.uint8)
    construct.write_classic(p, 31, 1, names, 123, 12, m)
    raw = open(p, "rb").read()
    want = (b"COBS:CLASSIC_INDEX" + struct.pack("<I", 1) + struct.pack("<I", 31) + struct.pack("<B", 1)
            + struct.pack("<I", 4) + struct.pack("<Q", 123) + struct.pack("<Q", 12)
            + b"n1\nn2\nn3\nn4\n" + b"CLASSIC_INDEX" + bytes([7]) * 123)
    assert raw == want
    ix = oracle.Index.open(p)
    assert (ix.term_size, ix.canonicalize, ix.num_hashes, ix.num_docs, ix.row_size) == (31, 1, 12, 4, 1)
    assert [ix.doc_name(i) for i in range(4)] == names and ix.signature_size(0) == 123
    assert ix.data_offset == len(want) - 123
    # file.cpp:92-120 compact_index_header_values: three (signature_size, num_hashes) parameters
    pk = str(tmp_path / "h.cobs_compact")
    params = [(100, 1), (200, 1), (3000, 1)]
    fnames = ["file_1", "file_2", "file_3"]
    mats = [np.zeros((s, 4096), dtype=np.uint8) for s, _ in params]
    construct.write_compact(pk, 31, 1, 4096, params, fnames, mats)
    raw = open(pk, "rb").read()
    head = (b"COBS:COMPACT_INDEX" + struct.pack("<IIBIIQ", 1, 31, 1, 3, 3, 4096)
            + b"".join(struct.pack("<QQ", s, h) for s, h in params) + b"file_1\nfile_2\nfile_3\n")
    assert raw.startswith(head)
    ix = oracle.Index.open(pk)
    assert [ix.signature_size(i) for i in range(3)] == [100, 200, 3000] and ix.num_hashes == 1
    assert [ix.doc_name(i) for i in range(3)] == fnames and ix.page_size == 4096
    # file.cpp:122-142 compact_index_header_padding: the matrix starts at a multiple of page_size,
    # the bytes between the names and the closing magic word are zero
    assert ix.data_offset % 4096 == 0 and raw[ix.data_offset - 13:ix.data_offset] == b"COMPACT_INDEX"
    assert not any(raw[len(head):ix.data_offset - 13])
    for ps, nn in ((8, 1), (8, 5), (16, 3), (4096, 0), (24, 7)):
        h = construct.compact_header(31, 1, ps, [(10, 1)], ["x" * (3 + i) for i in range(nn)])
        assert len(h) % ps == 0 and h.endswith(b"COMPACT_INDEX")
    # the engine's own host-side parser agrees on the geometry (no device needed)
    from cobs_amd import _capi
    import ctypes as C
    for path, total in ((p, 8), (pk, 3 * 8 * 4096)):
        b, c = (C.c_uint64 * 1)(), (C.c_uint64 * 1)()
        _capi.check(_capi.load().cobs_gpu_plan_shards(path.encode(), 1, 0, b, c, None))
        assert (b[0], c[0]) == (0, total)


def test_bit_layout_of_constructed_index(oracle, construct, tmp_path):
    """/root/reference/tests/classic_index_construction.cpp:39-85 restated: 33 documents from
    generate_documents_all, 3 hashes, fpr 0.1; document i is bit i % 8 of byte i / 8 of a row
    (classic_index.cpp:40-43), and no document holds more ones than the expected fill * 1.01"""
    query = oracle.random_sequence(10000, 1)
    docs = construct.generate_documents_all(query, 33, num_hashes=3)
    p = str(tmp_path / "cons.cobs_classic")
    sig = construct.classic_construct(docs, p, num_hashes=3, false_positive_rate=0.1)
    ix = oracle.Index.open(p)
    assert ix.num_docs == 33 and ix.num_hashes == 3 and ix.row_size == 5
    raw = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)[ix.data_offset:].reshape(sig, 5)
    bits = np.unpackbits(raw, axis=1, bitorder="little")[:, :33]          # bit o of byte k = document 8k + o
    ones = bits.sum(axis=0)
    # calc_average_set_bit_ratio(signature_size, 3, 0.1) (util/calc_signature_size.cpp:36-48)
    max_terms = max(d.num_terms for d in docs)
    ratio = 1.0 - (1.0 - 1.0 / sig) ** (3 * max_terms)
    assert ones.max() <= ratio * sig * 1.01
    # every bit of a document is at one of its own hash rows, and every hash row carries its bit
    for j in (0, 1, 7, 8, 15, 16, 32):
        rows = np.unique(docs[j].hashes.reshape(-1) % np.uint64(sig)).astype(np.int64)
        assert bits[rows, j].all()
        assert ones[j] == len(rows)
    # documents beyond 33 (padding bits of the last byte) carry nothing
    assert not np.unpackbits(raw, axis=1, bitorder="little")[:, 33:].any()


def test_indexes_with_different_term_sizes(oracle, tmp_path):
    """ClassicSearch over files with different term sizes (classic_search.cpp:413-449): the
    query must be as long as the LARGEST term size, every file gets its own threshold
    ceil(threshold * (len - k_i + 1)), max_counts is the sum of all files' hashes"""
    q = oracle.random_sequence(260, 91)
    pa = cases.make_classic(cases.tmp(tmp_path, "k31.cobs_classic"), 90, 997, 1, 31, 1, 0.3, 1,
                            planted={3: 1.0, 40: 0.6}, query=q)
    pb = cases.make_classic(cases.tmp(tmp_path, "k21.cobs_classic"), 50, 1201, 2, 21, 1, 0.3, 2,
                            planted={7: 1.0, 11: 0.5}, query=q)
    a, b = oracle.Index.open(pa), oracle.Index.open(pb)
    for qq in (q, q[:31], q[:100]):
        ca, cb = a.counts(qq), b.counts(qq)
        assert ca[3] == len(qq) - 30 and cb[7] == len(qq) - 20            # planted documents hold every term
        for t in (0.0, 0.5, 0.55, 1.0):
            res = oracle.search([a, b], qq, t)
            ta, tb = int(np.ceil(t * (len(qq) - 30))), int(np.ceil(t * (len(qq) - 20)))
            want = [(0, d, int(s)) for d, s in enumerate(ca[:90]) if s >= ta] + \
                   [(1, d, int(s)) for d, s in enumerate(cb[:50]) if s >= tb]
            want.sort(key=lambda h: (-h[2], h[0], h[1]))
            assert [(f, d, s) for (f, d, _n, s) in res] == want, (len(qq), t)
    with pytest.raises(oracle.OracleError):
        oracle.search([a, b], q[:30])          # shorter than the largest term size: "query too short"
    oracle.search([b], q[:21])                 # fine for the k = 21 file alone


QUIRK_FASTAS = {
    # a sequence shorter than k before a header: the reference keeps the buffer and resets pos to
    # 0, so the header line is appended and hashed as sequence (fasta_file.hpp:170-180)
    "short_then_header": b">d\nACGTACGTAC\n>second header with some text\nACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAAC\n",
    # after a run longer than k-1 the buffer holds k-1 characters and pos = k-1: a next line of
    # exactly k-1 characters following a comment line counts as "empty" and is dropped
    "line_of_k_minus_1": b">d\n" + b"ACGTTGCA" * 6 + b"\n>h2\n" + b"ACGTTGCAAC" * 3 + b"\n" + b"TTGACCAGTA" * 5 + b"\n",
    # two comment lines in a row after a sequence: the second is tested at index k-1
    "two_comments": b">d\n" + b"ACGTTGCA" * 6 + b"\n;first comment\n;a second comment line that is longer than thirty characters\n"
                    + b"GGATCCAGTA" * 5 + b"\n",
    "multi_line_sequence": b">d\n" + b"ACGTTGCAAC" * 4 + b"\n" + b"TGCATGCAAA" * 2 + b"\nACG\n\nACGTTGCAACGTTGCAACGTTGCAACGTTGCAACG\n",
    "crlf": b">d\r\n" + b"ACGTTGCAAC" * 4 + b"\r\n" + b"TGCATGCAAA" * 4 + b"\r\n",
    "only_short": b">d\nACGT\n>e\nTTTT\n",
}


def test_fasta_term_state_machine_restated_twice(oracle, construct, tmp_path):
    """FastaFile::process_terms (/root/reference/cobs/fasta_file.hpp:155-182) has edge behaviour
    that decides which bits an index holds (short sequences, comment lines after a sequence).
    The oracle follows the reference's loop literally (std::string buffer + pos); the product's
    reader (cobs_amd/csrc/documents.cpp, host code of libcobs_gpu.so) derives character runs.  Both
    must hash the same terms, and on ordinary FASTA both equal the plain grammar."""
    import cobs_amd
    k = 31
    for name, raw in QUIRK_FASTAS.items():
        p = tmp_path / (name + ".fasta")
        p.write_bytes(raw)
        bufs = list(construct.fasta_term_buffers(str(p), k))
        terms_oracle = [b[i:i + k] for b in bufs for i in range(len(b) - k + 1)]
        dl = cobs_amd.DocumentList()
        dl.add(str(p))
        assert dl[0].terms(k) == terms_oracle, name
        assert dl[0].size == construct.fasta_size(str(p))
        assert dl[0].num_terms(k) == sum(max(len(s) - k + 1, 0) for s in construct.fasta_sequences(str(p))), name
    # the quirks are real: header text is hashed / a line is dropped
    p = tmp_path / "short_then_header.fasta"
    bufs = list(construct.fasta_term_buffers(str(p), k))
    assert any(b">second header" in b for b in bufs)
    p = tmp_path / "line_of_k_minus_1.fasta"
    assert not any(b"ACGTTGCAAC" * 3 in b for b in construct.fasta_term_buffers(str(p), k))
    # ordinary FASTA (the reference's own corpus): identical to maximal runs of sequence lines
    for i in range(1, 8):
        f = os.path.join(os.path.dirname(__file__), "golden", "fasta", "sample%d.fasta" % i)
        f = f if os.path.exists(f) else f + ".gz"
        plain = [s[j:j + k] for s in construct.fasta_sequences(f) for j in range(len(s) - k + 1)]
        got = [b[j:j + k] for b in construct.fasta_term_buffers(f, k) for j in range(len(b) - k + 1)]
        assert got == plain


def test_compact_construction_follows_the_reference_test(oracle, construct, tmp_path):
    """/root/reference/tests/compact_index_construction.cpp:60-169 restated on the oracle's builder:
    33 documents of generate_documents_all, 3 hashes, fpr 0.1, page_size 2 -> three sub-indexes of
    16 documents; names in (size, then path inside a group of 16) order (:66-73, :89-94); every
    document's ones within 1.02 x the expected fill of ITS sub-index (:120-141); and each
    sub-index equals the classic index of the same 16 documents at the same signature size
    (:143-169)."""
    query = oracle.random_sequence(10000, 1)
    docs = construct.generate_documents_all(query, 33, num_hashes=3)
    p = str(tmp_path / "cc.cobs_compact")
    page_size, params = construct.compact_construct(docs, p, num_hashes=3, false_positive_rate=0.1, page_size=2)
    ix = oracle.Index.open(p)
    assert (ix.num_docs, ix.num_pages, ix.page_size, ix.num_hashes) == (33, 3, 2, 3) and page_size == 2
    by_size = sorted(docs, key=lambda d: (d.size, d.path))
    order = []
    for g in range(0, 33, 16):
        order += sorted(by_size[g:g + 16], key=lambda d: d.path)
    assert [ix.doc_name(i) for i in range(33)] == [d.name for d in order]
    raw = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
    off = ix.data_offset
    for g in range(3):
        sig = ix.signature_size(g)
        group = order[16 * g:16 * g + 16]
        assert sig == construct.calc_signature_size(max(d.num_terms for d in group), 3, 0.1) == params[g][0]
        m = raw[off:off + sig * 2].reshape(sig, 2)
        off += sig * 2
        bits = np.unpackbits(m, axis=1, bitorder="little")
        ratio = 1.0 - (1.0 - 1.0 / sig) ** (3 * max(d.num_terms for d in group))
        assert bits.sum(axis=0).max() <= ratio * sig * 1.02
        assert not bits[:, len(group):].any()                       # padding documents of the last group
        pc = str(tmp_path / ("g%d.cobs_classic" % g))
        construct.classic_construct(group, pc, num_hashes=3, signature_size=sig)
        _, _, names, csig, _, cm = construct.read_classic(pc)
        assert csig == sig and names == [d.name for d in group]
        assert np.array_equal(cm, m[:, :cm.shape[1]])               # :163-168 content equality
    assert off == len(raw)
    # queries against it: every document scores at least its true number of terms (compact_index_query.cpp)
    T = len(query) - 30
    got = ix.counts(query)
    idx_of = {d.name: i for i, d in enumerate(order)}
    for d in docs:
        assert got[idx_of[d.name]] >= d.num_terms or d.num_terms > T


def _independent_counts(path, query):
    """A second, independent restatement of the query path (no code shared with oracle/): header
    parsed here, canonicalisation as util/query.cpp:143-199 written from its description, hashes
    from the real xxHash library (python-xxhash), rows ANDed and summed with numpy."""
    import struct
    import xxhash
    raw = open(path, "rb").read()
    comp = {65: 84, 67: 71, 71: 67, 84: 65}

    def canon(kmer):
        n = len(kmer)
        for i in range(n // 2):
            f, r = kmer[i], comp[kmer[n - 1 - i]]
            if f < r:
                return kmer
            if f > r:
                return bytes(comp[c] for c in reversed(kmer))
        return kmer

    if raw.startswith(b"COBS:CLASSIC_INDEX"):
        _, k, can, ndocs, sig, nh = struct.unpack_from("<IIBIQQ", raw, 18)
        pos = 18 + 29
        for _ in range(ndocs):
            pos = raw.index(b"\n", pos) + 1
        pos += 13
        row = (ndocs + 7) // 8
        pages = [(sig, np.frombuffer(raw, np.uint8, sig * row, pos).reshape(sig, row))]
    else:
        assert raw.startswith(b"COBS:COMPACT_INDEX")
        _, k, can, nparams, ndocs, ps = struct.unpack_from("<IIBIIQ", raw, 18)
        pos = 18 + 25
        params = [struct.unpack_from("<QQ", raw, pos + 16 * i) for i in range(nparams)]
        pos += 16 * nparams
        for _ in range(ndocs):
            pos = raw.index(b"\n", pos) + 1
        pos += (ps - ((pos + 13) % ps)) % ps + 13
        nh = params[0][1]
        pages = []
        for sig, _h in params:
            pages.append((sig, np.frombuffer(raw, np.uint8, sig * ps, pos).reshape(sig, ps)))
            pos += sig * ps
    return _counts_from_pages(pages, k, can, nh, query, canon)


def _counts_from_pages(pages, k, can, nh, query, canon):
    import xxhash
    T = len(query) - k + 1
    out = []
    for sig, m in pages:
        acc = np.zeros(m.shape[1] * 8, dtype=np.int64)
        for i in range(T):
            kmer = query[i:i + k]
            if can:
                kmer = canon(kmer)
            rowbits = None
            for j in range(nh):
                r = m[xxhash.xxh64_intdigest(kmer, seed=j) % sig]
                rowbits = r if rowbits is None else (rowbits & r)
            acc += np.unpackbits(rowbits, bitorder="little")
        out.append(acc)
    return np.concatenate(out)


def test_planted_documents_of_the_procedural_index_against_an_independent_restatement(oracle):
    """round 5: the procedural benchmark index gets TRUE POSITIVES (oracle_plant / cobs_gpu_plant: documents that hold the
    terms of a text, a share keep / 1000 of them by a stated rule).  The rule is this project's own, so it is pinned the way
    the counting path is: written out a second time here -- the matrix materialised row by row, python-xxhash, the
    canonicalisation from its description, mix64 in Python integers, bits set as classic_index.cpp:40-43 lays them out -- and
    compared with the C checker's counts, compact and classic, H = 1 and 2"""
    import xxhash
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    M = (1 << 64) - 1

    def canon(kmer):
        n = len(kmer)
        for i in range(n // 2):
            f, r = kmer[i], comp[kmer[n - 1 - i]]
            if f < r:
                return kmer
            if f > r:
                return bytes(comp[c] for c in reversed(kmer))
        return kmer

    def mix64(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    text = oracle.random_sequence(260, 21)
    for kind, H, ps, sigs, D in ((1, 1, 16, [211, 307, 401], 3 * 8 * 16 - 5), (1, 2, 8, [257, 263], 8 * 8 + 3), (0, 1, 0, [499], 77)):
        k, seed, salt = 31, 6, 0xABCDEF
        ix = oracle.Index.synthetic(kind, k, 1, H, ps, sigs, D, seed)
        width = ps if kind else (D + 7) // 8
        pages = [(sg, np.stack([oracle.synth_row(kind, seed, ps, len(sigs), D, p, r, width) for r in range(sg)]))
                 for p, sg in enumerate(sigs)]
        docs = [0, D - 1, D // 2, 9]
        keep = [1000, 700, 333, 0]
        ix.plant(text, docs, keep, salt=salt)
        for d, kp in zip(docs, keep):
            p = d // (8 * ps) if kind else 0
            local = d - p * 8 * ps if kind else d
            sg, m = pages[p]
            for t in range(len(text) - k + 1):
                if mix64(salt ^ ((d << 32) & M) ^ t) % 1000 >= kp:
                    continue
                for j in range(H):
                    m[xxhash.xxh64_intdigest(canon(text[t:t + k]), seed=j) % sg, local // 8] |= 1 << (local % 8)
        for q in (text, text[20:150], text[100:131], oracle.random_sequence(100, 3)):
            want = _counts_from_pages(pages, k, 1, H, q, canon)
            got = ix.counts(q).astype(np.int64)
            assert np.array_equal(got, want), (kind, H)
        full = ix.counts(text)
        T = len(text) - k + 1
        assert full[docs[0]] == T and full[docs[3]] < T // 2 and T // 2 < full[docs[1]] < T


def test_oracle_against_an_independent_restatement(oracle, golden_dir, tmp_path):
    """the C oracle (oracle/cobs_oracle.c) and a from-scratch numpy + python-xxhash restatement
    agree on the golden fixtures and on random classic / compact indexes (H 1..3, canonicalize
    0/1, odd document counts, several sub-indexes)"""
    q50 = b"AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"
    for name in ("c1.cobs_classic", "c1.cobs_compact"):
        p = os.path.join(golden_dir, name)
        assert np.array_equal(_independent_counts(p, q50), oracle.Index.open(p).counts(q50).astype(np.int64))
    q = oracle.random_sequence(180, 5)
    files = [
        cases.make_classic(cases.tmp(tmp_path, "i1.cobs_classic"), 77, 499, 1, 31, 1, 0.3, 1, planted={3: 1.0}, query=q),
        cases.make_classic(cases.tmp(tmp_path, "i2.cobs_classic"), 130, 701, 3, 21, 0, 0.4, 2),
        cases.make_compact(cases.tmp(tmp_path, "i3.cobs_compact"), 3 * 8 * 4 - 5, 4, [211, 307, 401], 2, 31, 1, 0.3, 3,
                           planted={9: 0.7}, query=q),
        cases.make_compact(cases.tmp(tmp_path, "i4.cobs_compact"), 8 * 16 + 3, 16, [257, 263], 1, 15, 1, 0.5, 4),
    ]
    for p in files:
        ix = oracle.Index.open(p)
        for qq in (q, q[:31 if ix.term_size <= 31 else ix.term_size], q[40:140]):
            assert np.array_equal(_independent_counts(p, qq), ix.counts(qq).astype(np.int64)), p


def _independent_results(paths, query, threshold, num_results):
    """counts_to_result restated a second time, on top of _independent_counts (nothing shared with oracle/): per index
    threshold_i = ceil(threshold * T_i) in double (classic_search.cpp:444-449), only real documents pass the filter
    (:127-132, :166-175), the order is score descending then (index, document) ascending -- unless the query has ONE
    hash in total over all indexes, then index order (:134-145, :177-188) -- and num_results == 0 means all (:450)."""
    import math
    import struct
    rows, total_hashes = [], 0
    for f, p in enumerate(paths):
        raw = open(p, "rb").read()
        if raw.startswith(b"COBS:CLASSIC_INDEX"):
            _, k, _can, ndocs, _sig, nh = struct.unpack_from("<IIBIQQ", raw, 18)
        else:
            _, k, _can, nparams, ndocs, _ps = struct.unpack_from("<IIBIIQ", raw, 18)
            nh = struct.unpack_from("<QQ", raw, 18 + 25)[1]
        T = len(query) - k + 1
        total_hashes += T * nh
        thr = math.ceil(threshold * T)
        counts = _independent_counts(p, query)
        rows += [(f, d, int(counts[d])) for d in range(ndocs) if counts[d] >= thr]
    if total_hashes > 1:
        rows.sort(key=lambda r: (-r[2], r[0], r[1]))
    return rows[:num_results] if num_results else rows


def test_oracle_ranking_against_an_independent_restatement(oracle, tmp_path):
    """the ORDER of the results -- thresholds, limits that cut runs of equal scores, several indexes with different term
    sizes, the single-hash case -- from a second restatement of counts_to_result against the C oracle's (both written
    from reference classic_search.cpp:109-202, 444-451; the hit counts below them are pinned the same way above)"""
    q = oracle.random_sequence(140, 77)
    a = cases.make_compact(cases.tmp(tmp_path, "r1.cobs_compact"), 2 * 8 * 8 + 5, 8, [97, 131, 151], 1, 31, 1, 0.55, 5,
                           planted={3: 1.0, 40: 0.8}, query=q)                 # dense filter, few rows: many equal scores
    b = cases.make_classic(cases.tmp(tmp_path, "r2.cobs_classic"), 61, 89, 2, 21, 1, 0.6, 6, planted={7: 0.9}, query=q)
    c = cases.make_classic(cases.tmp(tmp_path, "r3.cobs_classic"), 33, 53, 1, 31, 0, 0.5, 7)
    for paths in ([a], [b], [a, b], [c, a, b], [c]):
        ixs = [oracle.Index.open(p) for p in paths]
        kmax = max(ix.term_size for ix in ixs)
        for qq in (q, q[:kmax], q[:kmax + 1], q[10:10 + kmax + 7], q[5:100]):
            for t in (0.0, 0.3, 0.5, 0.9, 1.0):
                for lim in (0, 1, 2, 5, 17, 1000):
                    got = [(f, d, s) for (f, d, _n, s) in oracle.search(ixs, qq, t, lim)]
                    assert got == _independent_results(paths, qq, t, lim), (paths, len(qq), t, lim)
