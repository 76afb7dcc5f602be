"""GPU index construction: files written by cobs_gpu_build_* are
byte-identical to the golden fixtures / to the numpy construction restatement, and the
reference's own Python test (python/tests/test_cobs_index.py) passes against the mirror."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

Q50 = "AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"


def test_reference_python_test_flow(gpu_lib, golden_dir, tmp_path):
    """python/tests/test_cobs_index.py:13-61 with `import cobs_amd as cobs`"""
    cobs = gpu_lib
    datadir = os.path.join(golden_dir, "fasta")
    cobs.disable_cache()
    l1 = cobs.DocumentList(datadir)
    assert l1.size() == 7
    l2 = cobs.DocumentList()
    l2.add_recursive(datadir)
    assert l2.size() == 7
    index_file = str(tmp_path / "python_test.cobs_classic")
    p = cobs.ClassicIndexParameters()
    p.clobber = True
    cobs.classic_construct(input=datadir, out_file=index_file, index_params=p)
    assert os.path.isfile(index_file)
    r = cobs.Search(index_file).search(Q50)
    assert len(r) == 7 and r[0].doc_name == "sample1" and r[0].score == 20
    index_file = str(tmp_path / "python_test.cobs_compact")
    p = cobs.CompactIndexParameters()
    p.clobber = True
    cobs.compact_construct(input=datadir, out_file=index_file, index_params=p)
    r = cobs.Search(index_file).search(Q50)
    assert len(r) == 7 and r[0].doc_name == "sample1" and r[0].score == 20


def test_files_equal_golden_fixtures(gpu_lib, golden_dir, tmp_path):
    for name, fn, params in (("c1.cobs_classic", gpu_lib.classic_construct, gpu_lib.ClassicIndexParameters()),
                             ("c1.cobs_compact", gpu_lib.compact_construct, gpu_lib.CompactIndexParameters())):
        out = str(tmp_path / name)
        fn(input=os.path.join(golden_dir, "fasta"), out_file=out, index_params=params)
        assert open(out, "rb").read() == open(os.path.join(golden_dir, name), "rb").read(), name
    # the *_construct_list variants (python/module.cpp:253-268, :332-347) on a populated DocumentList
    for name, fn, params in (("c1.cobs_classic", gpu_lib.classic_construct_list, gpu_lib.ClassicIndexParameters()),
                             ("c1.cobs_compact", gpu_lib.compact_construct_list, gpu_lib.CompactIndexParameters())):
        dl = gpu_lib.DocumentList()
        dl.add_recursive(os.path.join(golden_dir, "fasta"))
        assert dl.size() == 7
        params.clobber = True
        out = str(tmp_path / name)
        fn(dl, out, params)
        assert open(out, "rb").read() == open(os.path.join(golden_dir, name), "rb").read(), name


def _random_docs(oracle, n, seed):
    rng = np.random.default_rng(seed)
    docs = []
    for d in range(n):
        seqs = []
        for s in range(int(rng.integers(1, 4))):
            q = bytearray(oracle.random_sequence(int(rng.integers(5, 400)), seed * 1000 + d * 10 + s))
            if rng.random() < 0.3 and len(q) > 40:
                q[int(rng.integers(0, len(q)))] = ord("N")           # invalid base inside a document
            seqs.append(bytes(q))
        docs.append(("doc_%04d" % d, seqs))
    return docs


@pytest.mark.parametrize("canonicalize,num_hashes,k", [(1, 1, 31), (1, 3, 31), (0, 2, 31), (1, 2, 20), (0, 1, 40)])
def test_in_memory_documents_vs_restatement(gpu_lib, oracle, construct, tmp_path, canonicalize, num_hashes, k):
    docs = _random_docs(oracle, 75, 7 + k)
    dl = gpu_lib.DocumentList()
    kdocs = []
    for name, seqs in docs:
        dl.add_document(name, seqs)
        hs = [oracle.term_hashes(s, k, canonicalize, num_hashes)[0] for s in seqs if len(s) >= k]
        hashes = np.concatenate(hs) if hs else np.zeros((0, num_hashes), dtype=np.uint64)
        text_len = len(b"\n".join(seqs)) + 1
        kdocs.append(construct.Doc(name, name, text_len, sum(max(len(s) - k + 1, 0) for s in seqs), hashes))
    pc = gpu_lib.ClassicIndexParameters()
    pc.term_size, pc.canonicalize, pc.num_hashes, pc.false_positive_rate = k, canonicalize, num_hashes, 0.1
    got, want = str(tmp_path / "g.cobs_classic"), str(tmp_path / "w.cobs_classic")
    gpu_lib.classic_construct(list=dl, out_file=got, index_params=pc)
    construct.classic_construct(kdocs, want, term_size=k, canonicalize=canonicalize, num_hashes=num_hashes,
                                false_positive_rate=0.1)
    assert open(got, "rb").read() == open(want, "rb").read()
    pk = gpu_lib.CompactIndexParameters()
    pk.term_size, pk.canonicalize, pk.num_hashes, pk.false_positive_rate, pk.page_size = k, canonicalize, num_hashes, 0.1, 2
    got, want = str(tmp_path / "g.cobs_compact"), str(tmp_path / "w.cobs_compact")
    gpu_lib.compact_construct(list=dl, out_file=got, index_params=pk)
    construct.compact_construct(kdocs, want, term_size=k, canonicalize=canonicalize, num_hashes=num_hashes,
                                false_positive_rate=0.1, page_size=2)
    assert open(got, "rb").read() == open(want, "rb").read()
    # and the built index answers like the oracle
    s = gpu_lib.Search(got)
    ix = oracle.Index.open(got)
    q = docs[3][1][0] if len(docs[3][1][0]) >= k and b"N" not in docs[3][1][0] else oracle.random_sequence(100, 1)
    if canonicalize == 0 or b"N" not in q:
        assert np.array_equal(s.counts(q), ix.counts(q))


def test_write_synthetic_file_is_the_procedural_index(gpu_lib, oracle, tmp_path):
    """cobs_gpu_write_synthetic (the generator tool): the file it writes holds exactly the rows the
    checker's generator defines, in the reference's header format (the oracle's reader opens it),
    and the engine answers from the file -- resident or streamed -- as from the in-HBM original."""
    import cobs_amd
    from tests import cases
    sigs = [1201, 1789, 2503]
    ps, D = 112, 2 * 8 * 112 + 500
    pc = str(tmp_path / "syn.cobs_compact")
    cobs_amd.write_synthetic(pc, "compact", sigs, D, page_size=ps, seed=77)
    ix = oracle.Index.open(pc)
    assert (ix.num_pages, ix.page_size, ix.num_docs) == (3, ps, D)
    assert ix.doc_name(0) == "file_000000" and ix.doc_name(D - 1) == "file_%06d" % (D - 1)
    ref = oracle.Index.synthetic(1, 31, 1, 1, ps, sigs, D, 77)
    s_file = gpu_lib.Search(pc)
    s_stream = gpu_lib.Search(pc, hbm_budget=300 * 1024)
    s_mem = gpu_lib.Search.synthetic("compact", sigs, D, page_size=ps, seed=77)
    for page, row in ((0, 0), (0, 1200), (1, 17), (2, 2502)):
        assert np.array_equal(s_file.read_row(0, page, row, ps), oracle.synth_row(1, 77, ps, 3, D, page, row, ps))
    for q in cases.queries_acgt(3, 400, 40):
        want = ref.counts(q)
        assert np.array_equal(ix.counts(q), want)
        for s in (s_file, s_stream, s_mem):
            assert np.array_equal(s.counts(q), want)
    pk = str(tmp_path / "syn.cobs_classic")
    cobs_amd.write_synthetic(pk, "classic", [3001], 1003, seed=5)       # row bytes not a multiple of 8
    ref2 = oracle.Index.synthetic(0, 31, 1, 1, 0, [3001], 1003, 5)
    s2 = gpu_lib.Search(pk)
    for q in cases.queries_acgt(2, 300, 50):
        assert np.array_equal(oracle.Index.open(pk).counts(q), ref2.counts(q))
        assert np.array_equal(s2.counts(q), ref2.counts(q))


def test_construct_follows_process_terms_on_odd_fasta(gpu_lib, oracle, construct, tmp_path):
    """documents whose sequences are shorter than k, comment lines after sequences, CRLF: the GPU
    builder's files equal the oracle's construction byte for byte (both follow the reference's
    process_terms state machine, fasta_file.hpp:155-182; the signature size comes from num_terms)"""
    import cobs_amd
    from tests.test_oracle_pins import QUIRK_FASTAS
    d = tmp_path / "odd"
    d.mkdir()
    for name, raw in QUIRK_FASTAS.items():
        (d / (name + ".fasta")).write_bytes(raw)
    for ext, build_gpu, build_ref in (
            (".cobs_classic", cobs_amd.classic_construct, construct.classic_construct),
            (".cobs_compact", cobs_amd.compact_construct, construct.compact_construct)):
        pg, pr = str(tmp_path / ("g" + ext)), str(tmp_path / ("r" + ext))
        params = cobs_amd.ClassicIndexParameters() if "classic" in ext else cobs_amd.CompactIndexParameters()
        params.false_positive_rate = 0.1
        params.num_hashes = 2
        build_gpu(str(d), pg, params)
        docs = construct.fasta_dir_docs(str(d), 31, 1, 2)
        build_ref(docs, pr, num_hashes=2, false_positive_rate=0.1)
        assert open(pg, "rb").read() == open(pr, "rb").read(), ext


def test_combine_on_the_device(gpu_lib, oracle, construct, tmp_path):
    """classic_combine (classic_index.cpp:195-327): byte-aligned and unaligned document counts,
    one batch and several row batches; byte-identical to the numpy restatement; the combined
    index answers like the union of its parts"""
    import cobs_amd
    q = oracle.random_sequence(200, 12)
    for tag, counts in (("aligned", [16, 8, 24, 5]), ("unaligned", [5, 9, 3, 1, 14]), ("single", [11])):
        parts = []
        for i, n in enumerate(counts):
            p = cases.make_classic(cases.tmp(tmp_path, "%s%d.cobs_classic" % (tag, i)), n, 977, 2, 31, 1, 0.3, 30 + i,
                                   planted={0: 1.0}, query=q)
            parts.append(p)
        want = str(tmp_path / (tag + "_ref.cobs_classic"))
        construct.classic_combine(parts, want)
        for mem in (0, 300, 4096):                      # 0 = one batch; 300 bytes = a few rows per batch
            got = str(tmp_path / ("%s_gpu_%d.cobs_classic" % (tag, mem)))
            cobs_amd.classic_combine(parts, got, mem_bytes=mem)
            assert open(got, "rb").read() == open(want, "rb").read(), (tag, mem)
        s = gpu_lib.Search(want)
        c = s.counts(q)
        off = 0
        for p, n in zip(parts, counts):
            assert np.array_equal(c[off:off + n], oracle.Index.open(p).counts(q)[:n])
            off += n
    # mismatching parameters are refused
    other = cases.make_classic(cases.tmp(tmp_path, "other.cobs_classic"), 8, 983, 2, 31, 1, 0.3, 77)
    with pytest.raises(cobs_amd.CobsGpuError):
        cobs_amd.classic_combine([parts[0], other], str(tmp_path / "bad.cobs_classic"))


def test_construct_random_tool(gpu_lib, oracle, construct, tmp_path):
    """classic_construct_random (classic_index.cpp:661-725): the file equals the numpy restatement of
    the same generator byte for byte; fill ratio as the reference's "ratio of ones" (1 - e^(-mH/S))"""
    import cobs_amd
    got, want = str(tmp_path / "rg.cobs_classic"), str(tmp_path / "rr.cobs_classic")
    cobs_amd.classic_construct_random(got, signature_size=1999, num_documents=21, document_size=150, num_hashes=2, seed=5)
    construct.classic_construct_random(want, 1999, 21, 150, 2, 5)
    assert open(got, "rb").read() == open(want, "rb").read()
    big = str(tmp_path / "big.cobs_classic")
    cobs_amd.classic_construct_random(big, signature_size=200003, num_documents=203, document_size=70000, num_hashes=1, seed=9)
    k, canon, names, sig, nh, m = construct.read_classic(big)
    assert (k, canon, sig, nh, len(names)) == (31, 1, 200003, 1, 203) and names[202] == "file_000202"
    bits = np.unpackbits(m, axis=1, bitorder="little")[:, :203]
    fill = bits.mean(axis=0)
    expect = 1.0 - np.exp(-70000 / 200003)
    assert np.all(np.abs(fill - expect) < 0.01)
    assert not np.unpackbits(m, axis=1, bitorder="little")[:, 203:].any()
    s = gpu_lib.Search(big)                                  # and the engine reads it
    assert s.info(0).num_docs == 203


def test_build_into_hbm_and_text_batches(gpu_lib, oracle, construct, golden_dir, tmp_path):
    """cobs_gpu_build_index: construction straight into a resident query handle (no file) answers
    exactly like the file-built index; tiny text batches (every document its own upload, the
    batching of classic_index.cpp:565-659) give the same bytes"""
    import cobs_amd
    fasta = os.path.join(golden_dir, "fasta")
    Q50 = b"AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"
    for kind, golden in (("classic", "c1.cobs_classic"), ("compact", "c1.cobs_compact")):
        ix = oracle.Index.open(os.path.join(golden_dir, golden))
        s = cobs_amd.build_search(fasta, kind=kind)
        assert s.info(0).num_docs == 7 and s.signature_size(0, 0) == 8748
        assert np.array_equal(s.counts(Q50), ix.counts(Q50))
        assert [(r.doc_name, r.score) for r in s.search(Q50.decode())] == \
            [(n, sc) for (_, _, n, sc) in oracle.search(ix, Q50)]
        qs = cases.queries_acgt(4, 120, 3)
        assert s.search_hits(qs, 0.0, 3) == [cases.oracle_results([ix], q, 0.0, 3) for q in qs]
        params = cobs_amd.ClassicIndexParameters() if kind == "classic" else cobs_amd.CompactIndexParameters()
        params.text_batch_bytes = 700                            # smaller than any document: one document per batch
        out = str(tmp_path / ("b." + ("cobs_" + kind)))
        (cobs_amd.classic_construct if kind == "classic" else cobs_amd.compact_construct)(fasta, out, params)
        assert open(out, "rb").read() == open(os.path.join(golden_dir, golden), "rb").read()


def test_reference_construction_corpus_on_the_gpu(gpu_lib, oracle, construct, tmp_path):
    """the corpus of /root/reference/tests/compact_index_construction.cpp / classic_index_construction.cpp
    (generate_documents_all: 33 documents, 3 hashes, fpr 0.1, compact page_size 2) built by the GPU
    builder from in-memory documents: the files equal the oracle's byte for byte, and queries
    against them answer as the oracle does"""
    import cobs_amd
    query = oracle.random_sequence(10000, 1)
    docs = construct.generate_documents_all(query, 33, num_hashes=3)
    n = min(1000000, len(query) - 31)
    members = [[] for _ in range(33)]
    for i in range(n):
        for j in range(0, 33, i % 32 + 1):
            members[j].append(i)
    dl = cobs_amd.DocumentList()
    for d, m in zip(docs, members):
        dl.add_document(d.name, [query[i:i + 31] for i in m])
        assert len(m) == d.num_terms
    for e, d in zip(dl, docs):
        e.path = d.path                                      # same path order as the oracle's documents
    for kind in ("classic", "compact"):
        want, got = str(tmp_path / ("o.cobs_" + kind)), str(tmp_path / ("g.cobs_" + kind))
        if kind == "classic":
            construct.classic_construct(docs, want, num_hashes=3, false_positive_rate=0.1)
            params = cobs_amd.ClassicIndexParameters()
        else:
            construct.compact_construct(docs, want, num_hashes=3, false_positive_rate=0.1, page_size=2)
            params = cobs_amd.CompactIndexParameters()
            params.page_size = 2
        params.num_hashes, params.false_positive_rate = 3, 0.1
        (cobs_amd.classic_construct_list if kind == "classic" else cobs_amd.compact_construct_list)(dl, got, params)
        assert open(got, "rb").read() == open(want, "rb").read(), kind
        s = gpu_lib.Search(got)
        ix = oracle.Index.open(want)
        for q in (query[:500], query[2000:2100], query):
            assert np.array_equal(s.counts(q), ix.counts(q))


def test_drop_in_module_name(gpu_lib, golden_dir, tmp_path):
    """`import cobs_index as cobs` (the reference's module name): the flow of the reference's
    own python/tests/test_cobs_index.py, statement for statement, against this engine"""
    import cobs_index as cobs
    datadir = os.path.join(golden_dir)
    cobs.disable_cache()
    l1 = cobs.DocumentList(datadir + "/fasta")
    assert l1.size() == 7
    l2 = cobs.DocumentList()
    l2.add_recursive(datadir + "/fasta")
    assert l2.size() == 7
    for ext, params, construct_fn in ((".cobs_classic", cobs.ClassicIndexParameters, cobs.classic_construct),
                                      (".cobs_compact", cobs.CompactIndexParameters, cobs.compact_construct)):
        index_file = str(tmp_path / ("python_test" + ext))
        p = params()
        p.clobber = True
        construct_fn(input=datadir + "/fasta", out_file=index_file, index_params=p)
        assert os.path.isfile(index_file)
        s = cobs.Search(index_file)
        r = s.search("AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT")
        assert len(r) == 7
        assert r[0].doc_name == "sample1"
        assert r[0].score == 20
    assert isinstance(cobs.__version__, str)


# ---- document lists: every reader of the reference in front of the GPU builder ---------------------

def _build_both(cobs_amd, construct, D, root, tmp_path, tag, k=31, canonicalize=1, num_hashes=1, fpr=0.3,
                page_size=0, filter=0, batch=0):
    """GPU files from the library's own DocumentList vs the checker's construction of the checker's
    document list; -> paths of the two GPU files"""
    ents = D.document_list(root, filter)
    kdocs = D.docs(ents, k, canonicalize, num_hashes)
    out = []
    for ext, build_gpu, build_ref, params in (
            (".cobs_classic", cobs_amd.classic_construct, construct.classic_construct, cobs_amd.ClassicIndexParameters()),
            (".cobs_compact", cobs_amd.compact_construct, construct.compact_construct, cobs_amd.CompactIndexParameters())):
        params.term_size, params.canonicalize, params.num_hashes, params.false_positive_rate = k, canonicalize, num_hashes, fpr
        params.text_batch_bytes = batch
        kw = {}
        if "compact" in ext:
            params.page_size = page_size
            kw["page_size"] = page_size
        pg, pr = str(tmp_path / (tag + "_g" + ext)), str(tmp_path / (tag + "_r" + ext))
        build_ref(kdocs, pr, term_size=k, canonicalize=canonicalize, num_hashes=num_hashes, false_positive_rate=fpr, **kw)
        # both ways of setting bits: atomicOr into the matrix (1), byte planes + packing pass (2)
        for mode in (1, 2):
            params.set_bits_mode = mode
            params.clobber = True
            build_gpu(cobs_amd.DocumentList(root, filter), pg, params)
            assert open(pg, "rb").read() == open(pr, "rb").read(), (tag, ext, mode)
        out.append(pg)
    return out


@pytest.mark.parametrize("sub,canonicalize", [("fastq", 0), ("fasta_multi", 0), ("text", 0), ("cortex", 1),
                                              ("fastq", 1), ("fasta_multi", 1), ("text", 1), ("", 1)])
def test_index_from_each_document_type(gpu_lib, oracle, construct, golden_dir, tmp_path, sub, canonicalize):
    """the reference's per-format construction tests (tests/fastq_file.cpp:59-101,
    fasta_multifile.cpp:54-99: num_hashes 3, fpr 0.1, canonicalize 0; every k-mer of a document
    finds its document) on the reference's own data files, plus the mixed tree ("" = all of
    tests/golden/documents: text beyond the 64 KiB reader buffer, cortex files of three k-mer
    sizes, gz): files byte-identical to the checker's construction"""
    from oracle import documents as D
    root = os.path.join(golden_dir, "documents", sub) if sub else os.path.join(golden_dir, "documents")
    pc, pk = _build_both(gpu_lib, construct, D, root, tmp_path, sub or "all", canonicalize=canonicalize,
                         num_hashes=3, fpr=0.1, page_size=2 if sub else 0)
    ents = D.document_list(root)
    for p in (pc, pk):
        s = gpu_lib.Search(p)
        names = [s.doc_name(0, d) for d in range(s.info(0).num_docs)]
        for e in ents:
            if sub == "":
                break
            terms = e.terms(31)[:200]
            if canonicalize:
                terms = [t for t in terms if set(t) <= set(b"ACGT")]      # the query path rejects other characters
            for t, res in zip(terms, s.search_hits(terms, 0.0, 0)):
                assert len(res) == len(ents)
                # (names repeat in the cortex tree: three files carry the sample name "sample1")
                assert max(sc for (_, d, sc) in res if names[d] == e.name) >= 1, (e.name, t)


def test_index_from_generated_documents_small_batches(gpu_lib, oracle, construct, tmp_path):
    """the generated corpus of tests/test_documents.py (reader edge cases of every type) through
    the GPU builder with text batches of 4 KiB -- documents larger than a batch, many launches,
    both staging buffers in flight -- and with the default batch"""
    from oracle import documents as D
    from tests.test_documents import _write_corpus
    root = str(tmp_path / "corpus")
    _write_corpus(np.random.default_rng(int(os.environ.get("COBS_FUZZ_SEED", "77"))), root)   # soak: other seeds
    for tag, batch, canon, nh in (("small", 4096, 1, 2), ("default", 0, 0, 1)):
        _build_both(gpu_lib, construct, D, root, tmp_path, tag, canonicalize=canon, num_hashes=nh, fpr=0.2,
                    page_size=1, batch=batch)
    # other term sizes (the generic hashing path of build_kernel; .cobs_doc documents hold 31-mers only)
    for k in (20, 32):
        _build_both(gpu_lib, construct, D, root, tmp_path, "k%d" % k, k=k, num_hashes=2, fpr=0.2, page_size=2,
                    filter=D.TEXT)
        _build_both(gpu_lib, construct, D, root, tmp_path, "kq%d" % k, k=k, num_hashes=1, fpr=0.2, page_size=2,
                    filter=D.FASTQ, batch=1000)


def test_reference_corpus_as_cobs_doc_files(gpu_lib, oracle, construct, tmp_path):
    """tests/classic_index_construction.cpp:39-72 / compact_index_construction.cpp: the corpus of
    generate_documents_all written as .cobs_doc files (generate_test_case, tests/test_util.hpp:86-96),
    indexed from the directory -- equal to the checker's index of the same corpus"""
    from oracle import documents as D
    query = oracle.random_sequence(10000, 1)
    docs = construct.generate_documents_all(query, 33, num_hashes=3)
    members = [[] for _ in range(33)]
    for i in range(len(query) - 31):
        for j in range(0, 33, i % 32 + 1):
            members[j].append(i)
    root = tmp_path / "docs"
    root.mkdir()
    for j, m in enumerate(members):
        D.write_kmer_buffer(str(root / ("document_%06d.cobs_doc" % j)), "document_%06d" % j,
                            [oracle.canonicalize_kmer(query[i:i + 31])[0] for i in m])
    dl = gpu_lib.DocumentList(str(root))
    assert dl.size() == 33 and [d.term_count for d in dl] == [d.num_terms for d in docs]
    pc = gpu_lib.ClassicIndexParameters()
    pc.num_hashes, pc.false_positive_rate = 3, 0.1
    got, want = str(tmp_path / "g.cobs_classic"), str(tmp_path / "w.cobs_classic")
    gpu_lib.classic_construct(str(root), got, pc)
    construct.classic_construct(docs, want, num_hashes=3, false_positive_rate=0.1)
    assert open(got, "rb").read() == open(want, "rb").read()
    pk = gpu_lib.CompactIndexParameters()
    pk.num_hashes, pk.false_positive_rate, pk.page_size = 3, 0.1, 2
    got, want = str(tmp_path / "g.cobs_compact"), str(tmp_path / "w.cobs_compact")
    gpu_lib.compact_construct(str(root), got, pk)
    construct.compact_construct(docs, want, num_hashes=3, false_positive_rate=0.1, page_size=2)
    assert open(got, "rb").read() == open(want, "rb").read()
    # and straight into a query handle: no file in between
    s = gpu_lib.build_search(str(root), pk, kind="compact")
    ix = oracle.Index.open(want)
    assert np.array_equal(s.counts(query[:1000]), ix.counts(query[:1000]))


def test_classic_signature_comes_from_the_largest_file(gpu_lib, oracle, construct, tmp_path):
    """get_max_file_size (classic_index.cpp:521-563): the signature is sized by num_terms of the
    largest document BY SIZE, not by the largest term count -- a FASTA file that is mostly comments
    is larger than one with more sequence"""
    from oracle import documents as D
    root = tmp_path / "two"
    root.mkdir()
    (root / "a.fasta").write_bytes(b">a\n" + oracle.random_sequence(300, 1) + b"\n")
    (root / "b.fasta").write_bytes(b">b\n" + b";comment line, no sequence\n" * 30 + oracle.random_sequence(60, 2) + b"\n")
    ents = D.document_list(str(root))
    assert ents[1].size > ents[0].size and ents[1].num_terms(31) < ents[0].num_terms(31)
    pc, _ = _build_both(gpu_lib, construct, D, str(root), tmp_path, "two", page_size=1)
    assert oracle.Index.open(pc).signature_size(0) == construct.calc_signature_size(ents[1].num_terms(31), 1, 0.3)


def test_build_buffers_are_kept_and_can_be_released(gpu_lib, golden_dir):
    """the builders keep their staging memory for the next build of the process
    (cobs_gpu_build_release_buffers frees it): device memory returns to where it was"""
    import torch
    from cobs_amd import construct as C
    C.release_build_buffers()
    free0 = torch.cuda.mem_get_info()[0]
    C.build_search(os.path.join(golden_dir, "fasta")).close()
    held = free0 - torch.cuda.mem_get_info()[0]
    assert held >= (200 << 20)                     # one 256 MiB device text buffer at least
    C.build_search(os.path.join(golden_dir, "fasta")).close()
    assert free0 - torch.cuda.mem_get_info()[0] <= held + (64 << 20)      # reused, not re-allocated
    C.release_build_buffers()
    assert abs(free0 - torch.cuda.mem_get_info()[0]) < (64 << 20)
