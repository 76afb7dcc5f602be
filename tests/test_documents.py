"""Document lists and document readers (input side), CPU only.

1. Pins of the checker oracle/documents.py against the reference's own reader tests and their
   data files (tests/cortex_file.cpp, fastq_file.cpp, fasta_multifile.cpp, text_file.cpp; data in
   tests/golden/documents/, copied by tests/golden/copy_reference_documents.py).
2. The product's readers (cobs_amd/csrc/documents.cpp through cobs_gpu_doclist_*; host code of
   libcobs_gpu.so, no device call) against those known answers and, term for term, against the
   checker on the fixtures and on generated files that exercise the readers' buffer edges."""
import gzip
import os

import numpy as np
import pytest

from oracle import documents as D


@pytest.fixture(scope="module")
def docdir(golden_dir):
    return os.path.join(golden_dir, "documents")


@pytest.fixture(scope="module")
def capi():
    import cobs_amd
    return cobs_amd


def _lines(path):
    return open(path, "rb").read().split(b"\n")[:-1]


# ---- 1. the checker against the reference's expectations ---------------------------------------

def test_oracle_cortex_reference_tests(docdir):
    """tests/cortex_file.cpp:22-52 (header fields, 24158 k-mers, sorted list) and :54-88 (sample1
    at k = 31, 19, 15: the k-mers in file order)"""
    p = os.path.join(docdir, "cortex", "document.ctx")
    h = D.cortex_header(p)
    assert (h["version"], h["kmer_size"], h["words"], h["colors"], h["name"]) == (6, 31, 1, 1, "DRR030535")
    assert h["num_kmers"] == 24158
    kmers = D.windows(D.cortex_term_buffers(p, 31), 31)
    assert len(kmers) == 24158
    assert sorted(kmers) == _lines(os.path.join(docdir, "cortex", "document_sorted.txt"))
    for k in (31, 19, 15):
        got = D.windows(D.cortex_term_buffers(os.path.join(docdir, "cortex", "sample1-k%d.ctx" % k), k), k)
        assert got == _lines(os.path.join(docdir, "cortex", "sample1-k%d.txt" % k))


def test_oracle_fastq_reference_tests(docdir):
    """tests/fastq_file.cpp:38-57, :59-64"""
    s1, _ = D.fastq_index(os.path.join(docdir, "fastq", "sample1.fastq"))
    s2, h2 = D.fastq_index(os.path.join(docdir, "fastq", "sample2.fastq.gz"))
    assert (s1, s2) == (3518, 3001)
    e = D.load(os.path.join(docdir, "fastq", "sample2.fastq.gz"))[0]
    assert e.num_terms(31) == len(e.terms(31)) > 0
    assert len(D.document_list(os.path.join(docdir, "fastq"))) == 3


def test_oracle_multifasta_reference_tests(docdir):
    """tests/fasta_multifile.cpp:36-52, :54-60"""
    assert len(D.mfasta_index(os.path.join(docdir, "fasta_multi", "sample1.mfasta"))) == 1
    ix = D.mfasta_index(os.path.join(docdir, "fasta_multi", "sample2.mfasta"))
    assert len(ix) == 5 and ix[0][1] == 256 and ix[4][1] == 438
    terms = D.windows(D.mfasta_term_buffers(os.path.join(docdir, "fasta_multi", "sample2.mfasta"), ix[0][0], 31), 31)
    assert len(terms) == 256 - 30
    ents = D.document_list(os.path.join(docdir, "fasta_multi"))
    assert len(ents) == 6
    assert [e.name for e in ents] == ["sample1_000000"] + ["sample2_%06d" % i for i in range(5)]


def test_oracle_text_reference_tests(docdir):
    """tests/text_file.cpp:17-30"""
    p = os.path.join(docdir, "text", "sample1.txt")
    assert os.path.getsize(p) == 76
    assert len(D.windows(D.text_term_buffers(p, 31), 31)) == 76 - 30
    assert len(D.document_list(os.path.join(docdir, "text"))) == 2


def test_oracle_kmer_buffer_round_trip(tmp_path):
    """KMer<31>::init / to_string are inverses (kmer.hpp:54-99); a .cobs_doc holds its header name
    and (file size - header) / 8 k-mers (document_list.hpp:271-284)"""
    rng = np.random.default_rng(5)
    kmers = [bytes(rng.choice(list(b"ACGT"), 31).astype(np.uint8)) for _ in range(57)]
    for km in kmers:
        assert D.kmer_to_string(D.kmer_pack(km), 31) == km
    assert D.kmer_pack(b"A" * 30 + b"C") == bytes([1, 0, 0, 0, 0, 0, 0, 0])      # last four bases in byte 0
    assert D.kmer_pack(b"T" + b"A" * 30) == bytes([0, 0, 0, 0, 0, 0, 0, 0x30])   # 'A' pad, then the first three
    p = str(tmp_path / "d.cobs_doc")
    D.write_kmer_buffer(p, "document_000007", kmers)
    e = D.load(p)[0]
    assert (e.name, e.term_size, e.term_count, e.num_terms(31)) == ("document_000007", 31, 57, 57)
    assert e.terms(31) == kmers


# ---- 2. the product's readers ---------------------------------------------------------------------

def _same_entry(e, g):
    assert (e.path, e.name, e.type, e.size, e.subdoc_index, e.term_size, e.term_count) == \
        (g.path, g.name, int(g.type), g.size, g.subdoc_index, g.term_size, g.term_count)


def test_native_readers_reference_known_answers(capi, docdir):
    dl = capi.DocumentList()
    dl.add(os.path.join(docdir, "cortex", "document.ctx"))
    e = dl[0]
    assert (e.name, e.term_size, e.term_count, e.type) == ("DRR030535", 31, 24158, capi.FileType.Cortex)
    assert sorted(e.terms(31)) == _lines(os.path.join(docdir, "cortex", "document_sorted.txt"))
    for k in (31, 19, 15):
        dl = capi.DocumentList(os.path.join(docdir, "cortex", "sample1-k%d.ctx" % k))
        assert dl[0].terms(k) == _lines(os.path.join(docdir, "cortex", "sample1-k%d.txt" % k))
    dl = capi.DocumentList(os.path.join(docdir, "fastq"))
    assert [d.size for d in dl] == [3518, 3001, 4000] and [d.name for d in dl] == ["sample1", "sample2", "sample3"]
    assert dl[1].num_terms(31) == len(dl[1].terms(31))
    dl = capi.DocumentList(os.path.join(docdir, "fasta_multi"))
    assert dl.size() == 6 and dl[1].size == 256 and dl[5].size == 438 and len(dl[1].terms(31)) == 256 - 30
    dl = capi.DocumentList(os.path.join(docdir, "text"))
    assert dl.size() == 2 and dl[0].size == 76 and len(dl[0].terms(31)) == 76 - 30


@pytest.mark.parametrize("sub", ["cortex", "fastq", "fasta_multi", "text", "../fasta"])
def test_native_readers_equal_the_checker_on_the_fixtures(capi, docdir, sub):
    root = os.path.normpath(os.path.join(docdir, sub))
    ents, dl = D.document_list(root), capi.DocumentList(root)
    assert dl.size() == len(ents) > 0
    for e, g in zip(ents, dl):
        _same_entry(e, g)
        for k in (31, 15, 4, 1, 40):
            assert e.num_terms(k) == g.num_terms(k)
            assert e.terms(k) == g.terms(k), (e.path, k)


def _random_lines(rng, n, maxlen, alphabet=b"ACGT"):
    return [bytes(rng.choice(list(alphabet), int(rng.integers(0, maxlen))).astype(np.uint8)) for _ in range(n)]


def _write_corpus(rng, d):
    """files of every type whose shapes hit the readers' edges: lines shorter than k, empty lines,
    comments in odd places, CRLF, no final newline, text beyond the 64 KiB buffer"""
    os.makedirs(d, exist_ok=True)
    # FASTA / multi-FASTA
    for i in range(10):
        lines = []
        for _ in range(int(rng.integers(1, 5))):
            lines.append(b">" + bytes(rng.choice(list(b"abcdefghijklmnopqrstuvwxyz0123456789 "), int(rng.integers(0, 40))).astype(np.uint8)))
            for ln in _random_lines(rng, int(rng.integers(0, 7)), 70, b"ACGTN"):
                lines.append(ln)
                if rng.random() < 0.15:
                    lines.append(b";" + ln[:5])
                if rng.random() < 0.1:
                    lines.append(b"")
        nl = b"\r\n" if i == 3 else b"\n"
        body = nl.join(lines) + (b"" if i % 4 == 1 else nl)
        with open(os.path.join(d, "f%02d.fasta" % i), "wb") as f:
            f.write(body)
        with open(os.path.join(d, "m%02d.mfasta" % i), "wb") as f:
            f.write(body)
        with gzip.open(os.path.join(d, "g%02d.fa.gz" % i), "wb") as f:
            f.write(body)
    # FASTQ
    for i in range(4):
        recs = []
        for r in range(int(rng.integers(1, 30))):
            read = _random_lines(rng, 1, 120, b"ACGTN")[0]
            recs += [b"@read%d" % r, read, b"+", b"I" * len(read)]
        body = b"\n".join(recs) + (b"" if i == 2 else b"\n")
        opener = gzip.open if i == 1 else open
        with opener(os.path.join(d, "q%02d.fastq%s" % (i, ".gz" if i == 1 else "")), "wb") as f:
            f.write(body)
    # text: small, around the buffer size, several refills
    for i, n in enumerate((0, 7, 30, 31, 500, 65535, 65536, 65537, 65536 + 29, 65536 + 30, 131072 - 30, 200001)):
        with open(os.path.join(d, "t%02d.txt" % i), "wb") as f:
            f.write(bytes(rng.integers(32, 127, size=n).astype(np.uint8)).replace(b"~", b"\n"))
    # .cobs_doc
    for i in range(3):
        kmers = [bytes(rng.choice(list(b"ACGT"), 31).astype(np.uint8)) for _ in range(int(rng.integers(0, 40)))]
        D.write_kmer_buffer(os.path.join(d, "k%02d.cobs_doc" % i), "document_%06d" % i, kmers)


def test_native_readers_equal_the_checker_on_generated_files(capi, tmp_path):
    rng = np.random.default_rng(int(os.environ.get("COBS_FUZZ_SEED", "20240928")))     # soak: other seeds
    root = str(tmp_path / "corpus")
    _write_corpus(rng, root)
    ents, dl = D.document_list(root), capi.DocumentList(root)
    assert dl.size() == len(ents) > 40
    for e, g in zip(ents, dl):
        _same_entry(e, g)
        ks = (31,) if e.type == D.KMER_BUFFER else (31, 9, 2, 64)
        for k in ks:
            assert e.num_terms(k) == g.num_terms(k), (e.path, k)
            assert e.terms(k) == g.terms(k), (e.path, k)


def test_list_files_filters_and_sorting(capi, docdir, tmp_path):
    """.list files (document_list.hpp:360-381), type filters (:165-196), sort_by_size (:424-430),
    DocumentList::add, StringToFileType"""
    lst = tmp_path / "docs.list"
    lst.write_text("# a comment\n\n%s\nrel.txt\n" % os.path.join(docdir, "fastq", "sample3.fastq"))
    (tmp_path / "rel.txt").write_text("some text that is long enough for a 31-gram, yes\n")
    dl = capi.DocumentList(str(lst))
    assert [d.name for d in dl] == ["sample3", "rel"] or [d.name for d in dl] == ["rel", "sample3"]
    assert [d.path for d in dl] == sorted(d.path for d in dl)
    want = D.document_list(str(lst))
    assert [(d.path, d.name) for d in dl] == [(e.path, e.name) for e in want]
    # filters: by enum, by the CLI's strings
    root = os.path.dirname(docdir)
    everything = capi.DocumentList(root)
    assert everything.size() == len(D.document_list(root))
    for ft, name in ((capi.FileType.Fastq, "fastq"), (capi.FileType.Cortex, "cortex"), (capi.FileType.Text, "text"),
                     (capi.FileType.Fasta, "fasta")):
        a, b = capi.DocumentList(root, ft), capi.DocumentList(root, name)
        assert a.size() == b.size() == len(D.document_list(root, int(ft))) > 0
        assert all(d.type == ft for d in a)
    with pytest.raises(capi.CobsGpuError):
        capi.DocumentList(root, "no-such-type")
    # sort_by_size / sort_by_path
    everything.sort_by_size()
    keys = [(d.size, d.path) for d in everything]
    assert keys == sorted(keys)
    everything.sort_by_path()
    assert [d.path for d in everything] == sorted(d.path for d in everything)
    # add: one file at a time, in call order; unknown types and broken files are errors
    dl = capi.DocumentList()
    dl.add(os.path.join(docdir, "text", "sample2.txt"))
    dl.add(os.path.join(docdir, "fasta_multi", "sample2.mfasta"))
    assert dl.size() == 6 and dl[0].name == "sample2" and dl[5].subdoc_index == 4
    with pytest.raises(capi.CobsGpuError):
        dl.add(str(tmp_path / "docs.list"))
    bad = tmp_path / "bad.ctx"
    bad.write_bytes(b"CORTEY" + b"\0" * 100)
    with pytest.raises(capi.CobsGpuError) as e:
        dl.add(str(bad))
    assert "magic number not found" in str(e.value)
    nofq = tmp_path / "bad.fastq"
    nofq.write_bytes(b"@r\nACGT\nX\nIIII\n")
    with pytest.raises(capi.CobsGpuError) as e:
        dl.add(str(nofq))
    assert "does not start with +" in str(e.value)
    # a directory scan skips what it cannot load and goes on (:393-402)
    (tmp_path / "ok.txt").write_text("x" * 40)
    assert {d.name for d in capi.DocumentList(str(tmp_path))} == {"rel", "ok"}
    # in-memory documents
    dl = capi.DocumentList()
    dl.add_document("mem", [b"ACGTACGTAC", b"", b"GGGTTTAAACCC"])
    assert dl[0].type == capi.FileType.Memory and dl[0].num_terms(4) == 7 + 9 and len(dl[0].terms(4)) == 16
    assert dl[0].terms(11) == [b"GGGTTTAAACC", b"GGTTTAAACCC"]


def test_readers_survive_damaged_files(capi, docdir, tmp_path):
    """truncated and bit-flipped copies of every fixture: the readers answer with an error or with
    terms, never with a crash or an out-of-bounds read (the checker is not consulted: damaged files
    have no reference behaviour worth pinning)"""
    rng = np.random.default_rng(99)
    sources = []
    for sub in ("cortex", "fastq", "fasta_multi", "text"):
        for fn in sorted(os.listdir(os.path.join(docdir, sub))):
            if fn == "document_sorted.txt" or fn.endswith("-k15.txt") or fn.endswith("-k19.txt"):
                continue
            sources.append(os.path.join(docdir, sub, fn))
    kdoc = str(tmp_path / "ok.cobs_doc")
    D.write_kmer_buffer(kdoc, "doc", [b"ACGT" * 7 + b"ACG"] * 9)
    sources.append(kdoc)
    outcomes = {"ok": 0, "error": 0}
    for src in sources:
        raw = open(src, "rb").read()
        ext = src[src.index(".", src.rfind(os.sep)):]
        for trial in range(40):
            b = bytearray(raw)
            mode = trial % 4
            if mode == 0:
                b = b[:int(rng.integers(0, len(b) + 1))]
            elif mode == 1:
                for _ in range(int(rng.integers(1, 9))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif mode == 2:
                at = int(rng.integers(0, min(len(b), 64)))
                b[at:at + 4] = (0xFFFFFFFF if trial % 8 == 2 else int(rng.integers(0, 2 ** 31))).to_bytes(4, "little")
            else:
                b = b[:int(rng.integers(0, 80))] + bytes(rng.integers(0, 256, size=int(rng.integers(0, 50))).astype(np.uint8))
            p = str(tmp_path / ("damaged" + ext))
            with open(p, "wb") as f:
                f.write(bytes(b))
            dl = capi.DocumentList()
            try:
                dl.add(p)
            except capi.CobsGpuError:
                outcomes["error"] += 1
                continue
            for e in dl:
                for k in (31, 5):
                    try:
                        n = len(e.terms(k))
                        assert n >= 0 and e.num_terms(k) >= 0
                    except capi.CobsGpuError:
                        pass
            outcomes["ok"] += 1
    assert outcomes["ok"] > 50 and outcomes["error"] > 50, outcomes


def test_cpp_construction_mirror(docdir, tmp_path):
    """include/cobs_gpu_construct.hpp (cobs_gpu::DocumentList / DocumentEntry / FileType, the C++17
    mirror of cobs/document_list.hpp): compiled with g++ against libcobs_gpu.so, host-only calls,
    compared with the checker's view of the same files"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "construct_api")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "construct_api.cpp"), "-o", exe,
                           "-L", os.path.join(root, "cobs_amd"), "-lcobs_gpu",
                           "-Wl,-rpath," + os.path.join(root, "cobs_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, docdir], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().split("\n")
    ents = D.document_list(os.path.join(docdir, "fasta_multi"))
    assert lines[0] == "fasta_multi %d" % len(ents)
    for ln, e in zip(lines[1:], ents):
        assert ln == "%s %d %d %d %d" % (e.name, e.size, e.subdoc_index, e.num_terms(31), len(e.terms(31)))
    srt = _lines(os.path.join(docdir, "cortex", "document_sorted.txt"))
    assert lines[1 + len(ents)] == "cortex DRR030535 31 24158 24158 %s %s" % (srt[0].decode(), srt[-1].decode())
    fq = sorted(D.document_list(docdir, D.FASTQ), key=lambda e: (e.size, e.path))
    assert lines[2 + len(ents):2 + len(ents) + 3] == ["fastq %s %d" % (e.name, e.size) for e in fq]
    assert "Unknown file type nonsense" in lines[-2]
    assert lines[-1] == "refused Error: COBS index file must end with .cobs_classic"


def test_doc_list_and_doc_dump_sub_tools(oracle, golden_dir):
    """`cobs doc-list` / `cobs doc-dump` (reference src/cobs.cpp:75-161; host-only views of a document list): the list
    lines of print_document_list, and every document's terms -- canonical k-mers as canonicalize_kmer writes them
    (cobs/util/query.cpp:143-199), "Invalid DNA base pair: ..." for a term with a character that is not A/C/G/T --
    against the checker's readers and its canonicalize_kmer"""
    import subprocess
    from oracle import documents as OD
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cobs_amd", "cobs_gpu_query")
    assert os.path.exists(tool), "build cobs_amd/cobs_gpu_query first (make -C cobs_amd/csrc)"

    def _run(*args):
        return subprocess.run([tool] + list(args), capture_output=True, text=True, timeout=300)
    fasta = os.path.join(golden_dir, "fasta")
    r = _run("doc-list", fasta, "-k", "31")
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0] == "--- document list (7 entries) ---" and out[-5].startswith("documents: 7")
    entries = OD.document_list(fasta)
    assert [ln.split(" : ")[-1] for ln in out[1:8]] == [e.name for e in entries]
    assert [int(ln.split(" 31-mers ")[1].split(" : ")[0]) for ln in out[1:8]] == [e.num_terms(31) for e in entries]
    for k, flags in ((31, []), (15, []), (20, ["--no-canonicalize"])):
        r = _run("doc-dump", fasta, "-k", str(k), *flags)
        assert r.returncode == 0, r.stderr
        want = []
        for e in entries:
            for t in e.terms(k):
                if flags:
                    want.append(t.decode("latin-1"))
                    continue
                canon, good = oracle.canonicalize_kmer(bytes(t))
                want.append(canon.decode("latin-1") if good else "Invalid DNA base pair: " + t.decode("latin-1"))
        assert r.stdout.splitlines() == want, k
        assert "Found 7 documents." in r.stderr and ("document[6] : %d terms." % entries[6].num_terms(k)) in r.stderr
    # a text document with characters outside ACGT: those terms are reported, not printed
    text = os.path.join(golden_dir, "documents", "text")
    r = _run("doc-dump", text, "-k", "7")
    assert r.returncode == 0 and "Invalid DNA base pair: " in r.stdout


def test_print_parameters_and_print_kmers_sub_tools(oracle):
    """`cobs print-parameters` / `cobs print-kmers` (reference src/cobs.cpp:532-599, host only): the sizing formula of
    cobs/util/calc_signature_size.cpp:17-34 against the checker's, the canonical k-mers against the checker's
    canonicalize_kmer -- including the reference's loop bound, which leaves out a query's last k-mer"""
    import subprocess
    from oracle import construct as OC
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cobs_amd", "cobs_gpu_query")
    assert os.path.exists(tool), "build cobs_amd/cobs_gpu_query first (make -C cobs_amd/csrc)"

    def _run(*args):
        return subprocess.run([tool] + list(args), capture_output=True, text=True, timeout=60)
    r = _run("print-parameters")
    assert r.returncode == 0 and r.stdout == "%g\n" % (-1 / np.log(1 - 0.3))           # ostream's default: 6 digits
    for h, f, n_text, n in ((1, 0.3, "1000", 1000), (3, 0.1, "4Mi", 4 << 20), (2, 0.01, "5M", 5 * 10 ** 6), (7, 0.5, "123456789", 123456789)):
        r = _run("print-parameters", "-h", str(h), "-f", str(f), "-n", n_text)
        assert r.returncode == 0, r.stderr
        want = OC.calc_signature_size(n, h, f)
        lines = r.stdout.splitlines()
        assert lines[0] == "signature_size = %d" % want
        assert lines[1].startswith("signature_bytes = %d = " % (want // 8))
        num, unit = lines[1].split(" = ")[2].split(" ") if lines[1].count(" ") == 5 else (lines[1].split(" = ")[2].strip(), "")
        scale = {"": 0, "Ki": 1, "Mi": 2, "Gi": 3}[unit]
        assert abs(float(num) * 1024 ** scale - want // 8) <= 0.0005 * 1024 ** scale and float(num) < 1024
    assert _run("print-parameters", "-f", "1.5").returncode != 0                          # no such filter
    assert _run("print-parameters", "-n", "lots").returncode != 0
    rng = np.random.default_rng(5)
    for k in (31, 4, 15):
        q = "".join(rng.choice(list("ACGT"), size=80)) + "N" + "".join(rng.choice(list("ACGT"), size=40))
        r = _run("print-kmers", q, "-k", str(k))
        assert r.returncode == 0, r.stderr
        want = []
        for i in range(len(q) - k):                                                       # (not len - k + 1: as the reference)
            t = q[i:i + k].encode()
            canon, good = oracle.canonicalize_kmer(t)
            want.append(canon.decode() if good else "Invalid DNA base pair: " + t.decode())
        assert r.stdout.splitlines() == want, k
    assert _run("print-kmers", "ACGT", "-k", "31").stdout == ""
    assert _run("print-kmers").returncode != 0
