"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/cobs_gpu.h declares, parses index headers without a device and
fails loudly (no CPU fallback) when a compute entry point is used without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(headers=("cobs_gpu.h", "cobs_gpu_batch.h", "cobs_gpu_diag.h", "cobs_gpu_construct.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(cobs_gpu_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from cobs_amd import _capi
    lib = _capi.load()
    names = _declared_symbols()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), "libcobs_gpu.so does not export " + n
        assert n in _capi.SYMBOLS, "cobs_amd/_capi.py does not bind " + n
    assert sorted(_capi.SYMBOLS) == names
    assert lib.cobs_gpu_abi_version() == 2
    # the query path's headers stand alone: construction and document lists live in cobs_gpu_construct.h
    query_only = _declared_symbols(("cobs_gpu.h", "cobs_gpu_batch.h", "cobs_gpu_diag.h"))
    assert not [n for n in query_only if "doclist" in n or "_build_" in n or "combine" in n or "construct" in n]
    assert len(query_only) < len(names)
    # the drop-in boundary itself is thin: open / geometry / search / timers (+ the device list); batches, the RCCL
    # exchange and the procedural index are cobs_gpu_batch.h, planners and diagnostics cobs_gpu_diag.h
    boundary = _declared_symbols(("cobs_gpu.h",))
    assert len(boundary) <= 20 and not [n for n in boundary if "_batch_" in n or "_comm_" in n or "_plan" in n or "read_row" in n]


def test_struct_layouts_match_header():
    from cobs_amd import _capi
    assert C.sizeof(_capi.Options) == 32
    assert C.sizeof(_capi.Hit) == 12
    assert C.sizeof(_capi.IndexInfo) == 16 + 7 * 8 + 8 + 3 * 8
    assert C.sizeof(_capi.Synth) == 16 + 4 * 8 + 8


def test_no_oracle_in_product():
    """the product path must not import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cobs_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle" not in txt.lower(), (fn, "mentions the oracle")
    out = os.popen("ldd %s" % os.path.join(ROOT, "cobs_amd", "libcobs_gpu.so")).read()
    assert "oracle" not in out


def test_errors_without_a_device(golden_dir, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU-only hosts")
    import cobs_amd
    from cobs_amd import _capi
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search(os.path.join(golden_dir, "c1.cobs_classic"))
    assert e.value.status == _capi.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search.synthetic("compact", [100, 200], 100, page_size=16)
    assert e.value.status == _capi.ERR_NO_DEVICE


def test_open_errors_are_reported_before_any_device_work(golden_dir, tmp_path):
    import cobs_amd
    from cobs_amd import _capi
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search(str(tmp_path / "missing.cobs_classic"))
    assert e.value.status == _capi.ERR_OPEN
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search(os.path.join(golden_dir, "expected.json"))
    assert e.value.status == _capi.ERR_FORMAT
    # truncated matrix
    raw = open(os.path.join(golden_dir, "c1.cobs_classic"), "rb").read()
    p = tmp_path / "short.cobs_classic"
    p.write_bytes(raw[:4000])
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search(str(p))
    assert e.value.status == _capi.ERR_FORMAT
    # wrong version
    p2 = tmp_path / "v2.cobs_classic"
    p2.write_bytes(raw[:18] + b"\x02\x00\x00\x00" + raw[22:])
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search(str(p2))
    assert e.value.status == _capi.ERR_FORMAT
    with pytest.raises(cobs_amd.CobsGpuError) as e:
        cobs_amd.Search(os.path.join(golden_dir, "c1.cobs_classic"), shard_rank=3, shard_count=2)
    assert e.value.status in (_capi.ERR_ARG, _capi.ERR_NO_DEVICE)


def test_python_mirror_surface():
    """names and defaults of python/module.cpp:351-386"""
    import inspect

    import cobs_amd
    r = cobs_amd.SearchResult()
    assert r.doc_name == "" and r.score == 0
    sig = inspect.signature(cobs_amd.Search.search)
    assert list(sig.parameters) == ["self", "query", "threshold", "num_results"]
    assert sig.parameters["threshold"].default == 0.0 and sig.parameters["num_results"].default == 0
    assert isinstance(cobs_amd.__version__, str)
    # every name the reference module exports (python/module.cpp:100-386)
    for name in ("disable_cache", "DocumentList", "ClassicIndexParameters", "classic_construct",
                 "classic_construct_list", "CompactIndexParameters", "compact_construct",
                 "compact_construct_list", "SearchResult", "Search"):
        assert hasattr(cobs_amd, name), name
    for fn, first in ((cobs_amd.classic_construct, "input"), (cobs_amd.compact_construct, "input"),
                      (cobs_amd.classic_construct_list, "list"), (cobs_amd.compact_construct_list, "list")):
        ps = list(inspect.signature(fn).parameters)
        assert ps[:3] == [first, "out_file", "index_params"], ps
    import cobs_index                       # the reference's module name, same objects
    assert cobs_index.Search is cobs_amd.Search and cobs_index.SearchResult is cobs_amd.SearchResult
    p = cobs_amd.ClassicIndexParameters()
    assert (p.term_size, p.canonicalize, p.num_hashes, p.false_positive_rate) == (31, True, 1, 0.3)
    assert cobs_amd.CompactIndexParameters().page_size == 0


def test_bench_query_stream_is_the_reference_benchmark_stream(oracle):
    """bench.make_queries == std::mt19937(42) % 4 (reference src/cobs.cpp:709-720)"""
    import bench
    qs = bench.make_queries(3, 1000, seed=42)
    g = oracle.Mt19937Queries(42)
    assert [g.next(1030) for _ in range(3)] == qs
    cfg = bench.c3_config()
    assert cfg["signature_sizes"][0] == 250000 and cfg["signature_sizes"][-1] == 4000000
    assert len(cfg["signature_sizes"]) == 8 and cfg["page_size"] * 8 * 8 >= cfg["num_docs"]


def test_header_is_plain_c_and_cpp(tmp_path):
    """include/cobs_gpu.h is the FFI surface: it must compile as C99 (cgo / ctypes-style
    consumers) and as C++17, and the C++ mirror must compile against it"""
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    c = tmp_path / "t.c"
    c.write_text('#include "cobs_gpu.h"\nint main(void) { cobs_gpu_options o; o.struct_size = sizeof o; '
                 'return (int)o.struct_size == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc,
                           "-fsyntax-only", str(c)])
    for hdr, probe in (("cobs_gpu_batch.h", "cobs_gpu_synth d; d.num_docs = 1; return (int)d.num_docs == 0;"),
                       ("cobs_gpu_diag.h", "cobs_gpu_xfer x; x.peer = 1; return (int)x.peer == 0;")):
        cx = tmp_path / ("t_" + hdr.replace(".h", ".c"))
        cx.write_text('#include "%s"\nint main(void) { %s }\n' % (hdr, probe))
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(cx)])
    c2 = tmp_path / "t2.c"
    c2.write_text('#include "cobs_gpu_construct.h"\nint main(void) { cobs_gpu_build_params p; p.struct_size = sizeof p; '
                  'return (int)p.struct_size == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc,
                           "-fsyntax-only", str(c2)])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "cobs_gpu_search.hpp"\nint main() { cobs_gpu::SearchResult r; return r.score != 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(cpp)])


def test_absurd_procedural_geometry_is_refused():
    """sizes whose byte counts would wrap 64-bit arithmetic are an error before any allocation"""
    import cobs_amd
    from cobs_amd import _capi
    for kind, sigs, docs, ps in (("compact", [1 << 46] * 4, 4 * 8 * (1 << 27), 1 << 27),
                                 ("classic", [1 << 46], 4000000000, 0),
                                 ("compact", [1 << 40] * 2000, 2000 * 8 * 4096, 4096)):
        with pytest.raises(cobs_amd.CobsGpuError) as e:
            cobs_amd.Search.synthetic(kind, sigs, docs, page_size=ps)
        assert e.value.status in (_capi.ERR_UNSUPPORTED, _capi.ERR_ARG, _capi.ERR_NO_DEVICE), (kind, e.value)


def test_mock_rccl_covers_every_rccl_symbol_the_library_imports():
    """tests/mock_rccl (the stand-in communicator the N > 1 tests preload under the shipped library) must define every
    nccl* symbol libcobs_gpu.so imports: a new RCCL call in comm.cpp that the stand-in lacks would silently go to
    the real librccl in those tests"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tests", "mock_rccl", "build.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr

    def syms(path, flag):
        out = subprocess.run(["nm", "-D", flag, path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("nccl")}
    wanted = syms(os.path.join(root, "cobs_amd", "libcobs_gpu.so"), "--undefined-only")
    have = syms(os.path.join(root, "cobs_amd", "libmockrccl.so"), "--defined-only")
    assert wanted and wanted <= have, sorted(wanted - have)
