"""`cobs compact-construct-combine` (compact_combine_into_compact, reference
cobs/construction/compact_index.cpp:51-169): classic index files -> one compact index.  Host-only
file work in libcobs_gpu.so (cobs_gpu_combine_compact), checked on the CPU against the checker's
compact_construct of the same documents."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_classic_parts_become_the_compact_index(oracle, construct, tmp_path):
    import cobs_amd
    q = oracle.random_sequence(3000, 5)
    docs = construct.generate_documents_all(q, 40, num_hashes=2)
    ps = 2                                           # 16 documents per sub-index: 16 + 16 + 8
    ordered = sorted(docs, key=lambda x: (x.size, x.path))
    d = tmp_path / "parts"
    d.mkdir()
    parts = []
    for g in range(0, 40, 16):
        part = sorted(ordered[g:g + 16], key=lambda x: x.path)
        p = str(d / ("%02d.cobs_classic" % (g // 16)))
        sig = construct.calc_signature_size(max(x.num_terms for x in part), 2, 0.1)
        construct.classic_construct(part, p, num_hashes=2, false_positive_rate=0.1, signature_size=sig)
        parts.append(p)
    want, got = str(tmp_path / "w.cobs_compact"), str(tmp_path / "g.cobs_compact")
    construct.compact_construct(docs, want, num_hashes=2, false_positive_rate=0.1, page_size=ps)
    cobs_amd.compact_combine(parts, got, ps)
    assert open(got, "rb").read() == open(want, "rb").read()
    # the narrow index must come last; a page size the rows do not fill is refused (:85-90)
    with pytest.raises(cobs_amd.CobsGpuError):
        cobs_amd.compact_combine(parts[::-1], got, ps)
    with pytest.raises(cobs_amd.CobsGpuError):
        cobs_amd.compact_combine(parts, got, ps + 1)
    with pytest.raises(cobs_amd.CobsGpuError):
        cobs_amd.compact_combine([want], got, ps)    # not a classic index
    # the sub-tool: classic indexes of a directory in path order, page size from -p
    tool = os.path.join(ROOT, "cobs_amd", "cobs_gpu_query")
    out = str(tmp_path / "cli.cobs_compact")
    r = subprocess.run([tool, "compact-construct-combine", str(d), out, "-p", str(ps)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(want, "rb").read()
    # and the result answers like its parts
    ix = oracle.Index.open(out)
    assert (ix.num_pages, ix.page_size, ix.num_docs) == (3, ps, 40)
