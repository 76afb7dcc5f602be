"""scripts/compact_bench.py [DOCS] [DOC_MB] -- compact index construction with few / many sub-indexes
(one build_into per sub-index inside one BuildContext): time to a file and to a resident handle."""
import faulthandler
import os
import shutil
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: F401,E402
import cobs_amd  # noqa: E402
import construct_bench as cb  # noqa: E402

faulthandler.dump_traceback_later(90, exit=True)
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
doc_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
d = "/tmp/cobs_compact_bench"
cb.write_docs(os.path.join(d, "docs"), ndocs, int(doc_mb * 1e6))
dl = cobs_amd.DocumentList(os.path.join(d, "docs"))
print("listed", dl.size(), flush=True)
for ps in (64, 8, 2):
    p = cobs_amd.CompactIndexParameters()
    p.page_size = ps
    p.clobber = True
    out = os.path.join(d, "x.cobs_compact")
    for rep in range(2):
        t0 = time.time()
        cobs_amd.compact_construct(list=dl, out_file=out, index_params=p)
        dt = time.time() - t0
        print("  page_size %d rep %d: %.3f s" % (ps, rep, dt), flush=True)
    t0 = time.time()
    s = cobs_amd.build_search(list=dl, index_params=p, kind="compact")
    dr = time.time() - t0
    # the file (rows through the pinned buffers, padding squeezed out on the host) holds the matrix of the handle
    import numpy as np
    sf = cobs_amd.Search(out)
    for e in (dl[0], dl[dl.size() // 2]):
        q = b"".join(open(e.path, "rb").read().split(b"\n")[1:])[1000:1400]
        a, b = s.counts(q), sf.counts(q)
        assert np.array_equal(a, b) and int(a.max()) == len(q) - 30, (ps, e.name)
    sf.close()
    s.close()
    print("page_size %d (%d docs per sub-index, %d sub-indexes): file %.3f s, resident %.3f s, %d MB" % (
        ps, 8 * ps, (ndocs + 8 * ps - 1) // (8 * ps), dt, dr, os.path.getsize(out) >> 20), flush=True)
shutil.rmtree(d)
