"""scripts/open_time.py -- how long cobs_gpu_open takes to make an index FILE resident in HBM
(the reference's counterpart: mmap + page faults, or --load-complete's read loop,
cobs/util/query.cpp:38-88).  Writes the C3 index (18.4 GB) with the generator, opens it twice
(page cache warm), prints seconds and GB/s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench  # noqa: E402
import cobs_amd  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/c3_open_time.cobs_compact"
cfg = bench.c3_config()
t0 = time.time()
cobs_amd.write_synthetic(path, cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
size = os.path.getsize(path)
print("generator: %.1f GB in %.1f s" % (size / 1e9, time.time() - t0))
for rep in range(3):
    t0 = time.time()
    s = cobs_amd.Search(path)
    dt = time.time() - t0
    print("open #%d: %.3f s = %.1f GB/s (hbm bytes %.1f GB)" % (rep, dt, size / dt / 1e9, s.info(0).hbm_bytes / 1e9))
    q = bench.make_queries(4, 100)
    t0 = time.time()
    s.search_batch(q, 0.8)
    print("   first search %.3f s" % (time.time() - t0))
    s.close()
os.remove(path)
