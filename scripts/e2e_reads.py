"""scripts/e2e_reads.py -- read classification end to end through the host-buffer API
(cobs_gpu_search_batch via Search.search_packed, threshold 0.8: hits only): queries start in host
memory, results end in host memory; PCIe, hashing, the pipelined passes and the host-side result
handling are all inside the time.  C3 index (100 k documents), random reads."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, bench, cobs_amd
cfg = bench.c3_config()
s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
for n, bp in ((200000, 100), (1000000, 100), (1000000, 50), (400000, 150)):
    rng = np.random.default_rng(1)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * bp, dtype=np.uint8)]
    offs = np.arange(n + 1, dtype=np.uint64) * bp
    s.search_packed(text[:bp * 1000], offs[:1001], 0.8)      # warm
    for rep in range(2):
        t0 = time.time()
        o, h = s.search_packed(text, offs, 0.8)
        dt = time.time() - t0
    tm = s.timers(reset=True)
    print("%d reads x %d bp: %.3f s = %.2f M reads/s, %.2f GB/s of query text; timers %s" % (n, bp, dt, n / dt / 1e6, n * bp / dt / 1e9, {k: round(v, 3) for k, v in tm.items()}))
