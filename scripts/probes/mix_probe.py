"""scripts/probes/mix_probe.py -- do the small sub-indexes (tile columns the Infinity Cache holds: ~7.4 TB/s) and the large
ones (HBM-bound: ~6.3 TB/s) run faster when they are scanned AT THE SAME TIME than one after the other?  Two shards of
the C3 index (pages 0..4 / 4..7) as two handles on one GPU, the headline batch on each: both scans on one stream, then on
two streams (the work-groups of both launches share the CUs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
import cobs_amd

queries = B.make_queries(10000, 1000)
cfg = B.c3_config(1.0); cfg["num_hashes"] = 1
for n in (2, 4):
    hs = [B.make_index(cfg, 0, r, n) for r in range(n)]
    bs = []
    for h in hs:
        b = cobs_amd.Batch(h)
        b.set_queries(queries)
        bs.append(b)
    streams = [torch.cuda.Stream() for _ in range(n)]
    def one_stream():
        for b in bs:
            b.run(0.0, streams[0].cuda_stream)
    def many_streams(order):
        for i in order:
            bs[i].run(0.0, streams[i].cuda_stream)
    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3
    t_seq = timed(one_stream)
    print("%d shards: one after the other %.3f ms (incl. hashing of each)" % (n, t_seq), flush=True)
    for order in (list(range(n)), list(reversed(range(n))), [0, n - 1] + list(range(1, n - 1))):
        print("%d shards: on %d streams, launch order %s: %.3f ms" % (n, n, order, timed(lambda: many_streams(order))), flush=True)
    for b in bs:
        b.sync()
    del bs, hs
