"""scripts/probes/tile_width_probe.py -- the scan of the headline batch at tile widths 8 / 16 / 32 / 64 chunks (128 B ... 1 KB
of a row per work-group) where the Infinity Cache cannot hold a tile column: the largest sub-index of C3 alone
(4 M rows: 512 MB per 128-byte column), the 3 largest, and the C3 geometry at 8x the rows (147 GB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
import cobs_amd

queries = B.make_queries(10000, 1000)

def sweep(name, s, nq=10000):
    b = cobs_amd.Batch(s)
    b.set_queries(queries[:nq])
    for w in (0, 8, 16, 32, 64, 0, 8, 16):
        s.set_tuning("tile_w", w)
        for _ in range(2):
            b.run(0.0, 0)
        b.sync(); b.kernel_ms()
        for _ in range(5):
            b.run(0.0, 0)
        b.sync()
        ms = b.kernel_ms()["scan_ms"]
        algo = b.stats()["algorithmic_bytes"]
        print("%-34s tile_w %2d  scan %8.3f ms  %7.1f GB/s algorithmic  %.3f of 8 TB/s" % (name, w, ms, algo / ms / 1e6, algo / ms / 1e6 / 8000), flush=True)
    del b

cfg = B.c3_config(1.0); cfg["num_hashes"] = 1
sweep("C3 shard 7/8 (one sub-index, 4 M rows)", B.make_index(cfg, 0, 7, 8))
sweep("C3 shard 3/4 (two largest)", B.make_index(cfg, 0, 3, 4))
sweep("C3 whole", B.make_index(cfg, 0))
cfg8 = B.c3_config(8.0); cfg8["num_hashes"] = 1
sweep("C3 x 8 rows (147 GB)", B.make_index(cfg8, 0))
