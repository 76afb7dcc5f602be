#!/usr/bin/env python3
"""scripts/ab.py -- interleaved A/B timing of scan-kernel tuning hooks inside one
process (boxes drift by several percent with temperature; only interleaved runs compare)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c3"
    configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[2:]] or [{}]
    if shape == "c3":
        cfg, nq, kmers = bench.c3_config(), 10000, 1000
    elif shape == "c3short":
        cfg, nq, kmers = bench.c3_config(), 20000, 100
    elif shape.startswith("reads"):
        bp = int(shape[5:])
        cfg, nq, kmers = bench.c3_config(), 40000, bp - 30
    elif shape in ("c3h2", "c3h3"):     # several hash functions: the AND (aggregate_rows) variant of the scan
        cfg, nq, kmers = dict(bench.c3_config(), num_hashes=int(shape[-1])), 4000, 1000
    elif shape == "c2":
        cfg, nq, kmers = bench.c2_config(), 10000, 1000
    elif shape == "c4":
        r = (1600000 / 100000) ** (1.0 / 244)
        cfg = {"kind": "compact", "num_docs": 1000000, "page_size": 512,
               "signature_sizes": [int(100000 * r ** i) for i in range(245)]}
        nq, kmers = 1000, 1000
    elif shape == "c4big":      # C4 geometry with a 10x larger batch (more lookups per cached line)
        r = (1600000 / 100000) ** (1.0 / 244)
        cfg = {"kind": "compact", "num_docs": 1000000, "page_size": 512,
               "signature_sizes": [int(100000 * r ** i) for i in range(245)]}
        nq, kmers = 6000, 1000
    elif shape in ("big4", "small4"):     # only large (uncacheable tile columns) / only small sub-indexes of the C3 kind
        rows = 4000000 if shape == "big4" else 300000
        cfg = {"kind": "compact", "num_docs": 4 * 12544, "page_size": 1568, "signature_sizes": [rows] * 4}
        nq, kmers = 10000, 1000
    elif shape == "ps128":
        r = 16 ** (1.0 / 97)
        cfg = {"kind": "compact", "num_docs": 100000, "page_size": 128,
               "signature_sizes": [int(250000 * r ** i) for i in range(98)]}
        nq, kmers = 4000, 1000
    else:
        raise SystemExit("unknown shape")
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1,
                                  num_hashes=cfg.get("num_hashes", 1))
    b = cobs_amd.Batch(s)
    b.set_queries(bench.make_queries(nq, kmers))
    # a config is a comma-separated list of per-handle tuning keys: waves=2,tile_w=16,mq=1
    # (the old COBS_GPU_* spellings are accepted too)
    alias = {"COBS_GPU_WAVES": "waves", "COBS_GPU_TILE_W": "tile_w", "COBS_GPU_MQ": "mq"}
    configs = [{alias.get(k, k): v for k, v in c.items()} for c in configs]
    keys = sorted({k for c in configs for k in c})
    times = [[] for _ in configs]
    for rnd in range(7):
        for ci, c in enumerate(configs):
            for k in keys:
                s.set_tuning(k, -1 if k in ("mq", "graph") else 0)
            for k, v in c.items():
                s.set_tuning(k, int(v))
            for _ in range(3):
                b.run(0.0)
            b.sync()
            ms = b.kernel_ms()["scan_ms"]
            if rnd > 0:
                times[ci].append(ms)
    st = b.stats()
    for c, t in zip(configs, times):
        med = statistics.median(t)
        print("%-50s scan median %.3f ms  min %.3f  max %.3f   %.1f GB/s" %
              (c, med, min(t), max(t), st["algorithmic_bytes"] / med / 1e6))


if __name__ == "__main__":
    main()
