#!/usr/bin/env python3
"""scripts/stream_bench.py -- out-of-core (BASELINE config 5 style) measurement on one GPU:
a compact index file on local disk / page cache, opened with an HBM budget smaller than
the file, so every batch pass streams all chunks over PCIe."""
import os
import struct
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def write_compact(path, src, cfg):
    """dump the resident synthetic index `src` as a .cobs_compact file (reference layout)"""
    names = ["file_%06u" % i for i in range(cfg["num_docs"])]
    ps = cfg["page_size"]
    hdr = b"COBS:" + b"COMPACT_INDEX" + struct.pack("<I", 1)
    hdr += struct.pack("<IBIIQ", 31, 1, len(cfg["signature_sizes"]), len(names), ps)
    for s in cfg["signature_sizes"]:
        hdr += struct.pack("<QQ", s, 1)
    hdr += ("\n".join(names) + "\n").encode()
    pad = (ps - ((len(hdr) + 13) % ps)) % ps
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * pad + b"COMPACT_INDEX")
        for p, s in enumerate(cfg["signature_sizes"]):
            step = 200000
            for r in range(0, s, step):
                f.write(src.read_rows(0, p, r, min(step, s - r)).tobytes())


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    budget_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    cfg = bench.c3_config(scale)
    path = "/tmp/stream_bench.cobs_compact"
    src = cobs_amd.Search.synthetic("compact", cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    t0 = time.perf_counter()
    write_compact(path, src, cfg)
    size = os.path.getsize(path)
    print("wrote %.2f GB in %.1f s" % (size / 1e9, time.perf_counter() - t0), flush=True)
    qs = bench.make_queries(nq, 1000)
    ref = cobs_amd.Batch(src)
    ref.set_queries(qs[:8])
    ref.run(0.0)
    ref.sync()
    want = [ref.counts_host(i) for i in range(8)]
    del ref, src
    for label, budget in (("resident", 0), ("streamed", int(budget_gb * 2 ** 30))):
        t0 = time.perf_counter()
        s = cobs_amd.Search(path, hbm_budget=budget)
        t_open = time.perf_counter() - t0
        b = cobs_amd.Batch(s)
        b.set_queries(qs)
        b.run(0.0)
        b.sync()
        ok = all(np.array_equal(b.counts_host(i), want[i]) for i in range(8))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 3
        for _ in range(steps):
            b.run(0.0)
        b.sync()
        dt = (time.perf_counter() - t0) / steps
        print("%-9s open %.1f s  hbm %.2f GB  launches %d  step %.1f ms  %.0f queries/s  file bytes/step %.2f GB -> %.1f GB/s  bit-exact=%s"
              % (label, t_open, s.info(0).hbm_bytes / 1e9, b.stats()["scan_launches"], dt * 1e3, nq / dt, size / 1e9,
                 size / dt / 1e9 if budget else float("nan"), ok), flush=True)
        del b, s
    os.remove(path)


if __name__ == "__main__":
    main()
