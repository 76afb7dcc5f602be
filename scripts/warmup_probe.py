#!/usr/bin/env python3
"""scripts/warmup_probe.py [SHAPE] -- scan time against time under load: how long does the device take to reach the rate
it then sustains?  (Round 4: the C3 scan starts at 19.9-20.5 ms after the index is created and settles at 18.8 ms.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c3"
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    cfg = bench.c3_config()
    nq, kmers = (40000, int(shape[5:]) - 30) if shape.startswith("reads") else (10000, 1000)
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    b = cobs_amd.Batch(s)
    b.set_queries(bench.make_queries(nq, kmers))
    if idle:
        time.sleep(idle)
    t0 = time.perf_counter()
    group = 5 if shape == "c3" else 40
    line = []
    while time.perf_counter() - t0 < 6.0:
        for _ in range(group):
            b.run(0.0)
        b.sync()
        line.append((time.perf_counter() - t0, b.kernel_ms()["scan_ms"]))
    for t, ms in line:
        print("t = %5.2f s   scan %.3f ms" % (t, ms))


if __name__ == "__main__":
    main()
