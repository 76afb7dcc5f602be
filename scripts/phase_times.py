#!/usr/bin/env python3
"""scripts/phase_times.py SHAPE -- where a scan work-group's time goes (tuning build only:
make -C cobs_amd/csrc timing; run with COBS_GPU_LIBRARY=cobs_amd/libcobs_gpu_timing.so).
Prints the median cycles of each phase of wave 0 / 1 over the sampled work-groups."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cobs_amd  # noqa: E402

NAMES = ["entry", "setup (LUT, page + block-offset loads)", "first row indices", "first rows", "row loop",
         "merge", "expand + store"]


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "reads50"
    tune = dict(kv.split("=") for kv in sys.argv[2:])
    if shape.startswith("reads"):
        nq, kmers = 40000, int(shape[5:]) - 30
    else:
        nq, kmers = 10000, 1000
    cfg = bench.c3_config()
    s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
    for k, v in tune.items():
        s.set_tuning(k, int(v))
    slots = 4096
    s.set_tuning("phase_slots", slots)
    b = cobs_amd.Batch(s)
    b.set_queries(bench.make_queries(nq, kmers))
    for _ in range(3):
        b.run(0.0)
    b.sync()
    out = np.zeros(slots * 32, dtype=np.uint64)
    n = C.c_size_t(0)
    cobs_amd._capi.check(s._lib.cobs_gpu_batch_phase_stamps(b._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size, C.byref(n)))
    t = out[:n.value].reshape(-1, 4, 8).astype(np.int64)
    print("scan %.3f ms (timing build)" % b.kernel_ms()["scan_ms"])
    for w in range(4):
        tw = t[:, w, :7]
        ok = (tw[:, 0] > 0) & (tw[:, 6] > 0)
        if not ok.any():
            continue
        tw = tw[ok]
        d = np.diff(tw, axis=1)
        print("wave %d: %d samples, life %d cycles (median)" % (w, len(tw), int(np.median(tw[:, 6] - tw[:, 0]))))
        for k in range(6):
            print("   %-42s %7d" % (NAMES[k + 1], int(np.median(d[:, k]))))


if __name__ == "__main__":
    main()
