#!/usr/bin/env python3
"""scripts/calibrate_cpu_baseline.py -- is the CPU baseline (oracle/cobs_oracle.c, the plain-C port of the reference's
query path) as fast as the reference itself?  BASELINE.md section 4 promised +-15 % on the two shapes of its section 2,
where the REAL reference (its own sources, g++ -O3 -march=native, tlx replaced by a stand-in) was timed during the
survey: classic D = 10 000 / S = 1 000 000 and compact D = 100 000 / P = 8 / page_size 1568 (2.55 GB), 1000-k-mer
queries, threshold 0.8, index in RAM, bytes random with ~25 % ones, 1 and 8 threads, on an 8-vCPU Xeon @ 2.1 GHz -- the
CPU model of the build container this script runs in (the reference cannot be built here any more: extlib/tlx is empty
and a stand-in would pin nothing, so the survey's numbers are the other side of the comparison).
Runs on the CPU only (harness shape: reference src/cobs.cpp:605-671).  -> JSON on stdout; BASELINE.md section 4 and
bench.py's cpu_baseline.calibration quote it (profiles/r06_cpu_calibration.json)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

SURVEY = {  # queries/s of the real reference
    "classic_D10000_S1000000": {1: 434.0, 8: 549.0},
    "compact_D100000_P8_ps1568_2.55GB": {1: 89.0, 8: 74.0},
}


def random_pages(rows, width, seed):
    rs = np.random.RandomState(seed)
    out = []
    for s in rows:
        a = rs.randint(0, 256, size=(s, width), dtype=np.uint8)
        a &= rs.randint(0, 256, size=(s, width), dtype=np.uint8)          # ~25 % ones
        out.append(a)
    return out


def queries(n, kmers=1000, seed=42):
    rs = np.random.RandomState(seed)
    raw = rs.randint(0, 2 ** 32, size=n * (kmers + 30), dtype=np.uint64)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[(raw % 4).astype(np.int64)].reshape(n, kmers + 30)
    return [text[i].tobytes() for i in range(n)]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    O.build(native=True, force=True)
    qs = queries(2000)
    res = {"host": {"cpus": os.cpu_count(), "model": next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")},
           "seconds_per_point": seconds, "shapes": {}}
    shapes = [("classic_D10000_S1000000", 0, 10000, 0, [1000000]),
              ("compact_D100000_P8_ps1568_2.55GB", 1, 100000, 1568, [int(2.55e9 / 1568 / 8)] * 8)]
    for name, kind, D, ps, sigs in shapes:
        width = ps if kind else (D + 7) // 8
        ix = O.Index.from_memory(kind, 31, 1, 1, ps, sigs, D, random_pages(sigs, width, 1))
        ent = {}
        for threads in (1, 8):
            O.search_many(ix, qs[:20], 0.8, 0, threads=threads, seconds=2.0)       # warm
            n, dt = O.search_many(ix, qs, 0.8, 0, threads=threads, seconds=seconds)
            qps = n / dt
            ref = SURVEY[name][threads]
            ent["threads_%d" % threads] = {"port_queries_per_s": round(qps, 1), "reference_queries_per_s_survey": ref,
                                           "port_over_reference": round(qps / ref, 3), "queries": n}
        res["shapes"][name] = ent
        del ix
    ratios = [v["port_over_reference"] for e in res["shapes"].values() for v in e.values()]
    res["within_15_percent_everywhere"] = bool(all(0.85 <= r <= 1.15 for r in ratios))
    res["port_over_reference_min_max"] = [min(ratios), max(ratios)]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
