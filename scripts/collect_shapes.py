#!/usr/bin/env python3
"""scripts/collect_shapes.py TAG -- copy the rocprofv3 summaries of scripts/profile_shapes.sh from
gpurun_out/prof_TAG/<shape>/ (scratch) into profiles/ (tracked):

  profiles/TAG_<shape>_kernel_stats.csv   `rocprofv3 --kernel-trace --stats` per-kernel table
  profiles/TAG_<shape>_bench.json         the bench line the profiled process printed (HIP-event
                                          duration of the same launches, roofline)
  profiles/TAG_shapes.json                one entry per shape: kernel average from the stats table,
                                          HIP-event duration, algorithmic bytes, PMC traffic
  profiles/traffic.json                   HBM bytes per launch keyed by bench.py's workload key,
                                          stamped with the kernel source hash + git head

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) FETCH_SIZE on gfx950 counts a 128-byte request as 64 bytes, so the read side is
doubled; WRITE_SIZE is used as is."""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ("scan_kernel", "topk_kernel", "hash_kernel")


def counters(path, kernel="scan_kernel"):
    agg, meta = collections.defaultdict(list), {}
    if not os.path.exists(path):
        return {}, {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count",
                                      "Scratch_Size") if k in r}
            meta["Kernel_Name"] = r["Kernel_Name"]
    return {k: sum(v) / len(v) for k, v in agg.items()}, meta


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    tr_path = os.path.join(dst, "traffic.json")
    traffic = {}
    if os.path.exists(tr_path):
        try:
            old = json.load(open(tr_path))
            traffic = {k: v for k, v in old.items() if isinstance(v, dict) and "kernels_hash" in v}
        except Exception:
            traffic = {}
    shapes = {}
    for shape in sorted(os.listdir(src)):
        d = os.path.join(src, shape)
        stats = os.path.join(d, "trace", "bench_kernel_stats.csv")
        if not os.path.isdir(d) or not os.path.exists(stats):
            continue
        shutil.copy(stats, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, shape)))
        bench = None
        for ln in open(os.path.join(d, "trace.log"), errors="replace"):
            if ln.startswith('{"metric"'):
                bench = json.loads(ln)
        if bench is None:
            continue
        with open(os.path.join(dst, "%s_%s_bench.json" % (tag, shape)), "w") as f:
            json.dump(bench, f)
            f.write("\n")
        ent = {"bench_args": bench["config"]["workload"], "workload_key": bench.get("workload_key"),
               "kernels_hash": bench.get("kernels_hash"), "git_head": head,
               "hip_event_scan_ms": bench["roofline"]["scan_ms_per_launch"],
               "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
               "frac_of_8TBps_hip_events": bench["roofline"]["frac"], "kernels": {}}
        for r in csv.DictReader(open(stats)):
            if any(h in r["Name"] for h in HOT):
                ent["kernels"][r["Name"].replace("cobs_amd::", "")] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
        scan = [v for k, v in ent["kernels"].items() if "scan_kernel" in k]
        if scan:
            ent["rocprof_scan_avg_ms"] = scan[0]["avg_ms"]
            ent["frac_of_8TBps_rocprof"] = round(ent["algorithmic_bytes_per_launch"] / (scan[0]["avg_ms"] * 1e-3) / 8e12, 4)
        pmc = {}
        for sub in sorted(os.listdir(d)):
            c, meta = counters(os.path.join(d, sub, "bench_counter_collection.csv"))
            pmc.update(c)
            if meta:
                ent["dispatch"] = meta
        if pmc:
            ent["pmc_per_launch"] = pmc
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            rd, wr = pmc["FETCH_SIZE"] * 1024 * 2, pmc["WRITE_SIZE"] * 1024
            ent["hbm_read_bytes"], ent["hbm_write_bytes"], ent["hbm_bytes_per_launch"] = rd, wr, rd + wr
            ent["traffic_over_algorithmic"] = round((rd + wr) / ent["algorithmic_bytes_per_launch"], 4)
            if ent["workload_key"]:
                traffic[ent["workload_key"]] = {
                    "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "tag": tag, "shape": shape,
                    "kernels_hash": ent["kernels_hash"], "git_head": head,
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled per "
                              "MI355X_MICROARCH.md"}
        if "TCC_EA0_RDREQ_sum" in pmc and "TCC_EA0_RDREQ_DRAM_sum" in pmc:
            # can gfx950's counters tell HBM from the Infinity Cache?  "destined for DRAM" vs all memory-side read requests
            rq, dr = pmc["TCC_EA0_RDREQ_sum"], pmc["TCC_EA0_RDREQ_DRAM_sum"]
            ent["ea_read_requests"] = {"all": rq, "dram_destined": dr, "bytes_at_128_per_request": rq * 128,
                                       "dram_destined_share": round(dr / rq, 6) if rq else None}
            if ent["workload_key"] and ent["workload_key"] in traffic:
                traffic[ent["workload_key"]]["infinity_cache_separable"] = (
                    "no: TCC_EA0_RDREQ_DRAM_sum == TCC_EA0_RDREQ_sum (%d of %d requests): the Infinity Cache is memory-side, behind "
                    "the counters of the L2's fabric interface; HBM-pin traffic is bounded from below by the distinct rows of a "
                    "launch (bench.py roofline.unique_bytes_per_launch) and measured at a cache-cold batch (roofline.cache_cold)"
                    % (dr, rq)) if abs(dr - rq) <= 1e-6 * max(rq, 1) else "share of DRAM-destined read requests: %.4f" % (dr / rq)
        if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
            ent["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4)
        shapes[shape] = ent
    with open(os.path.join(dst, tag + "_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=1, sort_keys=True)
    with open(tr_path, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    for k, e in shapes.items():
        print("%-14s hip %.3f ms  rocprof %.3f ms  frac %.3f  traffic/algo %s" %
              (k, e["hip_event_scan_ms"], e.get("rocprof_scan_avg_ms", float("nan")), e["frac_of_8TBps_hip_events"],
               e.get("traffic_over_algorithmic")))


if __name__ == "__main__":
    main()
