#!/usr/bin/env python3
"""scripts/clk_probe.py -- clocks, package power and temperatures (rocm-smi, polled from a thread) next to the scan time
of the headline shape, 2 s idle and then 8 s of back-to-back launches: the C3 scan runs with the package at its power
limit (~1.2 kW) and sclk pulled below its 2.4 GHz maximum."""
import os, sys, time, subprocess, threading, re
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, cobs_amd
cfg = bench.c3_config()
s = cobs_amd.Search.synthetic(cfg["kind"], cfg["signature_sizes"], cfg["num_docs"], page_size=cfg["page_size"], seed=1)
b = cobs_amd.Batch(s)
b.set_queries(bench.make_queries(10000, 1000))
samples = []
stop = False
def poll():
    while not stop:
        t = time.perf_counter()
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
        clk = {m.group(1): m.group(2) for m in re.finditer(r"(\w+) clock level: \d+: \((\d+)Mhz\)", out)}
        pw = re.search(r"Package Power \(W\): ([\d.]+)", out)
        tp = re.findall(r"Temperature \(Sensor (\w+)\) \(C\): ([\d.]+)", out)
        samples.append((t, clk, pw.group(1) if pw else None, dict(tp)))
th = threading.Thread(target=poll); th.start()
time.sleep(2.0)
t0 = time.perf_counter()
rows = []
while time.perf_counter() - t0 < 8.0:
    for _ in range(5):
        b.run(0.0)
    b.sync()
    rows.append((time.perf_counter(), b.kernel_ms()["scan_ms"]))
stop = True; th.join()
print("load starts at", 0.0)
for t, clk, pw, tp in samples:
    near = [ms for (tt, ms) in rows if abs(tt - t) < 0.3]
    print("t=%6.2f  %s  power %s  temp %s  scan %s" % (t - t0, clk, pw, tp, ("%.2f" % (sum(near)/len(near))) if near else "-"))
