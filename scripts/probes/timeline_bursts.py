"""scripts/probes/timeline_bursts.py KERNEL_TRACE.csv [gap_ms] -- kernels grouped into bursts separated by idle gaps of
more than gap_ms (default 20): span, device-busy time (union of the kernel intervals) and the kernels by time."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
bursts, cur = [], []
for e in ev:
    if cur and (e[0] - max(x[1] for x in cur)) / 1e6 > gap:
        bursts.append(cur); cur = []
    cur.append(e)
if cur: bursts.append(cur)
def short(n):
    return n.replace("cobs_amd::", "").replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
for b in bursts:
    t0, t1 = b[0][0], max(x[1] for x in b)
    busy, end = 0, t0
    for s_, e_, _ in b:
        if e_ > end:
            busy += e_ - max(s_, end); end = e_
    by = collections.Counter()
    for s_, e_, n in b: by[short(n)] += e_ - s_
    if (t1 - t0) < 30e3: continue
    print("burst %9.3f ms  busy %9.3f ms (%.0f %%)  kernels %4d : %s" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(b),
          ", ".join("%s %.3f" % (k, v / 1e6) for k, v in by.most_common(5))))
