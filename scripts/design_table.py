"""Rewrites the rows of DESIGN.md section 4's shape table from profiles/<TAG>_shapes.json, so the
table always shows what scripts/profile_shapes.sh measured (python scripts/design_table.py r02)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = ["c3", "c2", "c4", "c3h3", "reads50", "reads100", "reads150", "c3hits", "reads100hits", "c3top10", "c3top10rows",
        "reads100top10"]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    shapes = json.load(open(os.path.join(ROOT, "profiles", tag + "_shapes.json")))
    path = os.path.join(ROOT, "DESIGN.md")
    lines = open(path).read().split("\n")
    head = next(i for i, ln in enumerate(lines) if ln.startswith("| shape (`scripts/profile_shapes.sh`)"))
    body = head + 2
    for n, key in enumerate(ROWS):
        cells = lines[body + n].split("|")
        s = shapes[key]
        scan = next(v for k, v in s["kernels"].items() if "scan_kernel" in k)
        cells[3] = " %.2f / %.2f ms " % (s["hip_event_scan_ms"], scan["avg_ms"])
        cells[4] = " %.1f GB " % (s["algorithmic_bytes_per_launch"] / 1e9)
        bold = "**" in cells[5]
        frac = "%.3f" % s["frac_of_8TBps_hip_events"]
        cells[5] = " **%s** " % frac if bold else " %s " % frac
        note = re.sub(r"^\s*[0-9.]+", "", cells[6])
        ratio = s["hbm_bytes_per_launch"] / s["algorithmic_bytes_per_launch"]
        cells[6] = " %.3f%s" % (ratio, note if note.strip() else " ")
        lines[body + n] = "|".join(cells)
        print(key, cells[3], cells[5], cells[6])
    open(path, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
