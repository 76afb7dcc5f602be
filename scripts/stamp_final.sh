#!/bin/bash
# scripts/stamp_final.sh TAG -- the lines of a round that depend on the HOST side of the calls (the scan kernels and their
# rocprofv3 shape set are stamped by scripts/stamp_round.sh and keyed on kernels.hip): the default bench line as the
# driver will run it, with the rocprofv3 kernel table of the SAME command (every kernel of the process: scan, ranking,
# pool ordering, the probes of the line), the one-rank-sharded lines, the default-call A/B and the small-batch latencies.
set -u
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# The default line REPLAYS roofline.traffic from profiles/traffic.json, keyed on the kernel source (bench.kernels_hash):
# a late edit of kernels.hip / geometry.cpp silently turns it into `traffic: null` (that happened in round 5).  This script
# is the round's last step, so it refuses to run on a stale stamp: re-run `scripts/profile_shapes.sh TAG c3 ...` on the
# GPU box and `python scripts/collect_shapes.py TAG` here first.  [VERDICT r5 item 8]
python - <<'PY' || { echo "stamp_final.sh: STALE TRAFFIC STAMP -- see above; nothing was measured" >&2; exit 7; }
import json, sys
sys.path.insert(0, ".")
import bench
have = bench.kernels_hash()
ent = json.load(open("profiles/traffic.json")).get("c3_q10000_k1000_h1")
if not ent or ent.get("kernels_hash") != have:
    sys.stderr.write("profiles/traffic.json[c3_q10000_k1000_h1] was measured for kernel source %s (%s), the library is built from %s:\n"
                     "the default bench line would carry roofline.traffic = null.\n" % (ent and ent.get("kernels_hash"), ent and ent.get("git_head"), have))
    sys.exit(1)
print("traffic stamp matches the kernel source (%s, measured at %s)" % (have, ent.get("git_head")))
PY
python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
python bench.py --one-rank-sharded --no-cpu-baseline > "$OUT/bench_c3_one_rank_sharded.json" 2> /dev/null
python scripts/default_call.py 256 > "$OUT/default_call.txt" 2>&1
python scripts/probes/fresh_buffer_probe.py > "$OUT/fresh_buffer_probe.txt" 2>&1
python scripts/latency.py > "$OUT/latency.txt" 2>&1
python scripts/sharded_call.py > "$OUT/sharded_call.txt" 2>&1
python scripts/probes/default_call_probe.py 16384 > "$OUT/default_call_probe.txt" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$REPO/bench.py" > "$OUT/bench_c3_profiled.json" 2> "$OUT/trace.err")
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/bench_kernel_stats.csv"
head -8 "$OUT/bench_kernel_stats.csv" | cut -c1-160
python - "$OUT/bench_c3.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("hbm_side", {}).get("batch"))
e = d["end_to_end"]
for k, v in e.items():
    if isinstance(v, dict): print(k, v.get("queries_per_s"), v.get("seconds"), v.get("hits"), v.get("fresh_result_array"), v.get("library_arena_view"), v.get("pcie_record_bytes"))
print(d["cpu_baseline"])
PY
