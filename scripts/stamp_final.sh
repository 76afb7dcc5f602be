#!/bin/bash
# scripts/stamp_final.sh TAG -- the lines of a round that depend on the HOST side of the calls (the scan kernels and their
# rocprofv3 shape set are stamped by scripts/stamp_round.sh and keyed on kernels.hip): the default bench line as the
# driver will run it, with the rocprofv3 kernel table of the SAME command (every kernel of the process: scan, ranking,
# pool ordering, the probes of the line), the one-rank-sharded lines, the default-call A/B and the small-batch latencies.
set -u
TAG=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/final_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
python bench.py --one-rank-sharded --no-cpu-baseline > "$OUT/bench_c3_one_rank_sharded.json" 2> /dev/null
python scripts/default_call.py 256 > "$OUT/default_call.txt" 2>&1
python scripts/probes/fresh_buffer_probe.py > "$OUT/fresh_buffer_probe.txt" 2>&1
python scripts/latency.py > "$OUT/latency.txt" 2>&1
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
