#!/bin/bash
# scripts/profile_construct.sh TAG [--docs N --doc-mb M ...] -- rocprofv3 evidence for GPU index
# construction: the bench line, build_kernel's time (--kernel-trace --stats) and
# its HBM traffic (separate --pmc passes, never combined with other traces).  Run on the GPU box:
#   gpurun -- 'bash scripts/profile_construct.sh r02'
# Results: gpurun_out/construct_<TAG>/...; the summaries are copied to profiles/ by hand
# (profiles/<TAG>_construct_*).
set -u
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/construct_$TAG
mkdir -p "$D"
ARGS="--docs 256 --doc-mb 4 $*"
python scripts/construct_bench.py $ARGS > "$D/bench.json" 2> "$D/bench.err"
tail -1 "$D/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o cb -- python scripts/construct_bench.py $ARGS --cpu-seconds 1 > "$D/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$D/pmc_fetch" -o cb -- python scripts/construct_bench.py $ARGS --cpu-seconds 1 > "$D/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$D/pmc_write" -o cb -- python scripts/construct_bench.py $ARGS --cpu-seconds 1 > "$D/pmc_write.log" 2>&1
find "$D" -name "*kernel_stats.csv" | head -1 | xargs head -6
python - "$D" <<'PY'
import csv, glob, sys
d = sys.argv[1]
for name in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(d + "/" + name + "/**/*counter_collection.csv", recursive=True):
        tot, n = {}, {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
        for k in tot:
            if "build_kernel" in k:
                print(name, k, "launches", n[k], "sum", tot[k], "per launch", tot[k] / n[k])
PY
