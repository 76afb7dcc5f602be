#!/bin/bash
# scripts/profile_c5.sh TAG -- BASELINE configs[4] on one GPU: the C3 index as an 18.4 GB
# .cobs_compact FILE (written once by the generator), streamed under an HBM budget of 6 GB through
# the two shared device buffers; bench line + rocprofv3 kernel / memory-copy trace with stats.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_c5
mkdir -p "$OUT"
export TMPDIR=/tmp
python $REPO/bench.py --config c5 --hbm-budget-gb 6 --steps 3 --warmup 1 > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
  python $REPO/bench.py --config c5 --hbm-budget-gb 6 --steps 2 --warmup 1 > "$OUT/trace.log" 2>&1
cd "$REPO"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" ! -name "*.err" -delete
find "$OUT" -name "*_trace.csv" -size +2M -delete
ls -la "$OUT" "$OUT/trace" 2>/dev/null | head -30
tail -c 400 "$OUT/bench_c5.json"
