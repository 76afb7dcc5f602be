#!/bin/bash
# scripts/profile.sh TAG -- rocprofv3 runs of the headline bench on the GPU box.
# Writes raw output under gpurun_out/prof_TAG/ (scratch); the summaries that are
# judged get copied into profiles/ by scripts/collect_profiles.py.
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"     # extra bench.py arguments (shape under the profiler)
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 2 $EXTRA"
cd /tmp
# 1) kernel trace + stats (per-kernel durations)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
# 2) PMC passes, one counter group per run (never combined with other trace domains)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -o bench -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o bench -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.csv" | head -50
# keep the merged-back payload small: drop everything but csv/log
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" -delete
du -sh "$OUT"
