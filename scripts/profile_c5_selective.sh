#!/bin/bash
# scripts/profile_c5_selective.sh -- evidence for the row-selective out-of-core pass (run on the GPU box):
# the batch-size sweep of scripts/row_fetch_bench.py and a rocprofv3 kernel table of 64- and 256-query passes
# (fetch_rows_kernel's duration = the PCIe time of the looked-up rows).  Output: gpurun_out/r03_c5_selective/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03_c5_selective
mkdir -p "$OUT"
export TMPDIR=/dev/shm
python scripts/row_fetch_bench.py 1.0 6 --keep > "$OUT/sweep.txt" 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o fetch -- python $REPO/scripts/row_fetch_bench.py 1.0 6 --keep 64 256 > "$OUT/trace.log" 2>&1
cd "$REPO"
rm -f /dev/shm/row_fetch_1.cobs_compact
find "$OUT" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.log" -delete
cat "$OUT/sweep.txt"
head -8 "$OUT/trace/fetch_kernel_stats.csv"
