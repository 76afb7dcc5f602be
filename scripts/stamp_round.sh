#!/bin/bash
# scripts/stamp_round.sh TAG -- everything profiles/ quotes for a round, measured ONCE, at its end, on the GPU box:
# the rocprofv3 shape set (scripts/profile_shapes.sh), the plain / one-rank-sharded / out-of-core bench lines, the
# default-call A/B, the small-batch latencies and the row-selective sweep.  Raw output: gpurun_out/prof_TAG/ and
# gpurun_out/stamp_TAG/ (scratch); scripts/collect_shapes.py TAG + scripts/collect_stamp.sh TAG copy the summaries
# into profiles/ (tracked).
set -u
TAG=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/stamp_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
python bench.py --one-rank-sharded --no-cpu-baseline > "$OUT/bench_c3_one_rank_sharded.json" 2> "$OUT/bench_c3_one_rank_sharded.err"
python bench.py --one-rank-sharded --no-cpu-baseline --exchange-chunks 1 > "$OUT/bench_c3_one_rank_sharded_1chunk.json" 2>/dev/null
python bench.py --config c2 --no-cpu-baseline > "$OUT/bench_c2.json" 2>/dev/null
python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_bench.json" 2> "$OUT/c5_bench.err"
python bench.py --config c5 --no-cpu-baseline --steps 5 --warmup 2 --queries 256 > "$OUT/c5_q256_bench.json" 2>/dev/null
COBS_GPU_ROW_RANGES=0 python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_columns_bench.json" 2>/dev/null
# round 4's plan (residency per FILE: the whole file streamed through two buffers of half the budget), and 512 MiB buffers
COBS_GPU_STREAM_BUF_KIB=0 python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_per_file_bench.json" 2>/dev/null
COBS_GPU_STREAM_BUF_KIB=524288 python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_buf512_bench.json" 2>/dev/null
rm -f /tmp/cobs_c5_1.cobs_compact
# configs[4] at its named size: a 184 GB file (tmpfs) under a 64 GB budget, per-slice residency and round 4's per-file plan
timeout 1500 python bench.py --config c5 --scale 10 --hbm-budget-gb 64 --index-file /dev/shm/cobs_c5_10.cobs_compact \
    --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/c5_184GB_bench.json" 2> "$OUT/c5_184GB_bench.err"
COBS_GPU_STREAM_BUF_KIB=0 timeout 1500 python bench.py --config c5 --scale 10 --hbm-budget-gb 64 --index-file /dev/shm/cobs_c5_10.cobs_compact \
    --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/c5_184GB_per_file_bench.json" 2> /dev/null
rm -f /dev/shm/cobs_c5_10.cobs_compact
python scripts/default_call.py 256 > "$OUT/default_call.txt" 2>&1
python scripts/probes/fresh_buffer_probe.py > "$OUT/fresh_buffer_probe.txt" 2>&1
python scripts/latency.py > "$OUT/latency.txt" 2>&1
python scripts/row_fetch_bench.py 1.0 6 > "$OUT/c5_selective.txt" 2>&1
scripts/probes/h2d_probe 4 > "$OUT/h2d_probe.txt" 2>&1
python scripts/clk_probe.py > "$OUT/clk_probe.txt" 2>&1
(for a in "8 0" "4 0" "2 0" "3 0" "8 1" "8 2"; do echo "== shards: N mode = $a"; python scripts/shard_times.py $a; done
 for m in 2 0; do echo "== streamed shards of the C3 file, 0.75 GB budget each: N mode = 8 $m"; python scripts/shard_times.py 8 $m 10000 0.75; done
 rm -f /tmp/cobs_c5_1.cobs_compact) > "$OUT/shard_times.txt" 2>&1
bash scripts/profile_shapes.sh "$TAG" > "$OUT/profile_shapes.log" 2>&1
tail -15 "$OUT/profile_shapes.log"
ls "$OUT"
