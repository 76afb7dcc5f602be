#!/bin/bash
# scripts/stamp_out_of_core.sh TAG -- the out-of-core lines of scripts/stamp_round.sh on their own (the streamed path
# lives in plan.cpp / stage.cpp / pass.cpp, outside the kernels the traffic replay is keyed on), plus configs[4] at
# its named size: a 184 GB file under a 64 GB budget.
set -u
TAG=${1:-r06}
OUT=$(pwd)/gpurun_out/stamp_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_bench.json" 2> "$OUT/c5_bench.err"
python bench.py --config c5 --no-cpu-baseline --steps 5 --warmup 2 --queries 256 > "$OUT/c5_q256_bench.json" 2>/dev/null
COBS_GPU_ROW_RANGES=0 python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_columns_bench.json" 2>/dev/null
COBS_GPU_STREAM_PACKED=0 python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > "$OUT/c5_device_pitch_bench.json" 2>/dev/null
python scripts/row_fetch_bench.py 1.0 6 > "$OUT/c5_selective.txt" 2>&1
rm -f /tmp/cobs_c5_1.cobs_compact
(for m in 2 0; do echo "== streamed shards of the C3 file, 0.75 GB budget each: N mode = 8 $m"; python scripts/shard_times.py 8 $m 10000 0.75; done) > "$OUT/shard_times_streamed.txt" 2>&1
rm -f /tmp/cobs_c5_1.cobs_compact
if [ "${2:-}" = "full" ]; then
  # (the box's /tmp is a 79 GB overlay; /dev/shm is tmpfs over its 3 TB of RAM)
  timeout 1500 python bench.py --config c5 --scale 10 --hbm-budget-gb 64 --index-file /dev/shm/cobs_c5_10.cobs_compact \
      --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/c5_184GB_bench.json" 2> "$OUT/c5_184GB_bench.err"
  # the same pass at the CLI's default threshold, hits only: the three streamed sub-indexes are counted in row ranges, the
  # selection runs after each one's last range -- no score rows (round 6; rounds 4-5 wrote 2 GB of them per pass)
  timeout 900 python bench.py --config c5 --scale 10 --hbm-budget-gb 64 --index-file /dev/shm/cobs_c5_10.cobs_compact \
      --no-cpu-baseline --steps 2 --warmup 1 --threshold 0.8 --hits-only > "$OUT/c5_184GB_hits_bench.json" 2> /dev/null
  rm -f /dev/shm/cobs_c5_10.cobs_compact
fi
grep -h '"metric"' "$OUT"/c5*_bench.json | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['config'].get('workload','')[:60], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'], j.get('bit_exact_vs_oracle'))"
tail -12 "$OUT/shard_times_streamed.txt"
