#!/bin/bash
# scripts/profile_shapes.sh TAG [shape ...] -- run ONCE per round, at its end (scripts/collect_shapes.py TAG then stamps
# profiles/): rocprofv3 evidence for every shape DESIGN.md quotes,
# on the GPU box.  Per shape: one --kernel-trace --stats run of bench.py (per-kernel durations + the
# bench line of the profiled process) and separate --pmc passes (FETCH_SIZE, WRITE_SIZE; for the
# shapes marked "full" also L2 hits/misses and SQ wave/wait counters).  PMC passes never carry
# another trace domain.  Raw output: gpurun_out/prof_TAG/<shape>/ (scratch);
# scripts/collect_shapes.py TAG copies the summaries into profiles/.
set -u
TAG=${1:-r03}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
declare -A SHAPES=(
  [c3]=""
  [c3q256]="--queries 256"
  [c3x8]="--scale 8"
  [c3x8q2500]="--scale 8 --queries 2500"
  [c2]="--config c2"
  [c4]="--config c4 --queries 1000"
  [c3h3]="--num-hashes 3 --queries 4000"
  [reads50]="--queries 40000 --kmers 20"
  [reads100]="--queries 40000 --kmers 70"
  [reads150]="--queries 40000 --kmers 120"
  [c3hits]="--threshold 0.8 --hits-only"
  [reads100hits]="--queries 40000 --kmers 70 --threshold 0.8 --hits-only"
  [c3top10]="--num-results 10"
  [c3top10rows]="--num-results 10 --topk-with-rows"
  [reads100top10]="--queries 40000 --kmers 70 --num-results 10"
)
FULL="c3 c3q256 c3x8 reads50"
WANT=${*:-c3 c3q256 c3x8 c3x8q2500 c2 c4 c3h3 reads50 reads100 reads150 c3hits reads100hits c3top10 c3top10rows reads100top10}
cd /tmp
for shape in $WANT; do
  EXTRA=${SHAPES[$shape]}
  D="$OUT/$shape"
  mkdir -p "$D"
  BENCH="python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 2 $EXTRA"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -o bench -- $BENCH > "$D/trace.log" 2>&1
  SHORT="python $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 $EXTRA"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$D/pmc_fetch" -o bench -- $SHORT > "$D/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$D/pmc_write" -o bench -- $SHORT > "$D/pmc_write.log" 2>&1
  if [[ " $FULL " == *" $shape "* ]]; then
    # memory-side request counters of the L2: all requests vs those "destined for DRAM" (do Infinity-Cache hits separate? r04: no)
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$D/pmc_ea_rd" -o bench -- $SHORT > "$D/pmc_ea_rd.log" 2>&1
    rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d "$D/pmc_ea_wr" -o bench -- $SHORT > "$D/pmc_ea_wr.log" 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$D/pmc_tcc" -o bench -- $SHORT > "$D/pmc_tcc.log" 2>&1
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$D/pmc_sq" -o bench -- $SHORT > "$D/pmc_sq.log" 2>&1
    rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d "$D/pmc_lds" -o bench -- $SHORT > "$D/pmc_lds.log" 2>&1
  fi
  grep -h '"metric"' "$D/trace.log" | tail -1 | cut -c1-200
done
cd "$REPO"
# keep the merged-back payload small: csv / log only, and no per-dispatch traces of the PMC passes
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" -delete
find "$OUT" -path "*pmc_*" -name "*kernel_trace.csv" -delete
du -sh "$OUT"
