#!/bin/bash
# scripts/rw_mix_compare.sh -- the read/write-mix probe and the scan kernel on the SAME box (boxes differ by +-4 %):
# scripts/probes/rw_mix_probe (all waves / 32 / 16 waves per CU) next to bench.py's HIP-event scan times of the
# shapes it models.  Output: $1 (default gpurun_out/rw_mix_compare.txt)
set -u
OUT=${1:-gpurun_out/rw_mix_compare.txt}
mkdir -p "$(dirname "$OUT")"
[ -x scripts/probes/rw_mix_probe ] || hipcc --offload-arch=gfx950 -O3 scripts/probes/rw_mix_probe.hip -o scripts/probes/rw_mix_probe
{
  for l in 0 5120 10240; do scripts/probes/rw_mix_probe $l | grep -v "anywhere\|gather only"; done
  for args in "--queries 40000 --kmers 20" "--queries 40000 --kmers 70" "--queries 40000 --kmers 120" ""; do
    python bench.py --no-cpu-baseline --steps 6 --warmup 2 $args 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('scan kernel, %s: %.3f ms per launch (x4 for 40 000 queries: %.2f ms)  %.1f GB/s algorithmic' % (j['workload_key'], j['roofline']['scan_ms_per_launch'], 4*j['roofline']['scan_ms_per_launch'], j['roofline']['achieved']))"
  done
} 2>&1 | tee $OUT
