#!/bin/bash
# scripts/sweep.sh -- kernel-variant / layout experiments on the GPU box (tuning hooks only)
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-34s q/s=%9.0f scan_ms=%7.3f GB/s=%7.1f frac=%.4f' % (sys.argv[1], d['value'], d['roofline']['scan_ms_per_launch'], d['roofline']['achieved'], d['roofline']['frac']))" "$1"; }
for v in 0 1 2 3 4 5 6 7 8; do COBS_GPU_SCAN_VARIANT=$v $B 2>/dev/null | pick "variant=$v"; done
for a in 64 128; do COBS_GPU_ROW_ALIGN=$a $B 2>/dev/null | pick "row_align=$a"; done
COBS_GPU_ROW_ALIGN=128 COBS_GPU_SCAN_VARIANT=3 $B 2>/dev/null | pick "row_align=128 variant=3"
