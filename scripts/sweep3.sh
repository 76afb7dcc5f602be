#!/bin/bash
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-34s q/s=%9.0f step_ms=%7.3f scan_ms=%7.3f GB/s=%7.1f frac=%.4f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['scan_ms_per_launch'], d['roofline']['achieved'], d['roofline']['frac']))" "$1"; }
for w in 64 32 16 8 4; do COBS_GPU_TILE_W=$w $B 2>/dev/null | pick "tile_w=$w"; done
for m in 64 96 128 144 176 208 240; do COBS_GPU_MALL_MB=$m $B 2>/dev/null | pick "auto mall=$m"; done
