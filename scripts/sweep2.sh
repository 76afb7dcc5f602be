#!/bin/bash
# MALL (Infinity Cache) residency experiment: same geometry, signature sizes scaled
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-34s q/s=%9.0f scan_ms=%7.3f GB/s=%7.1f frac=%.4f' % (sys.argv[1], d['value'], d['roofline']['scan_ms_per_launch'], d['roofline']['achieved'], d['roofline']['frac']))" "$1"; }
for s in 0.015625 0.03125 0.0625 0.125 0.25 0.5 1; do $B --scale $s 2>/dev/null | pick "scale=$s"; done
