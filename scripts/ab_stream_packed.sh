export TMPDIR=/tmp
for i in 1 2 3; do
for p in 1 0; do
COBS_GPU_STREAM_PACKED=$p python bench.py --config c5 --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read());print('packed=$p', j['ms_per_step'], j['roofline']['achieved'], j['streaming']['scan_launches_per_step'], j['streaming']['gpu_numa_node'])"
done; done
rm -f /tmp/cobs_c5_1.cobs_compact
