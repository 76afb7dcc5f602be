#!/bin/bash
# scripts/collect_stamp.sh TAG -- after scripts/stamp_round.sh TAG ran on the GPU box (gpurun merges gpurun_out/ back):
# copy what profiles/ quotes from gpurun_out/stamp_TAG/ and gpurun_out/prof_TAG/ into profiles/TAG_* (tracked).
set -eu
TAG=${1:-r06}
S=gpurun_out/stamp_$TAG
for f in bench_c3 bench_c3_one_rank_sharded bench_c3_one_rank_sharded_1chunk bench_c2 c5_bench c5_q256_bench c5_columns_bench c5_per_file_bench c5_buf512_bench c5_184GB_bench c5_184GB_hits_bench c5_184GB_per_file_bench; do
  [ -s "$S/$f.json" ] && grep -h '"metric"' "$S/$f.json" | tail -1 > "profiles/${TAG}_$f.json"
done
for f in default_call latency sharded_call c5_selective h2d_probe clk_probe shard_times fresh_buffer_probe; do
  [ -s "$S/$f.txt" ] && grep -v "amdgpu.ids" "$S/$f.txt" > "profiles/${TAG}_$f.txt"
done
python scripts/collect_shapes.py "$TAG"
python scripts/isa_resources.py "profiles/${TAG}_isa_resources.txt" > /dev/null
ls profiles | grep "^${TAG}_" | wc -l
