#!/bin/bash
# scripts/fuzz_soak.sh FIRST LAST -- the seeded fuzz tests (random index geometries, budgets, queries, document sets)
# under seeds FIRST..LAST (COBS_FUZZ_SEED; the suite itself runs seed 0).  Prints one line per seed; stops at the
# first failure and leaves its log in gpurun_out/fuzz_soak_SEED.log.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for seed in $(seq "${1:-1}" "${2:-10}"); do
  log=gpurun_out/fuzz_soak_$seed.log
  COBS_FUZZ_SEED=$seed timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_streaming.py tests/test_gpu_construct.py \
      tests/test_gpu_sharded.py::test_sharded_random_ties tests/test_gpu_rccl.py::test_native_sharded_search_random_ties tests/test_gpu_mock_ranks.py -x -q -m gpu > "$log" 2>&1
  rc=$?
  echo "seed $seed rc $rc: $(tail -1 "$log")"
  if [ $rc -ne 0 ]; then tail -40 "$log"; exit $rc; fi
  rm -f "$log"
done
