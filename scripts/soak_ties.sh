export TMPDIR=/tmp
for s in $(seq ${1:-1} ${2:-60}); do
  r=$(COBS_FUZZ_SEED=$s timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k random_ties 2>&1 | tail -1)
  echo "seed $s: $r"
  case "$r" in *failed*) COBS_FUZZ_SEED=$s python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k random_ties 2>&1 | grep -E "^E  |AssertionError" | head -12; exit 1;; esac
done
