#!/bin/bash
# scripts/asan_mock_ranks.sh [N ...] -- the multi-GPU host code (sharded.cpp, comm.cpp, multi.cpp: worker threads, the
# pipeline of passes, the exchanges) under AddressSanitizer WITH more than one rank: the sanitized build of the library
# (scripts/asan_engine.sh, built first) driven by tests/mock_rccl/run_ranks.py over the
# stand-in communicator, N thread-ranks on the one GPU (default N = 2 3; with virtual device ordinals for the first).
# The sanitizer runtime is preloaded BEFORE the stand-in.   gpurun -- 'bash scripts/asan_mock_ranks.sh'
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
bash scripts/asan_engine.sh > /dev/null || exit 1
bash tests/mock_rccl/build.sh > /dev/null || exit 1
RT=$(g++ -print-file-name=libasan.so)
# (use_sigaltstack=0: when a worker thread of the device-list handle ends, the sanitizer fails to unmap the alternate signal
# stack it gave that thread -- "failed to deallocate 0x10800 bytes", in AsanThread::Destroy, nothing of the library on the stack)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:use_sigaltstack=0
export COBS_GPU_LIBRARY="$REPO/cobs_amd/libcobs_gpu_asan.so" MOCK_RCCL_TIMEOUT_S=60
PRE="$RT $(g++ -print-file-name=libstdc++.so.6) $REPO/cobs_amd/libmockrccl.so"
rc=0
first=1
for n in ${*:-2 3}; do
  v=0; [ $first = 1 ] && v=$n; first=0
  log=gpurun_out/asan_mock_ranks_$n.log
  echo "== run_ranks.py $n (virtual device ordinals: $v) -> $log"
  MOCK_RCCL_VIRTUAL_DEVICES=$v LD_PRELOAD="$PRE" timeout 1500 python tests/mock_rccl/run_ranks.py "$n" 0 > "$log" 2>&1 || rc=1
  grep -m1 -A25 "ERROR: AddressSanitizer\|Traceback" "$log"
  tail -2 "$log"
done
# (run_batch_ranks.py and run_bench_ranks.py initialise torch's device runtime, which cannot be loaded under the preload:
# dlopen of libcaffe2_nvrtc.so fails -- as in scripts/asan_engine.sh)
exit $rc
