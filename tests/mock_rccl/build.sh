#!/bin/bash
# tests/mock_rccl/build.sh -- TEST INFRASTRUCTURE: cobs_amd/libmockrccl.so (mock_rccl.cpp).  Preloaded (LD_PRELOAD) into
# a process it answers the RCCL calls of the SHIPPED cobs_amd/libcobs_gpu.so in place of librccl.
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
g++ -O1 -g -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wall -Wno-deprecated-declarations \
    "$HERE/mock_rccl.cpp" -o "$REPO/cobs_amd/libmockrccl.so" -L/opt/rocm/lib -lamdhip64 -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
echo "built cobs_amd/libmockrccl.so"
