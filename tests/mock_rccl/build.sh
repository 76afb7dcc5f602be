#!/bin/bash
# tests/mock_rccl/build.sh -- TEST INFRASTRUCTURE: libmockrccl.so (mock_rccl.cpp) and a copy of the library linked
# against it instead of librccl: cobs_amd/libcobs_gpu_mockrccl.so (the same object files as libcobs_gpu.so).
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
REPO=$(cd "$HERE/../.." && pwd)
SRC=$REPO/cobs_amd/csrc
make -C "$SRC" > /dev/null
g++ -O1 -g -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wall -Wno-deprecated-declarations \
    "$HERE/mock_rccl.cpp" -o "$REPO/cobs_amd/libmockrccl.so" -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib
OBJS=$(sed -n 's/^OBJS := //p' "$SRC/Makefile")
(cd "$SRC" && hipcc --offload-arch=gfx950 -shared -o "$REPO/cobs_amd/libcobs_gpu_mockrccl.so" $OBJS -L"$REPO/cobs_amd" -lmockrccl -lz \
    -Wl,-rpath,'$ORIGIN')
echo "built cobs_amd/libcobs_gpu_mockrccl.so"
