#!/bin/bash
# scripts/asan_index.sh [SEED] -- the index header parser (cobs_amd/csrc/index_file.cpp) under
# AddressSanitizer + UBSan on 4000 damaged copies of each golden index file.  Host only.
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/cobs_asan_index
mkdir -p "$W"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I"$REPO/cobs_amd/csrc" -I"$REPO/include" \
    "$REPO/tests/asan/index_main.cpp" "$REPO/cobs_amd/csrc/index_file.cpp" -o "$W/asan_index"
"$W/asan_index" "${1:-1}" "$REPO/tests/golden/c1.cobs_classic" "$REPO/tests/golden/c1.cobs_compact"
