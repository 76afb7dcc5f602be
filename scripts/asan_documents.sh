#!/bin/bash
# scripts/asan_documents.sh [SEED ...] -- the document readers (cobs_amd/csrc/documents.cpp) under
# AddressSanitizer + UBSan on damaged copies of every fixture and of a generated corpus (truncated,
# bit-flipped, header fields overwritten, garbage tails): ~2100 files per seed.  Host only.
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/cobs_asan_documents
mkdir -p "$W"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I"$REPO/cobs_amd/csrc" -I"$REPO/include" \
    "$REPO/tests/asan/documents_main.cpp" "$REPO/cobs_amd/csrc/documents.cpp" -o "$W/asan_docs" -lz -lpthread
for seed in ${*:-1 2 3}; do
  rm -rf "$W/corpus" "$W/gen"
  python "$REPO/tests/asan/damaged_corpus.py" "$seed" "$W" > /dev/null
  ls "$W"/corpus/* | xargs "$W/asan_docs"
done
