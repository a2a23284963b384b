#!/bin/bash
# Samples per work unit: the library's choice (-1) against fixed sizes, large and small launches
set -o pipefail
for c in -1 4 8 16 32; do
  echo "== chunk $c"
  RT_CHUNK=$c timeout -k 10 100 python tools/quick.py 3840 250 2 0 | tail -1 || exit 1
  RT_CHUNK=$c timeout -k 10 100 python tools/quick.py 1920 125 2 0 | tail -1 || exit 1
  RT_CHUNK=$c timeout -k 10 100 python tools/quick.py 1200 500 2 0 | tail -1 || exit 1
done
