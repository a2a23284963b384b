#!/bin/bash
# tools/sweep_chunk.sh <lib.so>: kernel time against chunk_spp on the benchmark configurations
L=$1
for cfg in "1920 500" "1200 500" "3840 1000"; do
  for c in 50 20 10 5 2; do
    echo -n "chunk $c: "; RTMI_LIB=$PWD/$L RT_CHUNK=$c timeout -k 10 200 python3 tools/quick.py $cfg 1 || exit 1
  done
done
