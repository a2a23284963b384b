#!/bin/bash
# tools/ab2.sh <a.so> <b.so>: the two builds on large, medium and one-sample-per-pixel launches
for L in $1 $2 $1 $2; do
  echo "== $L"
  RTMI_LIB=$PWD/$L RT_CHUNK=-1 timeout -k 10 200 python3 tools/quick.py 1920 1000 1 || exit 1
  RTMI_LIB=$PWD/$L RT_CHUNK=-1 timeout -k 10 200 python3 tools/quick.py 1920 125 2 || exit 1
  RTMI_LIB=$PWD/$L RT_CHUNK=-1 timeout -k 10 200 python3 tools/quick.py 1920 8 2 || exit 1
  RTMI_LIB=$PWD/$L RT_CHUNK=1 timeout -k 10 200 python3 tools/quick.py 1920 1 3 0 sg || exit 1
done
