#!/bin/bash
set -o pipefail
R=$PWD
RTMI_LIB=$R/tools/librtmi_stamps.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 125 1 0 2>&1 | grep -v "^$" | tail -6 || exit 1
RTMI_LIB=$R/tools/librtmi_stamps.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 125 1 1 2>&1 | tail -6 || exit 1
for g in 12 13 14 16 22 32; do
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 0 gen$g | tail -1 || exit 1
done
