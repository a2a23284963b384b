#!/bin/bash
# Small launches (where the end of a launch weighs): tools/ab_small.sh name1 name2 ... (tools/librtmi_<name>.so)
set -o pipefail
R=$PWD
for rep in 1 2; do
for n in "$@"; do
  echo "== $n (round $rep)"
  RTMI_LIB=$R/tools/librtmi_$n.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 8 3 0 | tail -1 || exit 1
  RTMI_LIB=$R/tools/librtmi_$n.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 125 2 0 | tail -1 || exit 1
  RTMI_LIB=$R/tools/librtmi_$n.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 0 | tail -1 || exit 1
  RTMI_LIB=$R/tools/librtmi_$n.so timeout -k 10 100 python tools/progressive_fps.py 2>&1 | grep "1 spp/frame" || exit 1
done; done
