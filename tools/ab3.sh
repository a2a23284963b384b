#!/bin/bash
# A/B of library builds, the three kernel builds each: tools/ab3.sh <out-name> lib1.so lib2.so ...   (kernel ms, 4K x 250 spp and config 3)
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  for b in exact strict fast; do
    echo -n "$(basename $lib .so) "; RTMI_LIB=$R/$lib RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 $b | tail -1 || exit 1
  done
  if [ $rep -eq 1 ]; then echo -n "$(basename $lib .so) "; RTMI_LIB=$R/$lib RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1200 500 2 exact | tail -1 || exit 1; fi
done; done
