#!/bin/bash
# A/B of library builds on one GPU box, kernel time only: tools/ab_quick.sh <out-name> name1 name2 ... (tools/librtmi_<name>.so)
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for n in "$@"; do
  echo "== $n (round $rep)"
  RTMI_LIB=$R/tools/librtmi_$n.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 exact | tail -1 || exit 1
  if [ $rep -eq 1 ]; then
    RTMI_LIB=$R/tools/librtmi_$n.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 1 fast | tail -1 || exit 1
    RTMI_LIB=$R/tools/librtmi_$n.so RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1200 500 2 exact | tail -1 || exit 1
  fi
done; done
