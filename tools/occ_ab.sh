#!/bin/bash
# kernel time of library variants on 4K x 250 spp, exact build:  tools/occ_ab.sh <out-name> lib1.so lib2.so ...
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$(basename $lib .so) "; RTMI_LIB=$R/$lib RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 exact | tail -1 || exit 1
done; done 2>&1 | tee $O/occ.txt
