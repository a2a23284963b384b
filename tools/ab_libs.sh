#!/bin/bash
# A/B of library builds on one GPU box: tools/ab_libs.sh <out-name> lib1.so lib2.so ...
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so)
  RTMI_LIB=$R/$lib timeout -k 10 200 python bench.py --no-extras --steps 2 --warmup 1 > $O/$n.$rep.json 2>> $O/err.log || { echo "$n failed"; tail -3 $O/err.log; exit 1; }
  python -c "import json;d=json.load(open('$O/$n.$rep.json'));print('$n 4K x1000:',d['value'],'Msamples/s kernel',d['roofline']['kernel_ms'],'ms frac',d['roofline']['frac'])"
  RTMI_LIB=$R/$lib timeout -k 10 100 python tools/quick.py 1920 125 2 | tail -1
  RTMI_LIB=$R/$lib timeout -k 10 100 python tools/quick.py 1920 125 2 1 | tail -1
done; done
