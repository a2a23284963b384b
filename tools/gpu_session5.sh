#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s5
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for rep in 1 2; do
for lib in raytracer_amd/librtmi.so tools/librtmi_m1024.so; do
  n=$(basename $lib .so)
  RTMI_LIB=$R/$lib timeout -k 10 200 python bench.py --no-extras --steps 2 --warmup 1 > $O/$n.$rep.json 2>> $O/err.log || { echo "$n failed"; tail -3 $O/err.log; exit 1; }
  python -c "import json;d=json.load(open('$O/$n.$rep.json'));print('$n 4K x1000:',d['value'],'Msamples/s kernel',d['roofline']['kernel_ms'],'ms frac',d['roofline']['frac'])"
  RTMI_LIB=$R/$lib timeout -k 10 100 python tools/quick.py 1920 125 2 | tail -1
  RTMI_LIB=$R/$lib timeout -k 10 100 python tools/quick.py 1920 125 2 1 | tail -1
done; done
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 1 > $O/stamps.log 2>&1; cat $O/stamps.log
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 1 1 > $O/stamps_strict.log 2>&1; cat $O/stamps_strict.log
