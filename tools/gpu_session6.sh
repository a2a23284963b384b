#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
bash tools/ab_libs.sh ab3 tools/librtmi_prev.so raytracer_amd/librtmi.so
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 1 > $O/stamps.log 2>&1; cat $O/stamps.log
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 1 1 > $O/stamps_strict.log 2>&1; cat $O/stamps_strict.log
