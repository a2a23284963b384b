#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s14
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 0 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 1 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 1000 2 0 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1200 500 2 0 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 8 3 0 | tail -1 || exit 1
timeout -k 10 300 python tools/part_scaling.py 2>&1 | grep -v amdgpu || exit 1
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 300 python tools/part_stamps.py 2>&1 | grep "==\|per wave scan: pair" | cut -c1-200 | awk '/==/ || !seen[$0]++'
