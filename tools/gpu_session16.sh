#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s16
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for g in 16 22 32 45 64; do
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 0 gen$g | tail -1 || exit 1
  RTMI_SCAN=vector RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 0 gen$g | tail -1 | sed 's/^/   vector: /' || exit 1
done
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 1 gen32 | tail -1 || exit 1
RTMI_SCAN=vector RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 1 gen32 | tail -1 | sed 's/^/   vector: /' || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 0 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 1 | tail -1 || exit 1
