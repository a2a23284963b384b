#!/bin/bash
# First GPU session of a change: the GPU tests, then kernel times of the three builds and the phase stamps.
set -o pipefail
R=$PWD; O=$R/gpurun_out/${1:-r3a}; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for b in exact strict fast; do
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 $b | tail -1 || exit 1
done
for b in exact strict fast; do
  RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 120 python tools/quick.py 1920 125 2 $b > $O/stamps_$b.log 2>&1 || exit 1
  grep "wave-cycles per phase" $O/stamps_$b.log | tail -1
done
