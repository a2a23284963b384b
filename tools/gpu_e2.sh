#!/bin/bash
set -o pipefail
R=$PWD
for o in orig morton blocks shuffle; do
  timeout -k 10 100 python tools/sorted_scene.py $o 3840 250 | tail -1 || exit 1
  RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 100 python tools/sorted_scene.py $o 1920 60 2>&1 | grep "per wave scan" | tail -1 || exit 1
done
