#!/bin/bash
# Pool size of the unit hand-out: the library's own choice against fixed sizes (RTMI_POOL)
set -o pipefail
for pu in auto 64 256 512; do
  echo "== pool $pu"
  if [ $pu = auto ]; then unset RTMI_POOL; else export RTMI_POOL=$pu; fi
  timeout -k 10 100 python tools/progressive_fps.py 1920 200 2>&1 | grep "spp/frame" || exit 1
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 8 3 0 | tail -1 || exit 1
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 125 2 0 | tail -1 || exit 1
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1200 500 2 0 | tail -1 || exit 1
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 0 | tail -1 || exit 1
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 1 1 | tail -1 || exit 1
done
unset RTMI_POOL
timeout -k 10 300 python tools/part_scaling.py
