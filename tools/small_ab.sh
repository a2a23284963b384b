#!/bin/bash
# small and large launches, exact and fast builds, for each library given: tools/small_ab.sh lib...
for lib in "$@"; do
  for b in exact fast; do
    echo -n "$(basename $lib .so) "; RTMI_LIB=$PWD/$lib RT_CHUNK=-1 python tools/quick.py 1920 8 5 $b | sort -t' ' -k7 -n | awk '{print}' | sort -k7,7n | head -1
    echo -n "$(basename $lib .so) "; RTMI_LIB=$PWD/$lib RT_CHUNK=-1 python tools/quick.py 1920 1 6 $b sg | sort -k7,7n | head -1
    echo -n "$(basename $lib .so) "; RTMI_LIB=$PWD/$lib RT_CHUNK=-1 python tools/quick.py 3840 250 2 $b | tail -1
  done
done
