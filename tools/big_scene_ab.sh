#!/bin/bash
# Large generated scenes with and without the groups of the box hierarchy, one GPU session: tools/big_scene_ab.sh <out-name>
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
{
for g in 22 32 45 64 90; do
  echo -n "groups   "; RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 exact gen$g | tail -1 || exit 1
  echo -n "tiles    "; RTMI_NO_BOX_GROUPS=1 RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 exact gen$g | tail -1 || exit 1
done
} | tee $O/big_scene_ab.txt
