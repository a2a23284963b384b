#!/bin/bash
# Sphere order on the device, A/B in one GPU session: tools/order_ab.sh <out-name>   (RTMI_ORDER=curve: Z-order curve only; default: tiles split in space)
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
{
for rep in 1 2; do
for ord in curve split; do
  for b in exact fast; do
    echo -n "$ord "; RTMI_ORDER=$ord RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 3840 250 2 $b | tail -1 || exit 1
  done
done; done
for ord in curve split; do
  echo -n "$ord "; RTMI_ORDER=$ord RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1200 500 2 exact | tail -1 || exit 1
  echo -n "$ord "; RTMI_ORDER=$ord RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 100 2 exact sg | tail -1 || exit 1
  for g in 16 32 64; do
    echo -n "$ord "; RTMI_ORDER=$ord RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 exact gen$g | tail -1 || exit 1
  done
done
} | tee $O/order_ab.txt
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 1 exact gen90 | tee -a $O/order_ab.txt
