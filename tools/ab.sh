#!/bin/bash
# A/B of two builds of the library in one GPU session (device-to-device spread is ~10 %, so
# only numbers from one session compare):  tools/ab.sh <a.so> <b.so> [width spp reps strict scene]
A=$1; B=$2; W=${3:-1920}; S=${4:-1000}; R=${5:-2}; ST=${6:-0}; SC=${7:-sf}
for i in 1 2; do
  echo "== $A"; RTMI_LIB=$PWD/$A timeout -k 10 200 python3 tools/quick.py $W $S $R $ST $SC || exit 1
  echo "== $B"; RTMI_LIB=$PWD/$B timeout -k 10 200 python3 tools/quick.py $W $S $R $ST $SC || exit 1
done
