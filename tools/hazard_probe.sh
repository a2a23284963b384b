#!/bin/bash
# The product library against variant builds of the tile loop (tools/librtmi_<name>.so), same frame, three runs each.
set -o pipefail
R=$PWD; O=$R/gpurun_out/${1:-hazard}; shift; mkdir -p $O
: > $O/probe.txt
for lib in raytracer_amd/librtmi.so "$@"; do
  for b in exact fast; do
    RTMI_LIB=$R/$lib timeout -k 10 120 python tools/hazard_probe.py 3840 ${SPP:-4} $b >> $O/probe.txt 2>> $O/err.log || { echo "$lib failed"; tail -3 $O/err.log; exit 1; }
  done
done
cat $O/probe.txt
