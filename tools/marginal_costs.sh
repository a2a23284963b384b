#!/bin/bash
# What one more unit of each part of the scan costs: library variants that run a part twice (same results) or add
# padding instructions, timed against the product on 4K x 250 spp, exact build.
#   build here:   bash tools/marginal_costs.sh build
#   on the GPU:   bash tools/marginal_costs.sh run <out-name>
set -o pipefail
V="xt1:-DRT_X_DOUBLE_TILE=1 xt2:-DRT_X_DOUBLE_TILE=2 xt3:-DRT_X_DOUBLE_TILE=3 xd:-DRT_X_DOUBLE_DRAIN xa:-DRT_X_DOUBLE_APPEND"
if [ "$1" = build ]; then
  for v in $V; do make -s -C raytracer_amd/csrc variant NAME=${v%%:*} DEF="${v#*:}" || exit 1; done
  make -s -C raytracer_amd/csrc variant NAME=pad0 DEF="-DRT_X_PAD_VALU=40 -DRT_X_PAD_KIND=0" || exit 1
  make -s -C raytracer_amd/csrc variant NAME=pad1 DEF="-DRT_X_PAD_VALU=40 -DRT_X_PAD_KIND=1" || exit 1
  make -s -C raytracer_amd/csrc variant NAME=pad2 DEF="-DRT_X_PAD_VALU=40 -DRT_X_PAD_KIND=2" || exit 1
  make -s -C raytracer_amd/csrc variant NAME=t768 DEF="" MATTHREADS=768 || exit 1
  make -s -C raytracer_amd/csrc variant NAME=t512 DEF="" MATTHREADS=512 || exit 1
else
  L="raytracer_amd/librtmi.so"
  for n in xt1 xt2 xt3 xd xa pad0 pad1 pad2 t768 t512; do L="$L tools/librtmi_$n.so"; done
  bash tools/occ_ab.sh ${2:-marginal} $L
fi
