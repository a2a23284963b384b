#!/bin/bash
# Long runs of the diagnostic library (tools/librtmi_diag.so: every ray also takes the exact test over ALL spheres in index order; the
# line "rays whose nearest hit differs" must say 0):  tools/brute_soak.sh <out-name>
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
for cfg in "3840 100 exact sf" "3840 60 strict sf" "3840 100 fast sf" "3840 60 exact sg" "1920 100 exact gen32" "1920 60 exact gen64"; do
  set -- $cfg
  RTMI_LIB=$R/tools/librtmi_diag.so RT_CHUNK=-1 timeout -k 10 280 python tools/quick.py $1 $2 1 $3 $4 > $O/soak_$3_$4_$1.log 2>&1 || { tail -3 $O/soak_$3_$4_$1.log; exit 1; }
  echo "$cfg: $(grep 'rays whose nearest hit differs' $O/soak_$3_$4_$1.log | sed 's/.*rays traced in shared-ray mode [0-9]* of \([0-9]*\); rays whose nearest hit differs from the all-spheres exact test: \([0-9]*\).*/\1 rays, \2 differ/')"
done | tee $O/brute_soak.txt
