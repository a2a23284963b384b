#!/bin/bash
# instruction counters of the product library under environment switches, 1920x1080 x 50 spp, exact build, final scene:
#   tools/pmc_env_ab.sh <out-name>      (RTMI_NO_CULL=1: every tile multiplied; RTMI_ORDER=curve: tiles along the Z-order curve)
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default RTMI_NO_CULL=1 RTMI_ORDER=curve; do
  if [ $v != default ]; then export $v; fi
  RT_CHUNK=-1 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $O/$v -o pmc --output-format csv -- python3 $R/tools/quick.py 1920 50 1 exact > $O/$v.log 2>&1 || { tail -5 $O/$v.log; exit 1; }
  if [ $v != default ]; then unset ${v%%=*}; fi
  echo "$v: $(tail -1 $O/$v.log)"
  python3 - $O/$v <<'PY'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("  " + "  ".join("%s %.5g" % (k, v) for k, v in sorted(tot.items())))
PY
done | tee $O/pmc_env_ab.txt
