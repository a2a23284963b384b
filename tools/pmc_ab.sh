#!/bin/bash
# instruction counters of library variants on 1920x1080 x 50 spp, exact build:  tools/pmc_ab.sh <out-name> lib1.so lib2.so ...
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  RTMI_LIB=$R/$lib RT_CHUNK=-1 timeout -s KILL 120 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD} -d $O/$n -o pmc --output-format csv -- python3 $R/tools/quick.py 1920 50 1 ${BUILD:-exact} > $O/$n.log 2>&1 || { tail -5 $O/$n.log; exit 1; }
  tail -1 $O/$n.log
  python3 - $O/$n <<'PY'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("  ".join("%s %.4g" % (k, v) for k, v in sorted(tot.items())))
PY
done
