#!/bin/bash
# Counter passes (each its own run) on tools/quick.py for one library build:
#   tools/pmc_quick.sh <out-name> <lib.so> [width spp]
set -o pipefail
R=$PWD; N=$1; LIB=$R/$2; W=${3:-1920}; SPP=${4:-125}
O=$R/gpurun_out/$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RTMI_LIB=$LIB
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python3 $R/tools/quick.py $W $SPP 1 > $O/pmc_$name.log 2> $O/pmc_$name.err || return 1
  cp $(find $O/pmc_$name -name '*counter_collection.csv' | head -1) $O/pmc_$name.csv && rm -rf $O/pmc_$name
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU || exit 1
pass b SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
pass c SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY || exit 1
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$O/pmc_*.csv")):
    tot={}
    for r in csv.DictReader(open(f)):
        if 'trace_kernel' in r['Kernel_Name']:
            tot[r['Counter_Name']]=tot.get(r['Counter_Name'],0)+float(r['Counter_Value'])
    for k,v in tot.items(): print("%-32s %.4e"%(k,v))
PY
