#!/bin/bash
# Collects the evidence kept under profiles/<name>/ on a GPU box:  tools/profile_round.sh <name>
#   bench.json                 python3 bench.py (default workload, with the CPU baseline leg)
#   bench_under_rocprof.json   the same command under rocprofv3 --kernel-trace --stats (no CPU leg)
#   bench_kernel_stats.csv     its per-kernel summary
#   pmc_workload.txt           the workload the counter passes ran (bench.py matches it before it reports traffic)
#   pmc_sources.txt            identifies the kernel sources the passes ran on (bench.py matches that too)
#   pmc_*.csv                  counter passes (each its own run) on bench.py --steps 1 --warmup 0 --no-cpu
# Everything is written under gpurun_out/<name>/ ; copy what is to be kept into profiles/<name>/.
set -o pipefail
N=${1:-round}
R=$PWD
O=$R/gpurun_out/$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/kt.err || exit 1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
python3 - > $O/pmc_workload.txt <<PY
import json
b = json.load(open("$O/bench.json"))
print("%s, %s build, chunk %d" % (b["config"]["workload"], b["config"]["kernel"], b["config"]["chunk_spp"]))
PY
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_sources_id())" > $O/pmc_sources.txt
pass() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-extras > $O/pmc_$name.json 2> $O/pmc_$name.err || return 1
  cp $(find $O/pmc_$name -name '*counter_collection.csv' | head -1) $O/pmc_$name.csv
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU || exit 1
pass b SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
pass c SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 || exit 1
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
rm -rf $O/kt $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_fetch $O/pmc_write
ls -la $O
