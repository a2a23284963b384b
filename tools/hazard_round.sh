#!/bin/bash
# Evidence about the matrix scan's tile loop under gpurun_out/<name>/ (kept in profiles/r3_hazard/):
#   probe_tile_loops.txt   the product against the rejected loop forms (tools/librtmi_<form>.so), one 3840x2160 frame three times each
#   mfma_exec_probe.txt    EXEC changes around matrix instructions in flight
#   mfma_valu_raw_probe.txt  a vector write directly ahead of the matrix instruction that reads the register
set -o pipefail
R=$PWD; N=${1:-r3_hazard}; O=$R/gpurun_out/$N; mkdir -p $O
SPP=${SPP:-250} bash tools/hazard_probe.sh $N/loops tools/librtmi_pipelined.so tools/librtmi_rolling.so tools/librtmi_rolling_drained.so tools/librtmi_rolling_serial.so > $O/probe_tile_loops.txt 2>&1 || { tail -5 $O/probe_tile_loops.txt; exit 1; }
timeout -k 10 120 ./tools/mfma_exec_probe > $O/mfma_exec_probe.txt 2>&1 || exit 1
timeout -k 10 120 ./tools/mfma_valu_raw_probe > $O/mfma_valu_raw_probe.txt 2>&1 || exit 1
cat $O/probe_tile_loops.txt $O/mfma_exec_probe.txt $O/mfma_valu_raw_probe.txt
