#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > $O/noalloc.log 2>&1; echo "noalloc rc=$?" | tee -a $O/noalloc.log
tail -5 $O/noalloc.log
timeout -k 10 120 tools/mfma_scan_proto tests/golden/scene_sf.bin 8 > $O/proto.log 2>&1; echo "proto rc=$?"
cat $O/proto.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
