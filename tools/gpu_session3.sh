#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s3
mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/noalloc_alone.log 2>&1; echo "noalloc alone rc=$?"
tail -3 $O/noalloc_alone.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest.log
timeout -k 10 300 tools/mfma_scan_proto tests/golden/scene_sf.bin 150 > $O/proto.log 2>&1; echo "proto rc=$?"
cat $O/proto.log
