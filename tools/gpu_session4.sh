#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
timeout -k 10 200 python tools/quick.py 1920 100 3 > $O/quick.log 2>&1; echo "quick rc=$?"; cat $O/quick.log
RTMI_SCAN=vector timeout -k 10 200 python tools/quick.py 1920 100 3 > $O/quick_vec.log 2>&1; echo "quick vec rc=$?"; cat $O/quick_vec.log
timeout -k 10 200 python tools/quick.py 1920 100 3 1 > $O/quick_strict.log 2>&1; echo "quick strict rc=$?"; cat $O/quick_strict.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
timeout -k 10 300 python bench.py --no-extras --steps 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 1 > $O/stamps.log 2>&1; cat $O/stamps.log
