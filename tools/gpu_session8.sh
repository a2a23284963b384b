#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s8
mkdir -p $O
cd $R
timeout -k 10 200 python tools/dbg_tiny.py > $O/dbg.log 2>&1; echo "dbg rc=$?"; grep -E "diffs" $O/dbg.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -12 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
