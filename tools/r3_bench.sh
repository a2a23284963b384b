#!/bin/bash
# GPU tests, then the default bench line:  bash tools/r3_bench.sh <out-name>
set -o pipefail
R=$PWD; O=$R/gpurun_out/${1:-r3b}; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
