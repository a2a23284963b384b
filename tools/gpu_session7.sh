#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/s7
mkdir -p $O
cd $R
for i in 1 2 3; do timeout -k 10 100 python tools/quick.py 1920 50 1 1 | tail -1; done
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 $O/pytest.log
