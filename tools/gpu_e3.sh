#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/e3; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
tools/ab_quick.sh e3ab bperm handout
for g in 13 14 16; do
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 20 2 0 gen$g | tail -1 || exit 1
done
