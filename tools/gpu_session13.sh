#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r2b
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
{
echo "# 1  3-sphere scene 400x225, 100 spp, depth 50"
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 400 100 3 0 s3 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 400 100 3 1 s3 | tail -1 || exit 1
echo "# 2  3-sphere scene 1920x1080, 500 spp, depth 50"
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 500 3 0 s3 | tail -1 || exit 1
RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 500 3 1 s3 | tail -1 || exit 1
} > $O/configs_small.txt 2>&1 || { cat $O/configs_small.txt; exit 1; }
cat $O/configs_small.txt
