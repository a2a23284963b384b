#!/bin/bash
set -o pipefail
R=$PWD
bash tools/profile_round.sh r2a > gpurun_out/r2a_profile.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/r2a_profile.log; exit 1; }
cd $R
cat gpurun_out/r2a/bench.json
O=$R/gpurun_out/r2a
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 120 python tools/quick.py 1920 125 2 0 > $O/stamps_fast.log 2>&1 || exit 1
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 120 python tools/quick.py 1920 125 2 1 > $O/stamps_strict.log 2>&1 || exit 1
grep -h "phase\|share\|stamps" $O/stamps_fast.log $O/stamps_strict.log | tail -8
timeout -k 10 120 python tools/quick.py 1200 500 3 0 | tail -1
timeout -k 10 120 python tools/quick.py 1200 500 3 1 | tail -1
timeout -k 10 120 python tools/quick.py 1920 500 3 0 s3 | tail -1
