#!/bin/bash
# tools/gpu_ab.sh name1 name2 ...: the GPU tests on the product library, then an A/B of tools/librtmi_<name>.so builds
set -o pipefail
R=$PWD; O=$R/gpurun_out/ab; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
tools/ab_quick.sh abq "$@"
