#!/bin/bash
# GPU tests on the product library, instruction counters and kernel times against other libraries:  tools/tpo.sh <out-name> lib...
set -o pipefail
R=$PWD; N=$1; shift; O=$R/gpurun_out/$N; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
bash tools/pmc_ab.sh $N "$@" && bash tools/occ_ab.sh $N "$@"
