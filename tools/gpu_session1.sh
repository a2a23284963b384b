#!/bin/bash
# first GPU session of the round: tests, the no-allocation entry point, bench, unit-size sweep, phase stamps
set -o pipefail
R=$PWD
O=$R/gpurun_out/s1
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > $O/noalloc.log 2>&1; echo "noalloc rc=$?" | tee -a $O/noalloc.log
tail -5 $O/noalloc.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
for c in 8 16 32 63; do
  timeout -k 10 200 python bench.py --no-extras --steps 2 --warmup 1 --chunk $c > $O/chunk_$c.json 2>> $O/bench.err || exit 1
  python -c "import json;d=json.load(open('$O/chunk_$c.json'));print('chunk',d['config']['chunk_spp'],d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 2 > $O/stamps.log 2>&1 || exit 1
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python tools/quick.py 1920 100 2 1 >> $O/stamps.log 2>&1 || exit 1
cat $O/stamps.log
