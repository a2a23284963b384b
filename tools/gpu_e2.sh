#!/bin/bash
set -o pipefail
R=$PWD
for rep in 1 2; do
for o in lib orig morton kd kdm shuffle; do
  timeout -k 10 100 python tools/sorted_scene.py $o 3840 250 | tail -1 || exit 1
done; done
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "every_scan_variant" 2>&1 | tail -3
