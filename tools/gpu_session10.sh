#!/bin/bash
set -o pipefail
for i in 1 2; do timeout -k 10 100 python tools/quick.py 3840 250 2 0 | tail -1 || exit 1; done
timeout -k 10 100 python tools/quick.py 3840 250 2 1 | tail -1 || exit 1
timeout -k 10 200 python tools/dbg_sf.py 2>&1 | grep diffs || exit 1
