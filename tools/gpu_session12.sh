#!/bin/bash
set -o pipefail
R=$PWD
for g in 2 3 4 5 6 8; do
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 100 2 0 gen$g | tail -1 || exit 1
  RTMI_SCAN=vector RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 100 2 0 gen$g | tail -1 | sed 's/^/   vector: /' || exit 1
done
for g in 3 5 8; do
  RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 100 2 1 gen$g | tail -1 || exit 1
  RTMI_SCAN=vector RT_CHUNK=-1 timeout -k 10 100 python tools/quick.py 1920 100 2 1 gen$g | tail -1 | sed 's/^/   vector: /' || exit 1
done
cat > /tmp/part_one.py <<PY
import os, sys, numpy as np
sys.path.insert(0, "$R")
import raytracer_amd as rt
sph = np.fromfile("$R/tests/golden/scene_sf.bin", dtype=rt.SPHERE_DTYPE)
rt.init(0); rt.set_world(sph)
W=3840; H=rt.height_for(W); cam=rt.camera(W,H)
buf=np.zeros((H,W,4),np.uint8)
for n in (1, 8):
    for rep in range(2):
        st = rt.render(cam,W,H,250,50,chunk_spp=rt.CHUNK_AUTO,tile_rows=2,part_index=0,part_count=n,out=buf)['stats']
    print("parts", n, "kernel ms", st.kernel_ms, flush=True)
PY
RTMI_LIB=$R/tools/librtmi_stamps.so timeout -k 10 200 python /tmp/part_one.py 2>&1 | grep -v amdgpu | grep "parts\|lived"
