"""Renders one full-occupancy frame twice with the library named by RTMI_LIB and prints a digest of the f32
accumulators, the hit indices and the ray count of each run: python tools/hazard_probe.py [width] [spp] [build]
Used to compare forms of the matrix scan's tile loop (tools/hazard_probe.sh): a form that lets a load overwrite
the operands of a matrix instruction in flight gives digests that differ from the product's and from run to run."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt

w = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
build = sys.argv[3] if len(sys.argv) > 3 else "exact"
sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_sf.bin"), dtype=rt.SPHERE_DTYPE)
rt.init(0)
rt.set_world(sph)
h = rt.height_for(w)
cam = rt.camera(w, h)
keep = None
for rep in range(3):
    r = rt.render(cam, w, h, spp, 50, build=build, seed=17, chunk_spp=1, want_hit=True, want_accum=True)
    acc = r["accum"].view(np.uint32)
    if keep is None:
        keep = acc.copy()
    print("%s %s run %d: accum %s hit %s rays %d kernel %.2f ms; pixels differing from run 0: %d" % (
        os.path.basename(rt.LIB_PATH), build, rep, hashlib.sha256(acc.tobytes()).hexdigest()[:16],
        hashlib.sha256(r["hit"].tobytes()).hexdigest()[:16], int(r["stats"].rays), r["stats"].kernel_ms,
        int((acc != keep).any(axis=-1).sum())), flush=True)
rt.shutdown()
