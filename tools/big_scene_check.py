"""Scenes at the upper end of the matrix scan and beyond it, with the diagnostic library (RTMI_LIB=tools/librtmi_diag.so):
every ray's nearest hit against the exact test over all spheres.  python tools/big_scene_check.py [half extents ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
L = rt.lib()
L.rt_diag_counters.restype = None
L.rt_diag_counters.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
rt.init(0)
for g in [int(a) for a in sys.argv[1:]] or [127, 130]:
    sph = rt.scene_generate(g, 0)
    rt.set_world(sph)
    W, H = 256, 144
    cam = rt.camera(W, H)
    for strict in (False, True):
        r = rt.render(cam, W, H, 2, 8, strict_fp=strict, seed=3, chunk_spp=1, want_hit=True)
        c = (C.c_ulonglong * 16)()
        L.rt_diag_counters(c, 16)
        print("half extent %d: %d spheres, %s: rays %d, wave scans %d, lost hits %d, kernel %.1f ms, pixels that see a sphere %.2f"
              % (g, len(sph), "strict" if strict else "fast", c[1], c[5], c[8], r["stats"].kernel_ms, (r["hit"] >= 0).mean()), flush=True)
rt.shutdown()
