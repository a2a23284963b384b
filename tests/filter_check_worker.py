"""Worker of test_gpu_parity.py::test_filter_and_drain_equal_brute_force.  Run with RTMI_LIB pointing
at the diagnostic library (make -C raytracer_amd/csrc diag): its kernels also run the exact test over
ALL spheres for every ray and count the rays for which the filter + queue + drain (or the shared-ray
search) found another nearest hit.  Prints one JSON object per case."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import raytracer_amd as rt
    import stress_scenes as ss
    assert "diag" in os.path.basename(rt.LIB_PATH), "RTMI_LIB must name the diagnostic library"
    L = rt.lib()
    L.rt_diag_counters.restype = None
    L.rt_diag_counters.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    rt.init(0)
    cases = [("sf", None)] + [(k, k) for k in ss.VIEWS] + [("gen127", "gen"), ("gen130", "gen")]
    for name, kind in cases:
        if kind == "gen":
            # the generated scene at both ends of the switch from the matrix scan (at most 65504 spheres, image in
            # global memory) to the packed-FP32 scan: 64517 and 67601 spheres
            sph = rt.scene_generate(int(name[3:]), 0)
            W, H, spp, depth = 192, 108, 2, 8
            cam = rt.camera(W, H)
            kind = None
        elif kind is None:
            sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_sf.bin"), dtype=rt.SPHERE_DTYPE)
            W, H, spp, depth = 400, 225, 8, 50
            cam = rt.camera(W, H)
        else:
            sph = ss.random_scene(kind)
            W, H, spp, depth = 320, 180, 6, 12
        rt.set_world(sph)
        for build in ("exact", "strict", "fast"):
            for mode in (rt.MODE_MC, rt.MODE_CONST_U):
                if kind is not None:
                    cam = rt.camera(W, H, ss.view_params(rt, kind, defocus_angle=0.6 if mode == rt.MODE_MC else 0.0))
                r = rt.render(cam, W, H, spp if mode == rt.MODE_MC else 1, depth, mode=mode, build=build,
                              seed=7, chunk_spp=2, want_hit=True)
                c = (C.c_ulonglong * 16)()
                L.rt_diag_counters(c, 16)
                print(json.dumps({"scene": name, "build": build, "mode": mode, "rays": int(c[1]), "wave_scans": int(c[5]),
                                  "queued": int(c[3]), "lost_hits": int(c[8]),
                                  "seen": float((r["hit"] >= 0).mean())}), flush=True)
    rt.shutdown()


if __name__ == "__main__":
    main()
