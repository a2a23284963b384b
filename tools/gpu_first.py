"""First-light check on the GPU box: strict parity vs the CPU checker and rough timings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from oracle import refcpu as rc

GOLD = os.path.join(ROOT, "tests", "golden")
rt.init(0)
for scene in ("s3", "sf"):
    sph = np.fromfile(os.path.join(GOLD, "scene_%s.bin" % scene), dtype=rc.SPHERE_DTYPE)
    rt.set_world(sph)
    W, H = 400, 225
    cam = rt.camera(W, H)
    ccam = rc.camera(W, H)
    print(scene, "camera host==checker:", np.array_equal(rt.camera_as_array(cam), rc.camera_as_array(ccam)))
    g = np.load(os.path.join(GOLD, "d1_%s_400.npz" % scene))
    r = rt.render(cam, W, H, 1, 50, mode=rt.MODE_CONST_U, strict_fp=True, want_hit=True, want_accum=True)
    print(scene, "D1 strict: hit eq", np.array_equal(r["hit"], g["hit"]), "accum bit-eq",
          np.array_equal(r["accum"].view("u4"), g["accum"].view("u4")), "rgba eq", np.array_equal(r["rgba"], g["rgba"]),
          "ms", r["stats"].kernel_ms)
    r = rt.render(cam, W, H, 1, 50, mode=rt.MODE_CONST_U, strict_fp=False, want_hit=True, want_accum=True)
    rel = np.abs(r["accum"] - g["accum"]) / np.maximum(np.abs(g["accum"]), 1e-6)
    print(scene, "D1 fast: hit mismatches", int((r["hit"] != g["hit"]).sum()), "max rel colour err (hit-equal px)",
          float(rel[(r["hit"] == g["hit"])].max()), "rgba diff px", int((r["rgba"] != g["rgba"]).any(-1).sum()))
    spp = 8
    c = rc.render(ccam, sph, W, H, spp, 50, rng=rc.RNG_COUNTER, seed=7, chunk_spp=3, want_accum=True, want_hit=True)
    r = rt.render(cam, W, H, spp, 50, strict_fp=True, seed=7, chunk_spp=3, want_accum=True, want_hit=True)
    print(scene, "MC strict vs checker: accum bit-eq", np.array_equal(r["accum"].view("u4"), c["accum"].view("u4")),
          "rgba eq", np.array_equal(r["rgba"], c["rgba"]), "hit eq", np.array_equal(r["hit"], c["hit"]),
          "rays", r["stats"].rays, c["stats"].rays, "diff px", int((r["rgba"] != c["rgba"]).any(-1).sum()))
    for strict in (True, False):
        for (w, spp, chunk) in ((1200, 64, 16), (1920, 32, 8)):
            h = rt.height_for(w)
            cam2 = rt.camera(w, h)
            r = rt.render(cam2, w, h, spp, 50, strict_fp=strict, chunk_spp=chunk)
            st = r["stats"]
            n = len(sph)
            print(scene, "strict" if strict else "fast", w, "x", h, "spp", spp, "kernel_ms %.1f" % st.kernel_ms,
                  "Msamples/s %.1f" % (st.samples / st.kernel_ms / 1e3), "rays/sample %.3f" % (st.rays / st.samples),
                  "Gtests/s %.1f" % (st.sphere_tests / st.kernel_ms / 1e6),
                  "roofline frac %.3f" % (18.0 * st.sphere_tests / (st.kernel_ms * 1e-3) / 157.3e12))
rt.shutdown()
