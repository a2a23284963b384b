"""Worker of test_gpu_parity.py::test_every_scan_variant_gives_the_same_image: renders a few scenes with
every build and writes hit indices, accumulators and ray counts to an .npz file.  The test
runs it once as is, once with RTMI_SCAN=vector (packed-FP32 scan for every scene size), with RTMI_ORDER=caller / curve
(other orders of the spheres on the device) and with RTMI_NO_CULL / RTMI_NO_BOX_GROUPS (the box test ahead of the matrix
products switched off, or asked tile by tile)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import raytracer_amd as rt
    from test_gpu_parity import _tie_scene
    rt.init(0)
    out = {}
    p = rt.default_params()
    p.pos = rt.Vec3(9.0, 1.0, 2.0)
    p.target = rt.Vec3(0.0, 0.0, 0.0)
    p.vfov = 68.4
    W, H = 192, 108
    cam = rt.camera(W, H, p)
    for n, seed in ((40, 1), (600, 2), (1300, 3), (5000, 4), (9000, 5)):
        rt.set_world(_tie_scene(n, seed))
        for build in ("exact", "strict", "fast"):
            for rep in range(2):                       # twice: the image must not change from run to run
                r = rt.render(cam, W, H, 4, 20, build=build, seed=5, chunk_spp=2, want_hit=True, want_accum=True)
                key = "n%d_%s_%d" % (n, build, rep)
                out[key + "_hit"] = r["hit"].copy()
                out[key + "_accum"] = r["accum"].copy()
                out[key + "_rays"] = np.array([int(r["stats"].rays)], dtype=np.int64)
    rt.shutdown()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(sys.argv[1])
