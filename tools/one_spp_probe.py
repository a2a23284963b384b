"""Where a 1-spp frame's time goes: kernel time of the glass scene at 1920x1080, 1 sample per pixel, for several depth caps
(the longest paths set the end of the launch) and image sizes (python tools/one_spp_probe.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_sg.bin"), dtype=rt.SPHERE_DTYPE)
rt.init(0); rt.set_world(sph)
p = rt.default_params(); p.pos = rt.Vec3(9.0, 7.0, 3.0)
for w in (1920, 3840):
    h = rt.height_for(w); cam = rt.camera(w, h, p)
    buf = rt.host_framebuffer(h, w)
    for depth in (50, 12, 6, 3, 1):
        ks = []
        for f in range(6):
            st = rt.render(cam, w, h, 1, depth, seed=f, chunk_spp=1, out=buf)["stats"]
            ks.append(st.kernel_ms)
        print("%dx%d 1 spp depth %2d: kernel %.3f ms (min of 6), %.2f rays/sample, %.1f Mrays/s" % (
            w, h, depth, min(ks), st.rays / st.samples, st.rays / min(ks) / 1e3), flush=True)
    rt.host_free(buf)
rt.shutdown()
