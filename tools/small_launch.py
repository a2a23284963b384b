"""Small launches: kernel time of a few short frames per build, with the depth cap lowered to show what the longest
paths cost at the end of a launch.  python tools/small_launch.py   (RTMI_LIB=tools/librtmi_stamps.so adds the phase shares
and the waves' mean lifetime on stderr)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
rt.init(0)
for scene, w, spp in (("sf", 1920, 8), ("sg", 1920, 1), ("sf", 1920, 1), ("sf", 1200, 16)):
    sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % scene), dtype=rt.SPHERE_DTYPE)
    rt.set_world(sph)
    p = rt.default_params()
    if scene == "sg":
        p.pos = rt.Vec3(9.0, 7.0, 3.0)
    h = rt.height_for(w); cam = rt.camera(w, h, p)
    buf = rt.host_framebuffer(h, w)
    for build in ("exact", "fast"):
        for depth in (50, 8):
            ks = []
            for f in range(5):
                st = rt.render(cam, w, h, spp, depth, build=build, seed=f, chunk_spp=rt.CHUNK_AUTO, out=buf)["stats"]
                ks.append(st.kernel_ms)
            scans = float(st.wave_scans)
            print("%s %dx%d %d spp depth %2d %s: kernel %.3f ms (min of 5), %.2f rays/sample, chunk %d, 18 flop x tests / vector peak %.3f, "
                  "%.0f wave scans = %.1f per wave of 4096" % (scene, w, h, spp, depth, build, min(ks), st.rays / st.samples, st.chunk_spp,
                  18.0 * st.sphere_tests / (min(ks) * 1e-3) / 157.3e12, scans, scans / 4096), flush=True)
    rt.host_free(buf)
rt.shutdown()
