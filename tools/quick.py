"""Quick kernel timing: python tools/quick.py [width] [spp] [reps] [build: exact|strict|fast or 0|1|2] [scene]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
build = sys.argv[4] if len(sys.argv) > 4 else "exact"
build = int(build) if build.isdigit() else rt.BUILDS[build]
build_name = [k for k, v in rt.BUILDS.items() if v == build][0]
scene = sys.argv[5] if len(sys.argv) > 5 else "sf"
chunk = int(os.environ.get("RT_CHUNK", "50"))
if scene.startswith("gen"):        # genN: rt_scene_generate with half extent N
    sph = rt.scene_generate(int(scene[3:]), 0)
else:
    sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % scene), dtype=rt.SPHERE_DTYPE)
rt.init(0)
import time
t0 = time.perf_counter()
rt.set_world(sph)
print("set_world %d spheres: %.1f ms" % (len(sph), (time.perf_counter() - t0) * 1e3), flush=True)
h = rt.height_for(w)
cam = rt.camera(w, h)
buf = np.zeros((h, w, 4), np.uint8)
for r in range(reps):
    st = rt.render(cam, w, h, spp, 50, build=build, chunk_spp=chunk, out=buf)['stats']
    print("%s n=%d %dx%d spp %d: kernel %.2f ms  %.1f Msamples/s  frac %.4f  rays/sample %.3f" % (
        build_name, len(sph), w, h, spp, st.kernel_ms, st.samples / st.kernel_ms / 1e3,
        18.0 * st.sphere_tests / (st.kernel_ms * 1e-3) / 157.3e12, st.rays / st.samples) +
          ("  tiles/scan %.2f" % (st.tile_products / st.wave_scans) if st.wave_scans else ""), flush=True)
rt.shutdown()
