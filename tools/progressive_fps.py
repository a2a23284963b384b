"""Frames per second of progressive 1-spp accumulation with read-back into host memory
(the glass-heavy 1 spp/frame configuration, without the GLFW window): python tools/progressive_fps.py [width] [frames] [scene]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scene = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "sg"
sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % scene), dtype=rt.SPHERE_DTYPE)
rt.init(0)
rt.set_world(sph)
h = rt.height_for(w)
p = rt.default_params()
if scene == "sg":
    p.pos = rt.Vec3(9.0, 7.0, 3.0)
cam = rt.camera(w, h, p)
pin = "--no-pin" not in sys.argv
buf = rt.host_framebuffer(h, w) if pin else np.zeros((h, w, 4), np.uint8)   # the host keeps this framebuffer
print("framebuffer %s" % ("page-locked (rt_host_alloc): DMA read-back" if pin else "pageable"))
for spp in (1, 4):
    rt.render_progressive(cam, w, h, spp, 50, 0, out=buf)          # warm-up, allocations
    t0 = time.perf_counter()
    km = 0.0
    for f in range(frames):
        km += rt.render_progressive(cam, w, h, spp, 50, f, out=buf)["stats"].kernel_ms
    dt = time.perf_counter() - t0
    print("%s %dx%d progressive %d spp/frame, depth 50: %.1f frames/s with read-back (%.2f ms/frame, kernel %.2f ms), %.1f Msamples/s"
          % (scene, w, h, spp, frames / dt, dt / frames * 1e3, km / frames, w * h * spp * frames / dt / 1e6), flush=True)
if pin:
    rt.host_free(buf)
rt.shutdown()
