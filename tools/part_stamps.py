"""Where an eighth of the 4K frame loses against the full frame: run with RTMI_LIB=tools/librtmi_stamps.so;
the library prints wave scans, cycles per wave scan and wave lifetimes per launch (stderr)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_sf.bin"), dtype=rt.SPHERE_DTYPE)
rt.init(0); rt.set_world(sph)
W = 3840; H = rt.height_for(W); cam = rt.camera(W, H)
buf = np.zeros((H, W, 4), np.uint8)
for n, spp in ((1, 250), (8, 250), (8, 1000), (1, 1000)):
    for rep in range(2):
        st = rt.render(cam, W, H, spp, 50, chunk_spp=rt.CHUNK_AUTO, tile_rows=2, part_index=0, part_count=n, out=buf)['stats']
    sys.stderr.flush()
    print("== parts %d spp %d: kernel %.2f ms, %.1f Msamples/s, rays %d" % (n, spp, st.kernel_ms, st.samples / st.kernel_ms / 1e3, st.rays), file=sys.stderr, flush=True)
