"""Fixed cost of a launch: one eighth of the 4K frame (2-row tiles, part 3 of 8) against spp and unit size."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
sph = np.fromfile(os.path.join(ROOT, 'tests', 'golden', 'scene_sf.bin'), dtype=rt.SPHERE_DTYPE)
rt.init(0); rt.set_world(sph)
W = 3840; H = rt.height_for(W); cam = rt.camera(W, H)
buf = np.zeros((H, W, 4), np.uint8)
def t(spp, chunk):
    best = 1e9
    for rep in range(3):
        st = rt.render(cam, W, H, spp, 50, chunk_spp=chunk, tile_rows=2, part_index=3, part_count=8, out=buf)['stats']
        best = min(best, st.kernel_ms)
    return best
for chunk in (8, 4, 2, 1):
    a = [(spp, t(spp, chunk)) for spp in (125, 250, 500, 1000)]
    slope = (a[3][1] - a[1][1]) / 750.0
    print("chunk %d: %s  slope %.4f ms/spp  intercept %.2f ms" % (chunk, ["%d:%.1f" % x for x in a], slope, a[3][1] - slope * 1000))
