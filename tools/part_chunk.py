"""Unit size against the cost of splitting a frame: kernel time of the whole 3840x2160 x 1000 spp frame and of one
eighth of it (2-row tiles, part 3 of 8) for several samples per unit.  python tools/part_chunk.py [build]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
build = sys.argv[1] if len(sys.argv) > 1 else "exact"
sph = np.fromfile(os.path.join(ROOT, 'tests', 'golden', 'scene_sf.bin'), dtype=rt.SPHERE_DTYPE)
rt.init(0); rt.set_world(sph)
W = 3840; H = rt.height_for(W); cam = rt.camera(W, H)
buf = np.zeros((H, W, 4), np.uint8)
for chunk in (8, 16, 24, 32, 40, 63):
    full = min(rt.render(cam, W, H, 1000, 50, build=build, chunk_spp=chunk, out=buf)['stats'].kernel_ms for _ in range(2))
    part = min(rt.render(cam, W, H, 1000, 50, build=build, chunk_spp=chunk, tile_rows=2, part_index=3, part_count=8, out=buf)['stats'].kernel_ms for _ in range(3))
    print("%s, %2d samples per unit: full frame %.1f ms, one eighth %.1f ms = %.3f of an eighth of the full frame; per-chunk sums %.2f GB" % (
        build, chunk, full, part, full / 8 / part, W * H * 16 * ((1000 + chunk - 1) // chunk) / 1e9), flush=True)
rt.shutdown()
