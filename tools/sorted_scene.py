"""Does the order of the spheres on the device matter?  python tools/sorted_scene.py [order] [width] [spp]
order: lib (the library's own order) | orig | morton | shuffle | blocks | kd | kdm.  Every order but "lib" needs
RTMI_ORDER=caller in the environment (the script sets it).  The matrix scan tests 32 consecutive spheres per tile
and pays for appends per half tile that holds a candidate, so tiles that are compact in space should hold the
candidates of a wave's rays in fewer half tiles."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
order = sys.argv[1] if len(sys.argv) > 1 else "orig"
if order != "lib":
    os.environ["RTMI_ORDER"] = "caller"
import raytracer_amd as rt
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 125
sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_sf.bin"), dtype=rt.SPHERE_DTYPE)
def morton(ix, iz):
    m = 0
    for b in range(8):
        m |= ((ix >> b) & 1) << (2 * b) | ((iz >> b) & 1) << (2 * b + 1)
    return m
c = sph["center"].astype(np.float64)
def morton_order(idx):
    cc = c[idx]
    ix = np.clip(((cc[:, 0] + 12.0) * 8).astype(int), 0, 255); iz = np.clip(((cc[:, 2] + 12.0) * 8).astype(int), 0, 255)
    key = np.array([morton(int(a), int(b)) for a, b in zip(ix, iz)])
    return idx[np.argsort(key, kind="stable")]
def kd(idx, leaf_order):
    """Tile-aligned median splits along the longest axis: every leaf is one tile of 32."""
    m = len(idx)
    k = (m + 31) // 32
    if k <= 1:
        return leaf_order(idx)
    cc = c[idx]
    axis = int(np.argmax(cc.max(axis=0) - cc.min(axis=0)))
    s = idx[np.argsort(cc[:, axis], kind="stable")]
    left = 32 * (k // 2)
    return np.concatenate([kd(s[:left], leaf_order), kd(s[left:], leaf_order)])
small = np.array([i for i in range(len(sph)) if sph["radius"][i] < 100.0])
big = np.array([i for i in range(len(sph)) if sph["radius"][i] >= 100.0], dtype=int)
if order == "morton":
    sph = sph[morton_order(np.arange(len(sph)))]
elif order == "shuffle":
    sph = sph[np.random.default_rng(1).permutation(len(sph))]
elif order == "blocks":      # 4 x 4 blocks of the 22 x 22 grid, row-major inside a block
    bx = np.clip(((c[:, 0] + 11.0) / 5.5).astype(int), 0, 3); bz = np.clip(((c[:, 2] + 11.0) / 5.5).astype(int), 0, 3)
    sph = sph[np.lexsort((c[:, 2], c[:, 0], bz, bx))]
elif order == "kd":          # leaves ordered along their longest axis
    def along(idx):
        cc = c[idx]; axis = int(np.argmax(cc.max(axis=0) - cc.min(axis=0)))
        return idx[np.argsort(cc[:, axis], kind="stable")]
    sph = sph[np.concatenate([kd(small, along), big])]
elif order == "kdm":         # leaves in Z-order
    sph = sph[np.concatenate([kd(small, morton_order), big])]
rt.init(0)
rt.set_world(sph)
h = rt.height_for(w)
cam = rt.camera(w, h)
buf = np.zeros((h, w, 4), np.uint8)
for r in range(2):
    st = rt.render(cam, w, h, spp, 50, chunk_spp=-1, out=buf)['stats']
    print("%s: %dx%d spp %d: kernel %.2f ms  %.1f Msamples/s  rays/sample %.3f" % (
        order, w, h, spp, st.kernel_ms, st.samples / st.kernel_ms / 1e3, st.rays / st.samples), flush=True)
rt.shutdown()
