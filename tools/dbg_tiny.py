import os, sys, subprocess, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
def run():
    import raytracer_amd as rt, stress_scenes as ss
    rt.init(0)
    out = {}
    for kind in ("tiny", "wide"):
        sph = ss.random_scene(kind); rt.set_world(sph)
        W, H = 320, 180
        cam = rt.camera(W, H, ss.view_params(rt, kind, 0.0))
        for depth in (1, 12):
            r = rt.render(cam, W, H, 1, depth, mode=rt.MODE_CONST_U, strict_fp=True, seed=7, chunk_spp=1, want_hit=True, want_accum=True)
            out["%s_%d" % (kind, depth)] = (r["hit"].copy(), r["accum"].copy())
    rt.shutdown()
    return out
if len(sys.argv) > 1:
    import pickle
    pickle.dump(run(), open(sys.argv[1], "wb"))
else:
    import pickle
    subprocess.check_call([sys.executable, __file__, "/tmp/m.pkl"])
    subprocess.check_call([sys.executable, __file__, "/tmp/v.pkl"], env=dict(os.environ, RTMI_SCAN="vector"))
    m = pickle.load(open("/tmp/m.pkl", "rb")); v = pickle.load(open("/tmp/v.pkl", "rb"))
    for k in m:
        dh = np.argwhere(m[k][0] != v[k][0]); da = np.argwhere((m[k][1].view(np.uint32) != v[k][1].view(np.uint32)).any(axis=-1))
        print(k, "hit diffs", len(dh), "accum diffs", len(da))
        for y, x in list(da[:12]):
            pix = y * 320 + x
            print("   pixel", x, y, "unit", pix, "lane", pix % 64, "pool", pix // 64, "hit matrix", m[k][0][y, x], "vector", v[k][0][y, x])
