import os, sys, subprocess, pickle
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
def run():
    import raytracer_amd as rt
    rt.init(0)
    out = {}
    sph = np.fromfile("tests/golden/scene_s3.bin", dtype=rt.SPHERE_DTYPE); rt.set_world(sph)
    W, H = 400, 225
    cam = rt.camera(W, H)
    for rep in range(2):
        for depth in (1, 2, 3, 50):
            r = rt.render(cam, W, H, 1, depth, mode=rt.MODE_CONST_U, strict_fp=True, seed=7, chunk_spp=1, want_hit=True, want_accum=True)
            out["s3_%d_%d" % (depth, rep)] = (r["hit"].copy(), r["accum"].copy(), int(r["stats"].rays))
    rt.shutdown()
    return out
if len(sys.argv) > 1:
    pickle.dump(run(), open(sys.argv[1], "wb"))
else:
    subprocess.check_call([sys.executable, __file__, "/tmp/m.pkl"])
    subprocess.check_call([sys.executable, __file__, "/tmp/v.pkl"], env=dict(os.environ, RTMI_SCAN="vector"))
    m = pickle.load(open("/tmp/m.pkl", "rb")); v = pickle.load(open("/tmp/v.pkl", "rb"))
    for k in m:
        dh = np.argwhere(m[k][0] != v[k][0]); da = np.argwhere((m[k][1].view(np.uint32) != v[k][1].view(np.uint32)).any(axis=-1))
        print(k, "hit diffs", len(dh), "accum diffs", len(da), "rays", m[k][2], v[k][2])
        units = da[:, 0] * 400 + da[:, 1]
        print("   lanes", np.bincount(units % 64, minlength=64).tolist())
        print("   matrix zero", int((m[k][1][da[:, 0], da[:, 1]] == 0).all(axis=-1).sum()), "vector zero", int((v[k][1][da[:, 0], da[:, 1]] == 0).all(axis=-1).sum()))
        for y, x in []:
            print("   pixel", x, y, "unit", y * 400 + x, "hit", m[k][0][y, x], v[k][0][y, x], "accum", m[k][1][y, x], v[k][1][y, x])
