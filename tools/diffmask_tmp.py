import os, sys, subprocess, numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import raytracer_amd as rt
    sph = np.fromfile("tests/golden/scene_sf.bin", dtype=rt.SPHERE_DTYPE)
    rt.init(0); rt.set_world(sph)
    w = 1920; h = rt.height_for(w); cam = rt.camera(w, h)
    r = rt.render(cam, w, h, 1, 50, build="exact", seed=17, chunk_spp=1, want_hit=True, want_accum=True)
    np.savez(sys.argv[1], accum=r["accum"], hit=r["hit"], rays=int(r["stats"].rays))
    rt.shutdown(); sys.exit(0)
for name, lib in (("prod", "raytracer_amd/librtmi.so"), ("roll", "tools/librtmi_rolling_w3.so"), ("roll2", "tools/librtmi_rolling_w3.so")):
    subprocess.check_call([sys.executable, "/dev/stdin" if False else __file__, "/tmp/%s.npz" % name], env=dict(os.environ, RTMI_LIB=os.path.join(ROOT, lib)))
import raytracer_amd as rt
sph = np.fromfile("tests/golden/scene_sf.bin", dtype=rt.SPHERE_DTYPE)
a, b, c = (np.load("/tmp/%s.npz" % n) for n in ("prod", "roll", "roll2"))
for x, y, what in ((a, b, "product vs rolling"), (b, c, "rolling vs rolling")):
    d = (x["accum"].view("u4") != y["accum"].view("u4")).any(-1)
    print(what, "differing pixels", int(d.sum()), "of", d.size, "rays", int(x["rays"]), int(y["rays"]))
    hit = x["hit"][d]
    mt = np.where(hit >= 0, sph["mat_type"][np.maximum(hit, 0)], -1)
    print("  primary hit of differing pixels: sky %d ground %d lambert %d metal %d glass %d" % ((hit < 0).sum(), (hit == 0).sum(), ((mt == 0) & (hit > 0)).sum(), (mt == 1).sum(), (mt == 2).sum()))
    allh = x["hit"]; mta = np.where(allh >= 0, sph["mat_type"][np.maximum(allh, 0)], -1)
    print("  all pixels: sky %d ground %d lambert %d metal %d glass %d" % ((allh < 0).sum(), (allh == 0).sum(), ((mta == 0) & (allh > 0)).sum(), (mta == 1).sum(), (mta == 2).sum()))
    ys, xs = np.nonzero(d)
    if len(ys):
        print("  rows of differing pixels: min %d max %d; first ten (y, x):" % (ys.min(), ys.max()), list(zip(ys[:10].tolist(), xs[:10].tolist())))
        k = 0
        print("  example accum product/rolling:", x["accum"][ys[k], xs[k]], y["accum"][ys[k], xs[k]])
