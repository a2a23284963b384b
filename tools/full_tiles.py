import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import raytracer_amd as rt
n = int(sys.argv[1])
r = np.random.default_rng(77)
sph = np.zeros(n, dtype=rt.SPHERE_DTYPE)
sph["center"] = r.uniform(-0.5, 0.5, (n, 3)); sph["radius"] = r.uniform(4.0, 6.0, n)
sph["mat_type"] = r.integers(0, 3, n); sph["albedo"] = 0.5; sph["ior"] = 1.5
rt.init(0); rt.set_world(sph)
p = rt.default_params(); p.pos = rt.Vec3(0.1, 0.0, -0.1); p.target = rt.Vec3(3.0, 0.5, 1.0); p.vfov = 100.0
cam = rt.camera(512, 288, p)
rt.render(cam, 512, 288, 2, 6, strict_fp=False, seed=3, chunk_spp=1)
rt.shutdown()
