"""Bit-exact comparison of the exact and strict builds with the CPU checker on many random scenes (the scenes of
tests/test_gpu_parity.py::_tie_scene and generated scenes, sizes on both sides of every switch of the scan):
    python tools/random_scene_soak.py [scenes]          (checker = oracle/refcpu, test infrastructure)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt
from oracle import refcpu as rc
from test_gpu_parity import _tie_scene, CHECKED

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2024)
rt.init(0)
W, H = 128, 72
bad = 0; rays = 0
t0 = time.time()
for k in range(count):
    kind = k % 3
    if kind == 2:
        sph = rt.scene_generate(int(rng.integers(3, 40)), int(rng.integers(0, 1 << 30)))
        p = rt.default_params()
    else:
        n = int(rng.choice([33, 64, 65, 200, 488, 1000, 1024, 1025, 1500, 2048, 2049, 3000, 5000, 8200, 9000]))
        sph = _tie_scene(n, 1000 + k)
        if kind == 1:                                  # spread the scene out: boxes that do cull
            sph["center"] *= np.float32(rng.uniform(2.0, 12.0))
        p = rt.default_params()
        p.pos = rt.Vec3(*(float(v) for v in rng.uniform(-12, 12, 3)))
        p.target = rt.Vec3(0.0, 0.0, 0.0)
        p.vfov = float(rng.uniform(20.0, 90.0))
    rt.set_world(sph)
    cam = rt.camera(W, H, p)
    ccam = rc.Camera.from_buffer_copy(bytes(cam))
    for build, mc_rng in CHECKED:
        r = rt.render(cam, W, H, 3, 12, build=build, seed=k, chunk_spp=2, want_hit=True, want_accum=True)
        c = rc.render(ccam, sph, W, H, 3, 12, rng=mc_rng, seed=k, chunk_spp=2, want_hit=True, want_accum=True)
        ok = (np.array_equal(r["hit"], c["hit"]) and np.array_equal(r["accum"].view(np.uint32), c["accum"].view(np.uint32))
              and np.array_equal(r["rgba"], c["rgba"]) and int(r["stats"].rays) == int(c["stats"].rays))
        rays += int(r["stats"].rays)
        if not ok:
            bad += 1
            print("MISMATCH scene %d (%d spheres, kind %d) build %s" % (k, len(sph), kind, build), flush=True)
    if k % 10 == 9:
        print("%d scenes, %d rays, %d mismatches, %.0f s" % (k + 1, rays, bad, time.time() - t0), flush=True)
rt.shutdown()
print("random_scene_soak: %d scenes x 2 builds, %d rays, %d mismatches" % (count, rays, bad))
sys.exit(1 if bad else 0)
