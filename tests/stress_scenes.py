"""Scenes and views that stress the sphere filter's error bound (the expanded discriminant cancels
worst for large |O| + |C|): large coordinates with huge spheres, a small scene far from the origin of
coordinates, nested shells around the eye, and an eye more than 10^4 units from the spheres.
Shared by the GPU parity tests and the diagnostic-library worker."""
import numpy as np

SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("mat_type", "<i4"),
                         ("albedo", "<f4", 3), ("fuzz", "<f4"), ("ior", "<f4")])

# kind -> (spheres, eye, target, vfov in degrees as the reference's update_camera_view takes it:
# tan_f32 works in turns (util.h:134,142), so the half height of the view is tan(pi^2 * vfov / 180))
VIEWS = {
    "wide":   (300, (120.0, 40.0, 90.0), (0.0, 0.0, 0.0), 100.0),
    "tiny":   (200, (201.5, 50.5, -118.5), (200.0, 50.0, -120.0), 100.0),
    "nested": (40, (6.0, 2.0, 5.0), (0.0, 0.0, 0.0), 100.0),
    "far":    (150, (9000.0, 4000.0, 7000.0), (0.0, 0.0, 0.0), 0.1),
    "inside": (60, (0.3, 0.1, -0.2), (5.0, 0.5, 1.0), 100.0),
    # beyond the range in which the kernel's exact test may use its short form of the IEEE division (a ray's |d|^2 within
    # 2^-40 .. 2^40, origin and spheres within 2^24 of the origin of coordinates): a scene of huge spheres seen from 2e6
    # units away (|d|^2 = 4e12 for every camera ray), and the wide scene with one sphere 4e7 units away; both go through
    # the all-spheres test with the compiler's division
    "remote": (150, (1.6e6, 0.54e6, 1.06e6), (0.0, 0.0, 0.0), 7.0),
    "beyond": (300, (120.0, 40.0, 90.0), (0.0, 0.0, 0.0), 100.0),
}


def random_scene(kind, n=None, seed=5):
    n = VIEWS[kind][0] if n is None else n
    r = np.random.default_rng(seed)
    s = np.zeros(n, dtype=SPHERE_DTYPE)
    if kind == "wide":          # large coordinates, one huge ground, several big spheres
        s["center"] = r.uniform(-50, 50, (n, 3))
        s["radius"] = r.uniform(0.5, 5.0, n)
        s["center"][0] = (0, -5000, 0)
        s["radius"][0] = 4950.0
        s["center"][1] = (30, 10, -400)
        s["radius"][1] = 300.0
    elif kind == "tiny":        # small scene far from the origin of coordinates
        s["center"] = r.uniform(-1, 1, (n, 3)) + np.array([200.0, 50.0, -120.0])
        s["radius"] = r.uniform(0.05, 0.3, n)
    elif kind == "far":         # ordinary scene seen from more than 10^4 units away
        s["center"] = r.uniform(-60, 60, (n, 3))
        s["radius"] = r.uniform(1.0, 6.0, n)
    elif kind == "remote":      # huge spheres seen from 2e6 units away
        s["center"] = r.uniform(-6e5, 6e5, (n, 3))
        s["radius"] = r.uniform(1e4, 6e4, n)
    elif kind == "beyond":      # the wide scene and one sphere 4e7 units below it
        s["center"] = r.uniform(-50, 50, (n, 3))
        s["radius"] = r.uniform(0.5, 5.0, n)
        s["center"][0] = (0, -5000, 0)
        s["radius"][0] = 4950.0
        s["center"][1] = (30, 10, -400)
        s["radius"][1] = 300.0
        s["center"][2] = (0, -4.0e7, 0)
        s["radius"][2] = 1.0e7
    elif kind == "inside":      # the eye sits inside many overlapping spheres of very different sizes
        s["center"] = r.normal(0, 1.5, (n, 3))
        s["radius"] = np.exp(r.uniform(np.log(0.3), np.log(40.0), n))
    else:                        # nested glass shells around the look-at point
        s["center"] = r.normal(0, 0.05, (n, 3))
        s["radius"] = np.linspace(0.2, 3.0, n)
    s["mat_type"] = r.integers(0, 3, n)
    if kind == "nested":
        s["mat_type"] = 2
    s["albedo"] = r.uniform(0.2, 0.9, (n, 3))
    s["fuzz"] = r.uniform(0, 0.5, n)
    s["ior"] = 1.5
    return s


def view_params(mod, kind, defocus_angle=0.0):
    """rt_camera_params of the view (mod: raytracer_amd or oracle.refcpu; same field layout)."""
    _, pos, target, vfov = VIEWS[kind]
    p = mod.default_params()
    p.pos = mod.Vec3(*pos)
    p.target = mod.Vec3(*target)
    p.vfov = vfov
    p.defocus_angle = defocus_angle
    return p
