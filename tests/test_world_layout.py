"""The device order of the spheres (rt_world_layout, pure host code): spheres 32 t .. 32 t + 31 are tile t of the
matrix scan, which multiplies a tile only if its axis-aligned box meets a ray of the wave.  The reference has no such
order (hit() walks the caller's list, Main.c:688-713); what is checked here is that the order is a permutation that
depends on nothing but the scene, that the split in space gives smaller boxes than the plain Z-order curve, and that
the box test as the kernel states it (centre / half-extent slabs in f32, widened by 2^-9 (|O| + max |C|)) never drops
a sphere the exact test of Main.c:625-686 accepts."""
import os

import numpy as np
import pytest

from conftest import ROOT
import raytracer_amd as rt

TILE = 32


def scene(name):
    return np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % name), dtype=rt.SPHERE_DTYPE)


def filtered(sph):
    """Spheres the matrix filter holds (rt_set_world leaves those whose |C| + r exceeds four times the median to a test of their own)."""
    ext = np.linalg.norm(sph["center"].astype(np.float64), axis=1) + np.abs(sph["radius"].astype(np.float64))
    return ext <= 4.0 * np.sort(ext)[len(sph) // 2]


def tile_boxes(sph, order):
    """(lo, hi) per tile over the filtered spheres, float64."""
    keep = filtered(sph)
    boxes = []
    for t in range(0, len(order), TILE):
        idx = [i for i in order[t:t + TILE] if keep[i]]
        if not idx:
            boxes.append(None)
            continue
        c = sph["center"][idx].astype(np.float64)
        r = np.abs(sph["radius"][idx].astype(np.float64))[:, None]
        boxes.append(((c - r).min(axis=0), (c + r).max(axis=0)))
    return boxes


def area(box):
    d = box[1] - box[0]
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]


@pytest.mark.parametrize("name", ["sf", "sg", "s3"])
@pytest.mark.parametrize("split", [True, False])
def test_order_is_a_permutation_and_repeats(name, split):
    sph = scene(name)
    a = rt.world_layout(sph, split)
    b = rt.world_layout(sph.copy(), split)
    assert sorted(a.tolist()) == list(range(len(sph)))
    assert (a == b).all()


def test_small_and_degenerate_scenes():
    assert len(rt.world_layout(np.zeros(0, dtype=rt.SPHERE_DTYPE))) == 0
    sph = scene("sf")[:TILE]
    assert (rt.world_layout(sph, True) == rt.world_layout(sph, False)).all()       # one tile: nothing to split
    # coincident centres, a centre that is not finite, a huge and far sphere: still a permutation
    sph = scene("sf")[:100].copy()
    sph["center"][:50] = (1.0, 2.0, 3.0)
    sph["center"][50] = (np.nan, 0.0, 0.0)
    sph["center"][51] = (np.inf, 0.0, 0.0)
    sph["center"][52] = (0.0, -1e30, 0.0)
    sph["radius"][52] = 1e30
    order = rt.world_layout(sph, True)
    assert sorted(order.tolist()) == list(range(100))


def test_split_tiles_have_smaller_boxes_than_the_curve():
    sph = scene("sf")
    total = {}
    for split in (False, True):
        total[split] = sum(area(b) for b in tile_boxes(sph, rt.world_layout(sph, split)) if b is not None)
    assert total[True] < 0.8 * total[False], total


def test_larger_generated_scene():
    sph = rt.scene_generate(32, 0)                 # 4098 spheres, 129 tiles
    order = rt.world_layout(sph, True)
    assert sorted(order.tolist()) == list(range(len(sph)))
    boxes = [b for b in tile_boxes(sph, order) if b is not None]
    curve = [b for b in tile_boxes(sph, rt.world_layout(sph, False)) if b is not None]
    assert sum(map(area, boxes)) < 0.8 * sum(map(area, curve))


def test_box_test_keeps_every_sphere_the_exact_test_accepts():
    """The slab test in the kernel's arithmetic (f32, reciprocal directions, boxes as centre and half extents rounded
    outwards and widened per ray) against hit_sphere's acceptance (Main.c:625-686) in f64: a sphere a ray hits within
    (tmin, closest) must lie in a tile whose box the ray meets."""
    sph = scene("sf")
    order = rt.world_layout(sph, True)
    keep = filtered(sph)
    boxes = tile_boxes(sph, order)
    f32 = np.float32
    centre = np.zeros((len(boxes), 3), f32)
    half = np.full((len(boxes), 3), -np.inf, f32)
    for t, b in enumerate(boxes):
        if b is None:
            continue
        lo, hi = b[0] - 1e-4, b[1] + 1e-4
        c = (0.5 * (lo + hi)).astype(f32)
        centre[t] = c
        half[t] = np.nextafter(np.maximum(hi - c.astype(np.float64), c.astype(np.float64) - lo).astype(f32), f32(np.inf))
    c_near = f32(np.linalg.norm(sph["center"][keep].astype(np.float64), axis=1).max())
    tile_of = np.empty(len(sph), np.int64)
    tile_of[order] = np.arange(len(sph)) // TILE
    rng = np.random.default_rng(7)
    n_rays = 4000
    # origins on and around the spheres, among them points on sphere surfaces (bounce rays); directions of any length
    origin = np.where(rng.random((n_rays, 1)) < 0.5,
                      rng.uniform((-12, 0, -12), (12, 3, 12), (n_rays, 3)),
                      sph["center"][rng.integers(1, len(sph), n_rays)] + 0.2 * rng.standard_normal((n_rays, 3)))
    direction = rng.standard_normal((n_rays, 3)) * np.exp(rng.uniform(-3, 3, (n_rays, 1)))
    direction[::7, 1] = 0.0                                                        # rays parallel to a slab
    origin, direction = origin.astype(f32), direction.astype(f32)
    C = sph["center"].astype(np.float64)
    R2 = sph["radius"].astype(np.float64) ** 2
    checked = 0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for o, d in zip(origin, direction):
            o64, d64 = o.astype(np.float64), d.astype(np.float64)
            oc = C - o64
            a = d64 @ d64
            h = oc @ d64
            disc = h * h - a * ((oc * oc).sum(axis=1) - R2)
            ok = disc >= 0
            sq = np.sqrt(np.where(ok, disc, 0.0))
            near, far = (h - sq) / a, (h + sq) / a
            t_hit = np.where(near > 0.001, near, far)
            accepted = ok & keep & (t_hit > 0.001)
            if not accepted.any():
                continue
            closest = f32(min(t_hit[accepted].max() * 1.000001, 3.0e38))          # every accepted sphere lies within closest
            inv = f32(1.0) / d
            oi = o * inv
            widen = (np.sqrt((o * o).sum(dtype=f32)) * f32(1.0000005) + c_near) * f32(1.0 / 512.0)
            hh = half + widen
            m = centre * inv - oi
            t_in = np.maximum(np.fmax.reduce(m - hh * np.abs(inv), axis=1), f32(0.0))
            t_out = np.minimum(np.fmin.reduce(m + hh * np.abs(inv), axis=1), closest)
            meets = ~(t_out < t_in)
            assert meets[tile_of[np.nonzero(accepted)[0]]].all(), (o, d)
            checked += int(accepted.sum())
    assert checked > 2000
