#!/usr/bin/env python3
"""Generates tests/golden/ from the COMPILED REFERENCE (oracle/_ref, built by build_ref.sh).

Runs only where /root/reference exists (the build container).  The fixtures are data:
sphere lists, camera vectors and rendered images / hit-index maps produced by the
reference's own code (init_scene, init_camera, render_tile, get_ray, ray_color, hit_sphere).

  python3 oracle/make_golden.py            # everything except the long renders
  python3 oracle/make_golden.py --long     # also the high-spp Monte-Carlo goldens (~25 min)
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
GOLD = os.path.join(HERE, "..", "tests", "golden")


SG_CAM_POS = (9.0, 7.0, 3.0)     # eye point of the glass-heavy progressive configuration (looks at the origin)


def run(scene, cmd, width, spp, depth, threads=8, cam_pos=None):
    exe = os.path.join(REF, "ref_harness_" + scene)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        out = f.name
    env = dict(os.environ)
    if cam_pos is not None:
        env["REF_CAM_POS"] = "%r,%r,%r" % tuple(cam_pos)
    try:
        subprocess.check_call([exe, cmd, str(width), str(spp), str(depth), out, str(threads)],
                              stderr=subprocess.DEVNULL, env=env)
        return np.fromfile(out, dtype=np.uint8)
    finally:
        os.unlink(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--cameras-only", action="store_true", help="rewrite cameras.npz (and the scene files) only")
    ap.add_argument("--sf-2048", default=None, help="use an already rendered sf 400x225 2048-spp RGBA file")
    args = ap.parse_args()
    if not os.path.exists(os.path.join(REF, "ref_harness_s3")):
        subprocess.check_call([os.path.join(HERE, "build_ref.sh")])
    os.makedirs(GOLD, exist_ok=True)

    # scenes: raw sphere_t records (Main.c:74-79), as init_scene produced them under gcc
    for s in ("s3", "sf"):
        run(s, "scene", 400, 1, 1).tofile(os.path.join(GOLD, "scene_%s.bin" % s))

    # glass-heavy scene "SG" (the 1 spp/frame progressive configuration): the reference holds no definition of it (only a
    # screenshot), so it is DEFINED here: the final scene with every r = 0.2 sphere turned into
    # glass (Dielectric, ior 1.5); viewed from pos (9, 7, 3) looking at the origin.
    sf = np.fromfile(os.path.join(GOLD, "scene_sf.bin"), dtype=np.uint8).reshape(-1, 40).copy()
    rec = sf.view("<f4").reshape(-1, 10)
    small = rec[:, 3] == np.float32(0.2)
    sf.view("<i4").reshape(-1, 10)[small, 4] = 2
    rec[small, 9] = 1.5
    sf.tofile(os.path.join(GOLD, "scene_sg.bin"))

    # cameras at the benchmark widths (init_camera + update_camera_view, Main.c:839-928)
    cams = {}
    for w in (100, 400, 1200, 1920, 3840):
        raw = run("s3", "camera", w, 1, 1)
        cams["cam_%d" % w] = raw[:76].view("<f4").copy()
        cams["wh_%d" % w] = raw[76:84].view("<u4").copy()
    # the glass-heavy configuration's camera: the reference's update_camera_view after moving the eye
    cams["sg_pos"] = np.asarray(SG_CAM_POS, dtype="<f4")
    for w in (100, 1920):
        raw = run("s3", "camera", w, 1, 1, cam_pos=SG_CAM_POS)
        cams["cam_sg_%d" % w] = raw[:76].view("<f4").copy()
    np.savez_compressed(os.path.join(GOLD, "cameras.npz"), **cams)
    if args.cameras_only:
        return 0

    # deterministic constant-u renders + primary hit indices
    for s in ("s3", "sf"):
        for w in (100, 400):
            wh = cams["wh_%d" % w]
            W, H = int(wh[0]), int(wh[1])
            raw = run(s, "d1", w, 1, 50)
            col = raw[:W * H * 12].view("<f4").reshape(H, W, 3)
            px = raw[W * H * 12:W * H * 16].reshape(H, W, 4)
            hit = raw[W * H * 16:].view("<i4").reshape(H, W)
            d0 = run(s, "d0", w, 1, 50).view("<i4").reshape(H, W)
            assert np.array_equal(d0, hit)
            np.savez_compressed(os.path.join(GOLD, "d1_%s_%d.npz" % (s, w)), accum=col, rgba=px, hit=hit)

    # Monte-Carlo renders with the reference's LCG, per-64x64-tile seeds (render_tile)
    mc = [("s3", 400, 100, 50), ("s3", 100, 3, 5), ("sf", 400, 4, 50), ("sf", 100, 3, 5),
          ("s3", 400, 1, 2)]
    for s, w, spp, depth in mc:
        wh = cams["wh_%d" % w]
        W, H = int(wh[0]), int(wh[1])
        px = run(s, "mc", w, spp, depth).reshape(H, W, 4)
        np.savez_compressed(os.path.join(GOLD, "mc_lcg_%s_%d_%d_d%d.npz" % (s, w, spp, depth)), rgba=px)
    px = run("s3", "mcser", 400, 4, 50).reshape(225, 400, 4)
    np.savez_compressed(os.path.join(GOLD, "mc_lcgser_s3_400_4_d50.npz"), rgba=px)

    # converged references for the PSNR gate
    if args.long:
        px = run("s3", "mc", 400, 4096, 50).reshape(225, 400, 4)
        np.savez_compressed(os.path.join(GOLD, "mc_lcg_s3_400_4096_d50.npz"), rgba=px)
    if args.sf_2048:
        px = np.fromfile(args.sf_2048, dtype=np.uint8).reshape(225, 400, 4)
        np.savez_compressed(os.path.join(GOLD, "mc_lcg_sf_400_2048_d50.npz"), rgba=px)
    elif args.long:
        px = run("sf", "mc", 400, 2048, 50).reshape(225, 400, 4)
        np.savez_compressed(os.path.join(GOLD, "mc_lcg_sf_400_2048_d50.npz"), rgba=px)
    print("golden fixtures written to", os.path.normpath(GOLD))


if __name__ == "__main__":
    sys.exit(main())
