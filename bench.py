#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (pixels x spp per second) on the RTIOW final scene,
3840x2160, 1000 spp, depth 50 (the largest configuration the metric is quoted on), on 1..8 MI355X.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  A step = one full frame: every rank path-traces its interleaved row
tiles with the HIP kernel (through the C ABI, device-resident output), the rows are gathered
to rank 0 over RCCL and scattered into the full RGBA8 framebuffer on GPU 0.  The image is
fixed as N grows (strong scaling).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3      # MI355X vector FP32: 256 CU x 256 flop/clk x 2.4 GHz
FLOP_PER_TEST = 18            # hit_sphere's miss path (Main.c:628-654) with |d|^2 hoisted per ray


def cpu_baseline(width, spp, depth, threads):
    """The reference's own code (oracle/_ref, built from /root/reference in the build container)
    timed on this host; falls back to the CPU port when that binary did not travel."""
    from oracle import refcpu as rc
    height = rc.height_for(width)
    samples = width * height * spp
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_sf")
    sample = "final scene %dx%d, %d spp (of 1000), depth %d, %d threads over 64x64 tiles" % (
        width, height, spp, depth, threads)
    if os.path.exists(exe):
        with tempfile.NamedTemporaryFile(suffix=".rgba") as f:
            t0 = time.perf_counter()
            subprocess.check_call([exe, "mc", str(width), str(spp), str(depth), f.name, str(threads)],
                                  stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
        kind = "reference"
    else:
        sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_sf.bin"), dtype=rc.SPHERE_DTYPE)
        t0 = time.perf_counter()
        rc.render(rc.camera(width, height), sph, width, height, spp, depth, rng=rc.RNG_LCG_TILES,
                  threads=threads)
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": sample, "seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sf", choices=["sf", "s3"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--chunk", type=int, default=-1, help="samples per work unit (-1: the library chooses)")
    ap.add_argument("--tile-rows", type=int, default=2, help="rows per dealt tile: small tiles keep the deal fine-grained (2160 rows over 8 ranks)")
    ap.add_argument("--strict", type=int, default=0, help="1 = bit-reproducible build (no FMA)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-rebalance", action="store_true", help="keep the equal row-tile deal (N > 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the gather through host memory (rehearsal only)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--cpu-width", type=int, default=3840)
    ap.add_argument("--cpu-spp", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    if args.shared_gpu:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --shared-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where small collectives live

    W = args.width
    H = rt.height_for(W)
    sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % args.scene), dtype=rt.SPHERE_DTYPE)
    rt.init(local)
    rt.set_world(sph)
    cam = rt.camera(W, H)

    from raytracer_amd.multigpu import RowGather, weighted_row_maps

    class Part:                                                    # the current row-tile deal
        pass
    part = Part()

    def deal(weights):
        part.weights = [float(w) for w in weights]
        part.rows = weighted_row_maps(H, args.tile_rows, part.weights)
        # compact owned rows -> full image on rank 0
        part.rg = RowGather(part.rows, W, 4, torch.uint8, dev, rank, world, via_host=(args.backend == "gloo"))

    deal([1.0] * world)                                            # equal weights = tile t -> rank t % world
    part.rg.connect()                                              # peer connections exist before anything is timed

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        st = rt.render_device(cam, part.rg.mine.data_ptr(), W, H, args.spp, args.depth,
                              strict_fp=bool(args.strict), seed=0, chunk_spp=args.chunk, rows=part.rows[rank])
        part.rg.gather()                                              # RCCL over xGMI when world > 1
        return st

    for i in range(args.warmup):
        st = step()
        if world > 1 and not args.no_rebalance and i < max(1, args.warmup - 1):
            # MI355X devices hold different clocks on this compute-bound kernel (10 % apart is
            # common): deal the row tiles in proportion to the speed each rank just showed, so
            # that all ranks finish together (again after every warm-up step but the last, which
            # runs the final deal untimed).  The image does not depend on the deal.
            speed = torch.zeros(world, dtype=torch.float64, device=coll_dev)
            speed[rank] = len(part.rows[rank]) / max(st.kernel_ms, 1e-6)
            dist.all_reduce(speed, op=dist.ReduceOp.SUM)
            sp = speed.cpu().numpy()
            med = float(np.median(sp))
            deal(np.clip(sp, 0.75 * med, 1.33 * med).tolist())        # one odd timing must not starve a rank
    sync()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    sync()
    dt = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        img = part.rg.full.cpu().numpy()
        assert (img[..., 3] == 255).all(), "framebuffer not fully written"
        samples = W * H * args.spp
        kernel_ms = float(np.mean([s.kernel_ms for s in stats]))                 # HIP events on the library's stream
        tests = float(np.mean([s.sphere_tests for s in stats]))
        rays = float(np.mean([s.rays for s in stats]))
        own_samples = float(np.mean([s.samples for s in stats]))
        achieved = FLOP_PER_TEST * tests / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "Msamples/s (px*spp/s) RTIOW final scene" if args.scene == "sf" else "Msamples/s (px*spp/s) 3-sphere scene",
            "value": round(samples * args.steps / dt / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (the reference's generated %d-sphere scene, committed fixture)" % len(sph),
            "config": {"workload": "RTIOW %s scene %dx%d, %d spp, depth %d" % (
                           "final" if args.scene == "sf" else "3-sphere", W, H, args.spp, args.depth),
                       "spheres": int(len(sph)), "chunk_spp": int(stats[-1].chunk_spp), "tile_rows": args.tile_rows,
                       "kernel": "strict" if args.strict else "fast",
                       "parallelism": "row-tiles x%d, %s gather to rank 0%s" % (
                           world, "RCCL" if args.backend == "nccl" else "gloo (host-staged)",
                           ", all ranks on ONE gpu (rehearsal)" if args.shared_gpu else ""),
                       "rows_per_rank": [int(len(r)) for r in part.rows]},
            "roofline": {"bound": "valu_fp32", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                         # HBM bytes per launch of trace_kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
                         # passes on this command (the counter files are kept under profiles/;
                         # FETCH_SIZE doubled, the gfx950 correction): 2.27 GB written + 0.01-0.2 GB read.
                         # Only known for the default workload on one GPU, null otherwise
                         "traffic": (2.3e9 if (world == 1 and W == 3840 and args.spp == 1000
                                                and int(stats[-1].chunk_spp) == 63 and not args.strict) else None),
                         "kernel": "trace_kernel", "kernel_ms": round(kernel_ms, 3),
                         "flop_per_sphere_test": FLOP_PER_TEST,
                         "sphere_tests_per_launch": tests, "rays_per_sample": round(rays / own_samples, 4)},
            "kernel_only_msamples_per_s": round(own_samples / kernel_ms / 1e3, 3),
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_width, args.cpu_spp, args.depth, args.cpu_threads)
        print(json.dumps(out), flush=True)

    rt.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
