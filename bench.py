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

The timed build is the exact build (the library's default): reference-order IEEE arithmetic, bit-identical
to the CPU checker, deterministic mode bit-identical to the reference.  At N = 1 the same run also reports,
after the timed loop: the other two builds on the same workload (strict = the reference's rejection samplers,
fast = FMA and hardware approximations), the rate with the read-back of every frame into page-locked host
memory included (rt_render instead of rt_render_device), and the CPU baseline (the reference's own code on
the host).
"""
import argparse
import csv
import glob
import hashlib
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
FLOP_PER_TEST = 18            # ALGORITHMIC: hit_sphere's miss path (Main.c:628-654) with |d|^2 hoisted per ray
PEAK_F16_MFMA_TFLOPS = 2500.0 # MI355X dense f16/bf16 matrix peak
FLOP_PER_MFMA = 32 * 32 * 16 * 2     # one v_mfma_f32_32x32x16_f16
MFMA_PER_TILE = 4             # 32 spheres x 64 rays x 32 K entries: two ray halves x two K halves

PARITY_BAR = {
    "exact": "bit-exact: hit indices, f32 accumulators, RGBA8 bytes and ray counts equal the CPU restatement of the "
             "reference run with the same counter generator and the same loop-free samplers; deterministic mode (1 spp, "
             "centre sample) equals the compiled reference bit for bit at depth 50; PSNR >= 40 dB against the reference's "
             "converged render",
    "strict": "bit-exact: as exact, with the reference's rejection samplers and its order of random draws",
    "fast": "deterministic output; PSNR >= 40 dB against the reference's converged render; every ray's nearest hit equals "
            "the exact test over all spheres (diagnostic build); not reproducible on a CPU",
}


def kernel_sources_id():
    """Identifies the kernel sources a counter pass was taken on: counters of another kernel are not reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def host_cores():
    """(cores the host reports, cores this job may use).  The reference sizes its thread pool from
    get_core_count() = sysconf(_SC_NPROCESSORS_ONLN) (src/util.c:512-520); a job confined to a share of
    a large host (affinity mask, or the share the launcher states in OMP_NUM_THREADS) must not start
    two threads per core of the whole machine."""
    reported = os.cpu_count() or 1
    usable = reported
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
        if share > 0:
            usable = min(usable, share)
    except ValueError:
        pass
    return reported, max(1, usable)


def cpu_baseline(width, spp, depth, threads=0):
    """The reference's own code (oracle/_ref, built from /root/reference in the build container)
    timed on this host in the reference's threading shape: 64x64 tiles, one OS thread per tile,
    2 x cores thread slots, slot i % T joined before reuse (Main.c:1002-1054).  Falls back to the CPU
    port (a tile pool of the same size) when that binary did not travel."""
    from oracle import refcpu as rc
    reported, usable = host_cores()
    if threads <= 0:
        threads = 2 * usable                                          # Main.c:1002
    height = rc.height_for(width)
    samples = width * height * spp
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_sf")
    sample = ("final scene %dx%d, %d spp (of 1000), depth %d; %d thread slots = 2 x the %d cores this job may use "
              "(the host reports %d), one thread per 64x64 tile" % (width, height, spp, depth, threads, usable, reported))
    if os.path.exists(exe):
        with tempfile.NamedTemporaryFile(suffix=".rgba") as f:
            t0 = time.perf_counter()
            subprocess.check_call([exe, "mcref", str(width), str(spp), str(depth), f.name, str(threads)],
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
    return {"value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": usable, "kind": kind,
            "host_cores": reported, "threads": threads,
            "sample": sample, "seconds": round(dt, 2)}


def measured_traffic(kernel_substr, workload_key):
    """HBM bytes per launch of the path-tracing kernel from the newest committed rocprofv3 --pmc
    passes (profiles/<round>/pmc_fetch.csv + pmc_write.csv, each its own run of this command):
    Counter_Value is in KiB; FETCH_SIZE is doubled (on gfx950 it tallies 128-byte requests as 64).
    None if no pass for this workload AND these kernel sources is on file (profiles/<round>/pmc_workload.txt
    names the workload, pmc_sources.txt the sources the passes ran on): counters are never carried over to
    another kernel."""
    best = None
    sources = kernel_sources_id()
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "*"))):
        fw, ff, fk, fs = (os.path.join(d, n) for n in ("pmc_write.csv", "pmc_fetch.csv", "pmc_workload.txt", "pmc_sources.txt"))
        if not (os.path.exists(fw) and os.path.exists(ff) and os.path.exists(fk) and os.path.exists(fs)):
            continue
        if open(fk).read().strip() != workload_key or open(fs).read().strip() != sources:
            continue
        tot = {}
        for path, name, scale in ((fw, "WRITE_SIZE", 1024.0), (ff, "FETCH_SIZE", 2048.0)):
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == name]
            if not vals:
                tot = None
                break
            tot[name] = scale * sum(vals) / len(vals)
        if tot:
            best = {"bytes": tot["WRITE_SIZE"] + tot["FETCH_SIZE"], "write": tot["WRITE_SIZE"], "fetch": tot["FETCH_SIZE"],
                    "source": os.path.relpath(d, ROOT), "busy": measured_busy(d, kernel_substr)}
    return best


def measured_busy(d, kernel_substr):
    """Occupation of the two pipes the kernel issues to, from the SQ counter passes of the same profile directory:
    vector ALU = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) and matrix cores = SQ_VALU_MFMA_BUSY_CYCLES, both over
    GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs; lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU).  None if absent."""
    val = {}
    for name in ("pmc_a.csv", "pmc_b.csv"):
        path = os.path.join(d, name)
        if not os.path.exists(path):
            return None
        for r in csv.DictReader(open(path)):
            if kernel_substr in r["Kernel_Name"]:
                val[r["Counter_Name"]] = val.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    try:
        simd_cycles = val["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        return {"valu_issue": round(4.0 * val["SQ_ACTIVE_INST_VALU"] / simd_cycles, 3),
                "mfma": round(val["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 3),
                "lanes": round(val["SQ_THREAD_CYCLES_VALU"] / (64.0 * val["SQ_ACTIVE_INST_VALU"]), 3),
                "mfma_flop_per_launch": 512.0 * val["SQ_INSTS_VALU_MFMA_MOPS_F16"],
                "vector_instructions_per_launch": val["SQ_INSTS_VALU"] - val.get("SQ_INSTS_MFMA", 0.0)}
    except (KeyError, ZeroDivisionError):
        return None


class MockRt:
    """--mock: stands in for the library so that the N > 1 plumbing and the JSON line can be checked
    on a machine without a GPU (tests only; the line says "mock": true and measures nothing)."""
    SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("mat_type", "<i4"),
                             ("albedo", "<f4", 3), ("fuzz", "<f4"), ("ior", "<f4")])

    class _St:
        pass

    def __init__(self, torch):
        self.torch = torch
        self.buffers = {}

    def init(self, dev): pass
    def set_world(self, sph): self.n = len(sph)
    def shutdown(self): pass
    def height_for(self, w): return int(np.float32(w) / np.float32(16.0 / 9.0))
    def camera(self, w, h): return None
    def chunk_for(self, w, h, spp, chunk=-1, progressive=False): return 8 if chunk < 0 else (chunk or spp)

    def render_device(self, cam, ptr, W, H, spp, depth, rows=None, chunk_spp=-1, **kw):
        self.buffers[ptr][:len(rows)] = 255
        st = self._St()
        st.samples = len(rows) * W * spp
        st.rays = 3 * st.samples
        st.sphere_tests = st.rays * self.n
        st.kernel_ms = 1.0 + 0.001 * len(rows)
        st.chunk_spp = self.chunk_for(W, H, spp, chunk_spp)
        st.wave_scans = st.rays // 64
        st.scan_tiles = (self.n + 31) // 32
        st.tile_products = st.wave_scans * st.scan_tiles // 2           # as if the cull skipped every other tile
        return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="sf", choices=["sf", "s3"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--chunk", type=int, default=-1, help="samples per work unit (-1: the library chooses, from the full image size and spp)")
    ap.add_argument("--tile-rows", type=int, default=2, help="rows per dealt tile: small tiles keep the deal fine-grained (2160 rows over 8 ranks)")
    ap.add_argument("--build", default="exact", choices=["exact", "strict", "fast"], help="the build the main loop times")
    ap.add_argument("--other-steps", type=int, default=2, help="N = 1: steps of each of the other two builds after the timed loop (0 = skip)")
    ap.add_argument("--readback-steps", type=int, default=2, help="N = 1: steps of rt_render into page-locked host memory (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: only the timed loop (no other builds / read-back / CPU legs)")
    ap.add_argument("--no-rebalance", action="store_true", help="keep the equal row-tile deal (N > 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the gather through host memory (rehearsal only)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: run the process-group calls of the N > 1 path (init, all_reduce, gather, barrier) with one rank, "
                         "so that a one-GPU box executes the RCCL branch")
    ap.add_argument("--mock", action="store_true", help="no GPU, no kernel: checks the N > 1 plumbing and the JSON shape (needs --backend gloo)")
    ap.add_argument("--n1-value", type=float, default=0.0, help="the N = 1 value of the same workload: adds efficiency_vs_n1")
    ap.add_argument("--cpu-width", type=int, default=3840)
    ap.add_argument("--cpu-spp", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=0, help="thread slots of the CPU leg (0 = 2 x usable cores, the reference's rule)")
    args = ap.parse_args()
    if args.no_extras:
        args.other_steps = 0
        args.readback_steps = 0
        args.no_cpu = True

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)"
                         % (args.gpus, world))
    if args.mock:
        if args.backend != "gloo":
            raise SystemExit("bench.py: --mock needs --backend gloo")
        rt = MockRt(torch)
        dev = torch.device("cpu")
    else:
        import raytracer_amd as rt
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
        if args.shared_gpu:
            if args.backend != "gloo":
                raise SystemExit("bench.py: --shared-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where small collectives live
    on_gpu = dev.type == "cuda"

    W = args.width
    H = rt.height_for(W)
    sph = np.fromfile(os.path.join(ROOT, "tests", "golden", "scene_%s.bin" % args.scene), dtype=rt.SPHERE_DTYPE)
    rt.init(local)
    rt.set_world(sph)
    cam = rt.camera(W, H)
    # one unit size for all ranks: resolved from the full image, never from a rank's share of it
    chunk = rt.chunk_for(W, H, args.spp, args.chunk)

    from raytracer_amd.multigpu import RowGather, weighted_row_maps

    class Part:                                                    # the current row-tile deal
        pass
    part = Part()

    def deal(weights):
        part.weights = [float(w) for w in weights]
        part.rows = weighted_row_maps(H, args.tile_rows, part.weights)
        # compact owned rows -> full image on rank 0
        part.rg = RowGather(part.rows, W, 4, torch.uint8, dev, rank, world, via_host=(args.backend == "gloo"),
                            always_collective=dist_on)
        if args.mock:
            for b in part.rg._send:
                rt.buffers[b.data_ptr()] = b

    deal([1.0] * world)                                            # equal weights = tile t -> rank t % world
    rows_initial = [int(len(r)) for r in part.rows]
    part.rg.connect()                                              # peer connections exist before anything is timed

    def sync():
        if dist_on:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    gather_ms = []

    def step(build):
        st = rt.render_device(cam, part.rg.mine.data_ptr(), W, H, args.spp, args.depth,
                              build=build, seed=0, chunk_spp=chunk, rows=part.rows[rank])
        t0 = time.perf_counter()
        part.rg.gather()                                              # RCCL over xGMI when world > 1
        if on_gpu:
            torch.cuda.synchronize()
        gather_ms.append((time.perf_counter() - t0) * 1e3)
        return st

    for i in range(args.warmup):
        st = step(args.build)
        if dist_on and not args.no_rebalance and i < max(1, args.warmup - 1):
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
    gather_ms.clear()
    sync()
    t0 = time.perf_counter()
    stats = [step(args.build) for _ in range(args.steps)]
    sync()
    dt = time.perf_counter() - t0

    kernel_ms = float(np.mean([s.kernel_ms for s in stats]))                 # HIP events on the library's stream
    per_rank_kernel_ms = [kernel_ms]
    per_rank_gather_ms = [float(np.mean(gather_ms))]
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        k = torch.zeros(2 * world, dtype=torch.float64, device=coll_dev)
        k[rank] = kernel_ms
        k[world + rank] = float(np.mean(gather_ms))
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        k = k.cpu().numpy()
        per_rank_kernel_ms = [round(float(v), 3) for v in k[:world]]
        per_rank_gather_ms = [round(float(v), 3) for v in k[world:]]
        # whole-frame statistics: sums over ranks
        tot = torch.tensor([float(np.mean([s.sphere_tests for s in stats])), float(np.mean([s.rays for s in stats])),
                            float(np.mean([s.samples for s in stats]))], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tests_all, rays_all, samples_all = [float(v) for v in tot.cpu().numpy()]
        # Does the assembled frame hold every rank's rows where they belong?  Rank 0 renders rows that OTHER ranks
        # own itself (the image does not depend on who renders a row) and compares them with what the gather
        # delivered, byte for byte; not timed.
        gather_verified = None
        if rank == 0 and not args.mock and world > 1:
            foreign = np.concatenate([np.asarray(part.rows[r]) for r in range(1, world)])
            probe = np.unique(foreign[np.linspace(0, len(foreign) - 1, 16).astype(np.int64)]).astype(np.uint32)
            mine = torch.zeros((len(probe), W, 4), dtype=torch.uint8, device=dev)
            rt.render_device(cam, mine.data_ptr(), W, H, args.spp, args.depth, build=args.build, seed=0,
                             chunk_spp=chunk, rows=probe)
            torch.cuda.synchronize()
            got = part.rg.full[torch.as_tensor(probe.astype(np.int64), device=dev)]
            gather_verified = bool(torch.equal(mine, got))
            assert gather_verified, "gathered rows differ from rank 0's own render of the same rows"
        dist.barrier()
    else:
        tests_all = float(np.mean([s.sphere_tests for s in stats]))
        rays_all = float(np.mean([s.rays for s in stats]))
        samples_all = float(np.mean([s.samples for s in stats]))

    def rates_of(sts):
        """Kernel-level rates of a list of launches: the executed matrix work against the dense f16 peak (the
        unit the ray-sphere tests run on), and the reference formulation's flop count (18 per test) against the
        vector FP32 peak (a speed-up measure: it exceeds 1, it is not a roofline)."""
        ms = float(np.mean([s.kernel_ms for s in sts]))
        tests = float(np.mean([s.sphere_tests for s in sts]))
        mflop = float(np.mean([float(s.tile_products) * MFMA_PER_TILE * FLOP_PER_MFMA for s in sts]))
        return {"kernel_ms": ms, "tests": tests, "matrix_flop": mflop,
                "matrix_tflops": mflop / (ms * 1e-3) / 1e12,
                "algorithmic_tflops": FLOP_PER_TEST * tests / (ms * 1e-3) / 1e12}

    if rank == 0:
        img = part.rg.full.cpu().numpy()
        assert (img[..., 3] == 255).all(), "framebuffer not fully written"
        samples = W * H * args.spp
        kname = args.build
        rates = rates_of(stats)                                                   # this rank's launches
        workload = "RTIOW %s scene %dx%d, %d spp, depth %d" % (
            "final" if args.scene == "sf" else "3-sphere", W, H, args.spp, args.depth)
        traffic = None
        if world == 1 and not args.mock:
            traffic = measured_traffic("%s_build::trace_kernel" % kname, "%s, %s build, chunk %d" % (workload, kname, chunk))
        busy = (traffic or {}).get("busy") or {}
        on_matrix = rates["matrix_flop"] > 0
        if on_matrix:
            # The ray-sphere tests run on the matrix cores (one 32-term f16 inner product per test, f32 accumulation;
            # the vector ALU compares and runs the exact f32 test on what passes), so that is the roofline: executed
            # matrix flop (4 instructions of 32768 flop per 32-sphere tile a wave scan multiplies, padding rows included,
            # counted by the kernel; SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 of the committed counter pass agrees) per
            # second of kernel time against the dense f16 peak.  A scan skips the tiles whose bounding box none of its
            # rays meets, so fewer flop are executed for the same result: the fraction measures the pipe's use, the
            # metric (samples per second) the result.
            roof = {"bound": "mfma", "resource": "f16 matrix cores (v_mfma_f32_32x32x16_f16)",
                    "achieved": round(rates["matrix_tflops"], 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(rates["matrix_tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                    "numerator": "executed matrix flop (tiles multiplied x 4 MFMA x 32768; a wave scan multiplies only the tiles "
                                 "whose bounding box one of its rays meets) / kernel time",
                    "tiles_multiplied_per_wave_scan": round(float(np.mean([float(s.tile_products) / max(float(s.wave_scans), 1.0) for s in stats])), 3),
                    "tiles_per_scene": int(stats[-1].scan_tiles),
                    "matrix_flop_per_launch": rates["matrix_flop"],
                    "matrix_flop_per_launch_from_counters": busy.get("mfma_flop_per_launch"),
                    # what the kernel waits for since it multiplies a third of the tiles: the vector ALU's issue port
                    "binding_resource": "vector ALU issue port", "binding_resource_busy": busy.get("valu_issue")}
        else:
            roof = {"bound": "valu_fp32", "resource": "vector ALU (packed-FP32 scan: fewer than 32 spheres)",
                    "achieved": round(rates["algorithmic_tflops"], 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(rates["algorithmic_tflops"] / PEAK_FP32_TFLOPS, 4),
                    "numerator": "algorithmic flop (18 per ray-sphere test) / kernel time"}
        algorithmic_bytes = W * H * 16                      # one f32 colour sum per pixel is all the path has to write
        roof.update({
            "traffic": round(traffic["bytes"]) if traffic else None,
            "traffic_ratio": round(traffic["bytes"] / algorithmic_bytes, 2) if traffic else None,
            "traffic_note": "HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB units) over width x height x 16 B",
            "traffic_source": traffic["source"] if traffic else None,
            "kernel": "%s_build::trace_kernel" % kname, "kernel_ms": round(rates["kernel_ms"], 3),
            "sphere_tests_per_launch": rates["tests"], "wave_scans_per_launch": float(np.mean([s.wave_scans for s in stats])),
            "rays_per_sample": round(rays_all / samples_all, 4)})
        scans = float(np.mean([s.wave_scans for s in stats]))
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
            "config": {"workload": workload,
                       "spheres": int(len(sph)), "chunk_spp": int(stats[-1].chunk_spp), "tile_rows": args.tile_rows,
                       "kernel": kname, "parity_bar": PARITY_BAR[kname],
                       "parallelism": "row-tiles x%d, %s gather to rank 0%s" % (
                           world, "RCCL" if args.backend == "nccl" else "gloo (host-staged)",
                           ", all ranks on ONE gpu (rehearsal)" if args.shared_gpu else ""),
                       "rows_per_rank": [int(len(r)) for r in part.rows]},
            "roofline": roof,
            # What binds the kernel, from the committed counter passes of THESE kernel sources on THIS workload
            # (null otherwise): the share of the time the vector-ALU issue port (the binding resource) and the
            # matrix pipe were occupied, the share of the 64 lanes at work per vector instruction, and the
            # vector instructions a wave spends per scan of the sphere list.
            "valu_issue_busy": busy.get("valu_issue"),
            "mfma_busy": busy.get("mfma"),
            "lane_occupation": busy.get("lanes"),
            "vector_instr_per_wave_scan": round(busy["vector_instructions_per_launch"] / scans, 1) if busy and scans > 0 else None,
            "counters_source": traffic["source"] if traffic else None,
            # the reference's formulation (18 flop per ray-sphere test on the vector ALU) would need this multiple of
            # the 157.3 TFLOP/s vector peak to keep up: a speed-up over that formulation, not a roofline fraction
            "algorithmic_vs_vector_peak": round(rates["algorithmic_tflops"] / PEAK_FP32_TFLOPS, 4),
            "kernel_only_msamples_per_s": round(float(np.mean([s.samples for s in stats])) / rates["kernel_ms"] / 1e3, 3),
            "gather_ms": round(per_rank_gather_ms[0], 3),
        }
        if args.mock:
            out["mock"] = True
        if dist_on:
            out["backend"] = dist.get_backend()
            out["dist_world"] = dist.get_world_size()
            out["kernel_ms_per_rank"] = per_rank_kernel_ms
            out["gather_ms_per_rank"] = per_rank_gather_ms
            out["rows_per_rank_initial"] = rows_initial
            out["rows_per_rank"] = [int(len(r)) for r in part.rows]
            out["sphere_tests_all_ranks"] = tests_all
            out["gather_verified"] = gather_verified
            if args.n1_value > 0:
                out["efficiency_vs_n1"] = round(out["value"] / (world * args.n1_value), 4)
        if world == 1 and not args.mock:
            if args.other_steps > 0:
                out["builds"] = {}
                for other in ("strict", "fast", "exact"):
                    if other == kname:
                        continue
                    step(other)                                                  # warm-up of that build
                    sync()
                    t1 = time.perf_counter()
                    ost = [step(other) for _ in range(args.other_steps)]
                    sync()
                    odt = time.perf_counter() - t1
                    r = rates_of(ost)
                    out["builds"][other] = {"value": round(samples * args.other_steps / odt / 1e6, 3), "unit": "Msamples/s",
                                            "steps": args.other_steps, "kernel_ms": round(r["kernel_ms"], 3),
                                            "roofline_frac": round(r["matrix_tflops"] / PEAK_F16_MFMA_TFLOPS, 4),
                                            "algorithmic_vs_vector_peak": round(r["algorithmic_tflops"] / PEAK_FP32_TFLOPS, 4),
                                            "parity_bar": PARITY_BAR[other]}
                step(kname)                                                      # the frame compared below is the timed build's
                img = part.rg.full.cpu().numpy()
            if args.readback_steps > 0:
                # the metric as a C host sees it: rt_render writes the frame into the caller's
                # (page-locked) host framebuffer; kernel + resolve + one DMA read-back per frame
                buf = rt.host_framebuffer(H, W)
                try:
                    rt.render(cam, W, H, args.spp, args.depth, build=kname, seed=0, chunk_spp=chunk, out=buf)
                    t1 = time.perf_counter()
                    for _ in range(args.readback_steps):
                        rst = rt.render(cam, W, H, args.spp, args.depth, build=kname, seed=0,
                                        chunk_spp=chunk, out=buf)["stats"]
                    rdt = time.perf_counter() - t1
                    assert np.array_equal(buf, img), "host-pointer entry and device-pointer entry disagree"
                    out["value_with_readback"] = round(samples * args.readback_steps / rdt / 1e6, 3)
                    out["readback"] = {"steps": args.readback_steps, "ms_per_step": round(rdt / args.readback_steps * 1e3, 3),
                                       "kernel_ms": round(float(rst.kernel_ms), 3), "resolve_ms": round(float(rst.resolve_ms), 3),
                                       "frame_bytes": int(W * H * 4),
                                       "into": "page-locked host memory (rt_host_alloc), PCIe-inclusive"}
                finally:
                    rt.host_free(buf)
            if not args.no_cpu:
                out["cpu_baseline"] = cpu_baseline(args.cpu_width, args.cpu_spp, args.depth, args.cpu_threads)
        print(json.dumps(out), flush=True)

    rt.shutdown()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
