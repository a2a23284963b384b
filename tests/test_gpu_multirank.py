"""N > 1 on the GPU: three processes share the one device of the test box, each renders its
(unequal) share of the row tiles with the HIP kernel and the rows are gathered to rank 0.  The
assembled image must equal a single full-frame call byte for byte (the counter RNG makes a pixel
independent of the deal), in both builds.  On an 8-GPU node the same code runs with one device
per rank and RCCL (bench.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_scene

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("build", [0, 1, 2])
def test_three_ranks_one_gpu(rt, tmp_path, build):
    out = str(tmp_path / "full.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "mp_gpu_worker.py"), out, str(build)]
    subprocess.run(cmd, check=True, timeout=240, cwd=ROOT)
    got = np.load(out)
    counts = np.load(out + ".counts.npy")
    assert counts.sum() == got.shape[0] and counts[2] > counts[0]          # the deal followed the weights
    rt.set_world(load_scene("sf"))
    W, H = 200, rt.height_for(200)
    ref = rt.render(rt.camera(W, H), W, H, 6, 8, build=build, seed=3, chunk_spp=4)["rgba"]
    assert np.array_equal(got, ref)


def test_bench_two_ranks_one_gpu(tmp_path):
    """bench.py's N = 2 flow end to end (re-deal after the first warm-up step included)."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--width", "640",
           "--spp", "40", "--chunk", "10", "--backend", "gloo", "--shared-gpu"]
    res = subprocess.run(cmd, check=True, timeout=240, cwd=ROOT, stdout=subprocess.PIPE)
    line = [l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and sum(out["config"]["rows_per_rank"]) == 360
    assert 0 < out["roofline"]["frac"] <= 1.0 and out["roofline"]["bound"] == "mfma"
    assert out["gather_verified"] is True          # rank 0 re-rendered rows of rank 1: equal to what the gather delivered


def test_bench_rccl_calls_with_one_rank():
    """The RCCL branch of bench.py's N > 1 path on the one GPU of the test box: a process group of ONE rank on the
    "nccl" backend (RCCL refuses two ranks on a device, not one) runs the same calls as the 8-GPU job:
    init_process_group with a device id, all_reduce and gather on device tensors, barrier."""
    import json
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "2", "--width", "640",
           "--spp", "40", "--chunk", "10", "--force-dist", "--no-extras"]
    res = subprocess.run(cmd, check=True, timeout=240, cwd=ROOT, stdout=subprocess.PIPE, env=env)
    out = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["dist_world"] == 1 and out["value"] > 0
    assert out["rows_per_rank"] == [360] and out["gather_verified"] is None        # no other rank's rows to re-render


@pytest.mark.parametrize("build", ["exact", "strict", "fast"])
def test_several_contexts_in_one_process(rt, build):
    """rt_init_devices: three contexts (sharing the one device of the test box) split a frame by
    rows, one host thread each.  Image, accumulators, hit indices and ray counts equal the
    single-context call; the deal adapts between frames without changing the image."""
    import raytracer_amd as m
    sph = load_scene("sf")
    W, H = 240, m.height_for(240)
    rt.set_world(sph)
    cam = rt.camera(W, H)
    one = rt.render(cam, W, H, 5, 20, build=build, seed=9, chunk_spp=2, want_hit=True, want_accum=True)
    try:
        m.init_devices([0, 0, 0])
        m.set_world(sph)
        for frame in range(3):                                    # later frames use measured speeds
            many = m.render(cam, W, H, 5, 20, build=build, seed=9, chunk_spp=2, want_hit=True, want_accum=True)
            assert np.array_equal(many["rgba"], one["rgba"])
            assert np.array_equal(many["accum"], one["accum"]) and np.array_equal(many["hit"], one["hit"])
            assert int(many["stats"].rays) == int(one["stats"].rays)
            assert int(many["stats"].samples) == W * H * 5
        # a caller's own partition is split further, not overridden
        part = np.zeros((H, W, 4), np.uint8)
        m.render(cam, W, H, 5, 20, build=build, seed=9, chunk_spp=2, part_index=1, part_count=2, out=part)
        rows = m.part_row_map(H, 16, 1, 2)
        assert np.array_equal(part[rows], one["rgba"][rows])
        other = np.setdiff1d(np.arange(H), rows)
        assert not part[other].any()
        # fewer tiles than contexts: two rows in all
        tiny = np.zeros((H, W, 4), np.uint8)
        m.render(cam, W, H, 5, 20, build=build, seed=9, chunk_spp=2, rows=[7, 100], out=tiny)
        assert np.array_equal(tiny[[7, 100]], one["rgba"][[7, 100]]) and not tiny[:7].any()
        # progressive frames over several contexts: the running sums stay with the rows' owners
        for f in range(3):
            prog = m.render_progressive(cam, W, H, 2, 20, f, build=build, seed=9, chunk_spp=2, want_accum=True)
        six = rt.render(cam, W, H, 6, 20, build=build, seed=9, chunk_spp=2, want_accum=True)
        assert np.array_equal(prog["rgba"], six["rgba"]) and np.array_equal(prog["accum"], six["accum"])
        with pytest.raises(m.RtError, match="one device"):
            m.render_device(cam, 0, W, H, 1, 5)
    finally:
        m.init(0)                                                  # back to the session fixture's state
