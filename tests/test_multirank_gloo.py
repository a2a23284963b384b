"""The N>1 layout on CPU: two gloo ranks each produce their interleaved row tiles (with the CPU
checker standing in for the kernel), RowGather assembles them on rank 0, and the result must
equal a single-rank render byte for byte.  Exercises raytracer_amd.multigpu and the
rt_part_rows / rt_part_row_at arithmetic that bench.py uses with RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_camera_golden, load_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tile_rows, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracer_amd as rt
    from raytracer_amd.multigpu import RowGather
    from oracle import refcpu as rc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, W, H = load_camera_golden(100)
    sph = load_scene("sf")
    rows = [rt.part_row_map(H, tile_rows, i, world) for i in range(world)]
    rg = RowGather(rows, W, 4, torch.uint8, torch.device("cpu"), rank, world)
    part = rc.render(rc.camera(W, H), sph, W, H, 3, 6, rng=rc.RNG_COUNTER, seed=4, chunk_spp=2,
                     tile_rows=tile_rows, part_index=rank, part_count=world, threads=2)["rgba"]
    rg.mine[:len(rows[rank])] = torch.from_numpy(part[rows[rank]])     # compact, as rt_render_device writes
    full = rg.gather()
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,tile_rows", [(2, 16), (3, 8)])
def test_row_gather_two_ranks(tmp_path, world, tile_rows):
    from oracle import refcpu as rc
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), tile_rows, out), nprocs=world, join=True)
    _, W, H = load_camera_golden(100)
    ref = rc.render(rc.camera(W, H), load_scene("sf"), W, H, 3, 6, rng=rc.RNG_COUNTER, seed=4, chunk_spp=2)["rgba"]
    assert np.array_equal(np.load(out), ref)


def _worker_weighted(rank, world, port, weights, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from raytracer_amd.multigpu import RowGather, weighted_row_maps
    from oracle import refcpu as rc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, W, H = load_camera_golden(100)
    sph = load_scene("s3")
    # ranks agree on the deal the way bench.py does: each contributes its own speed to a sum
    speed = torch.zeros(world, dtype=torch.float64)
    speed[rank] = weights[rank]
    dist.all_reduce(speed, op=dist.ReduceOp.SUM)
    rows = weighted_row_maps(H, 8, speed.tolist())
    rg = RowGather(rows, W, 4, torch.uint8, torch.device("cpu"), rank, world)
    full = rc.render(rc.camera(W, H), sph, W, H, 2, 6, rng=rc.RNG_COUNTER, seed=9, threads=2)["rgba"]
    rg.mine[:len(rows[rank])] = torch.from_numpy(full[rows[rank]])      # this rank's rows only
    out = rg.gather()
    dist.barrier()
    if rank == 0:
        np.save(out_path, out.numpy())
    dist.destroy_process_group()


def test_weighted_row_deal_two_ranks(tmp_path):
    from oracle import refcpu as rc
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker_weighted, args=(2, _free_port(), [1.0, 0.55], out), nprocs=2, join=True)
    _, W, H = load_camera_golden(100)
    ref = rc.render(rc.camera(W, H), load_scene("s3"), W, H, 2, 6, rng=rc.RNG_COUNTER, seed=9)["rgba"]
    assert np.array_equal(np.load(out), ref)


def test_weighted_row_maps_properties():
    import raytracer_amd as rt
    from raytracer_amd.multigpu import weighted_row_maps
    for height, tile_rows in ((225, 16), (2160, 16), (56, 8), (7, 16)):
        for weights in ([1.0], [1, 1], [1, 1, 1, 1, 1, 1, 1, 1], [1.0, 0.9, 1.1, 0.5], [3, 1]):
            maps = weighted_row_maps(height, tile_rows, weights)
            allr = np.sort(np.concatenate(maps))
            assert np.array_equal(allr, np.arange(height))                 # a partition of the image
            assert all(np.all(np.diff(m) > 0) for m in maps if len(m) > 1)  # ascending
            if len(set(weights)) == 1:                                      # equal speeds = modulo deal
                for i, m in enumerate(maps):
                    assert np.array_equal(m, rt.part_row_map(height, tile_rows, i, len(weights)))
            ntiles = (height + tile_rows - 1) // tile_rows
            share = np.asarray(weights, float) / np.sum(weights)
            tiles = np.array([len(set(m // tile_rows)) for m in maps])
            assert np.all(np.abs(tiles - share * ntiles) <= 1.0 + 1e-9)      # within one tile of the ideal share
    with pytest.raises(ValueError):
        weighted_row_maps(100, 16, [1.0, 0.0])
