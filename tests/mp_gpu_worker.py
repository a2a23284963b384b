"""Worker of test_gpu_multirank.py: one rank of an N-rank render on ONE GPU (every rank opens
device 0; gloo moves the rows through host memory, since RCCL refuses two ranks on a device).
Same calls as bench.py: weighted row deal agreed by all_reduce, rt_render_device into the
compact buffer, RowGather to rank 0.  Launched with torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path, build = sys.argv[1], int(sys.argv[2])
    import raytracer_amd as rt
    from raytracer_amd.multigpu import RowGather, weighted_row_maps
    from conftest import load_scene
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.zeros(1, device=dev)
    rt.init(0)
    rt.set_world(load_scene("sf"))
    W, H = 200, rt.height_for(200)
    cam = rt.camera(W, H)
    speed = torch.zeros(world, dtype=torch.float64)
    speed[rank] = 1.0 + 0.35 * rank                         # pretend the ranks differ in speed
    dist.all_reduce(speed, op=dist.ReduceOp.SUM)
    rows = weighted_row_maps(H, 8, speed.tolist())
    rg = RowGather(rows, W, 4, torch.uint8, dev, rank, world, via_host=True)
    for _ in range(2):                                      # second call reuses every buffer
        rt.render_device(cam, rg.mine.data_ptr(), W, H, 6, 8, build=build, seed=3, chunk_spp=4,
                         rows=rows[rank])
        full = rg.gather()
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.cpu().numpy())
        np.save(out_path + ".counts.npy", np.array([len(r) for r in rows]))
    rt.shutdown()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
