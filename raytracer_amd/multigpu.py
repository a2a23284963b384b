"""Row-tile gather for the one-process-per-GPU layout.

Each rank path-traces the row tiles t with t % world == rank into a COMPACT buffer
(rt_render_device with part_index/part_count); gather_rows() moves every rank's rows to
rank 0 with one torch.distributed gather (RCCL over xGMI with the "nccl" backend, gloo on
CPU in the tests) and scatters them into the full framebuffer.  There is no reduction:
row tiles are disjoint.  The reference has no counterpart (single process, Main.c:990-1054).
"""
import torch
import torch.distributed as dist


def weighted_row_maps(height, tile_rows, weights):
    """Deal the row tiles of an image to ranks in proportion to `weights` (e.g. measured GPU
    speeds), keeping every rank's tiles spread over the whole image (smooth weighted round robin:
    each tile goes to the rank with the largest accumulated credit).  Equal weights give the
    t % world == rank deal of rt_part_row_at.  Returns one ascending row array per rank."""
    import numpy as np
    world = len(weights)
    w = np.asarray(weights, dtype=np.float64)
    if world == 0 or not np.all(np.isfinite(w)) or np.any(w <= 0):
        raise ValueError("weights must be positive and finite")
    w = w / w.sum()
    credit = np.zeros(world)
    rows = [[] for _ in range(world)]
    ntiles = (height + tile_rows - 1) // tile_rows
    for t in range(ntiles):
        credit += w
        r = int(np.argmax(credit - 1e-12 * np.arange(world)))     # ties: lowest rank
        credit[r] -= 1.0
        rows[r].extend(range(t * tile_rows, min(height, (t + 1) * tile_rows)))
    return [np.asarray(r, dtype=np.int64) for r in rows]


class RowGather:
    def __init__(self, row_maps, width, channels, dtype, device, rank, world, via_host=False, always_collective=False):
        """row_maps[i] = image rows owned by rank i (ascending), e.g. raytracer_amd.part_row_map.
        via_host: the process group cannot move device tensors (gloo with GPU buffers, used to
        rehearse the N > 1 path on a one-GPU box): stage the rows through host memory.
        always_collective: call the process group's gather even when it has one rank (a one-GPU box can then
        run the RCCL calls of the N > 1 path: RCCL refuses two ranks on one device, not one)."""
        self.rank, self.world = rank, world
        self.collective = world > 1 or bool(always_collective)
        self.via_host = bool(via_host) and torch.device(device).type != "cpu"
        self.counts = [len(r) for r in row_maps]
        self.max_rows = max(self.counts) if self.counts else 0
        height = sum(self.counts)
        # every rank sends a buffer of the same shape: pad to the largest part.  Two send buffers
        # alternate, so that the frame being rendered never writes the rows a collective that is
        # still in flight reads; an event per buffer marks the end of the collective that used it.
        self._send = [torch.zeros((self.max_rows, width, channels), dtype=dtype, device=device) for _ in range(2)]
        self._cur = 0
        self._on_gpu = torch.device(device).type == "cuda"
        self._done = [None, None]
        self.full = torch.zeros((height, width, channels), dtype=dtype, device=device) if rank == 0 else None
        self.recv = ([torch.zeros_like(self._send[0]) for _ in range(world)]
                     if (rank == 0 and self.collective) else None)
        self.idx = [torch.as_tensor(r, dtype=torch.int64, device=device) for r in row_maps] if rank == 0 else None
        if self.via_host and self.collective:
            self.host_mine = torch.zeros(self._send[0].shape, dtype=dtype)
            self.host_recv = [torch.zeros_like(self.host_mine) for _ in range(world)] if rank == 0 else None

    @property
    def mine(self):
        """This rank's send buffer for the next gather() (compact rows, padded)."""
        ev = self._done[self._cur]
        if ev is not None:
            ev.synchronize()                   # the collective that last read this buffer has finished
            self._done[self._cur] = None
        return self._send[self._cur]

    def connect(self):
        """One gather of whatever the buffers hold: makes the backend set up its peer connections
        (RCCL opens them on first use) outside any timed region."""
        self._gather(self._send[self._cur])
        if self._on_gpu:
            torch.cuda.synchronize()

    def gather(self):
        """Collect self.mine from all ranks into self.full on rank 0; then switch send buffers."""
        full = self._gather(self._send[self._cur])
        if self._on_gpu:
            ev = torch.cuda.Event()
            ev.record()
            self._done[self._cur] = ev
        self._cur ^= 1
        return full

    def _gather(self, mine):
        if self.collective:
            if self.via_host:
                self.host_mine.copy_(mine)
                dist.gather(self.host_mine, self.host_recv, dst=0)
                if self.rank == 0:
                    for i in range(self.world):
                        self.recv[i].copy_(self.host_recv[i])
            else:
                dist.gather(mine, self.recv, dst=0)
            if self.rank == 0:
                for i in range(self.world):
                    self.full.index_copy_(0, self.idx[i], self.recv[i][:self.counts[i]])
        else:
            self.full.index_copy_(0, self.idx[0], mine[:self.counts[0]])
        return self.full
