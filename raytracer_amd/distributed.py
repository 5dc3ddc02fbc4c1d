"""Multi-GPU frames: interleaved row stripes per rank, one gather of the uint8 tiles to rank 0.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL on ROCm and
runs over xGMI inside a node, ``gloo`` is used by the CPU tests of the partition
logic).  The scene (a few KB) is replicated; stripe ``k`` of ``stripe_rows`` rows
goes to rank ``k % world``.  Pixels are independent and the jitter stream is keyed
on the pixel, not on evaluation order, so the assembled frame equals the
single-GPU frame exactly.  The only exchange is the gather of the finished tiles
(24.9 MB of uint8 at 4K in total): every sender has a direct xGMI link to rank 0,
so there is no ring and no reduction.

PyTorch is plumbing here (device buffers, streams, the process group); rendering
is done by ``librt_amd.so`` writing straight into the tensor's memory.
"""
from __future__ import annotations

import numpy as np

from . import packing


def stripe_rows_for(height, world, preferred=16):
    """Stripe height: at most ``preferred`` rows, small enough that every rank gets several stripes."""
    rows = max(1, min(preferred, height // max(1, 4 * world)))
    return rows


def rank_rows(height, world, rank, stripe_rows):
    return packing.selected_rows(0, height, 0, height, stripe_rows, world, rank)


def max_rows_per_rank(height, world, stripe_rows):
    return max(len(rank_rows(height, world, r, stripe_rows)) for r in range(world))


def assemble(tiles, height, width, world, stripe_rows, xp=np):
    """Rank 0: place each rank's packed rows at their image rows.  ``tiles[r]`` is (>=rows_r, W, 3)."""
    if xp is np:
        out = np.zeros((height, width, 3), dtype=tiles[0].dtype)
        for r in range(world):
            ys = rank_rows(height, world, r, stripe_rows)
            out[ys] = tiles[r][: len(ys)]
        return out
    import torch
    out = torch.empty((height, width, 3), dtype=tiles[0].dtype, device=tiles[0].device)
    for r in range(world):
        ys = rank_rows(height, world, r, stripe_rows)
        idx = torch.as_tensor(ys, dtype=torch.long, device=tiles[0].device)
        out.index_copy_(0, idx, tiles[r][: len(ys)])
    return out


class DistributedRenderer:
    """Per-rank renderer + preallocated device tiles + the gather."""

    def __init__(self, renderer, width, height, group=None, stripe_rows=None, flags=0):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.r = renderer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.W, self.H = int(width), int(height)
        self.flags = int(flags)
        self.stripe_rows = stripe_rows or stripe_rows_for(self.H, self.world)
        self.my_rows = rank_rows(self.H, self.world, self.rank, self.stripe_rows)
        self.max_rows = max_rows_per_rank(self.H, self.world, self.stripe_rows)
        dev = torch.device("cuda", renderer.device)
        self.tile = torch.zeros((self.max_rows, self.W, 3), dtype=torch.uint8, device=dev)
        self.gathered = None
        self.frame = None
        if self.rank == 0 and self.world > 1:
            self.gathered = [torch.zeros_like(self.tile) for _ in range(self.world)]
            self.index = [torch.as_tensor(rank_rows(self.H, self.world, r, self.stripe_rows), dtype=torch.long, device=dev)
                          for r in range(self.world)]
            self.frame = torch.empty((self.H, self.W, 3), dtype=torch.uint8, device=dev)

    def render(self, seed=0):
        """Render this rank's stripes and gather; returns (frame on rank 0 else None, this rank's stats)."""
        torch, dist = self.torch, self.dist
        stream = torch.cuda.current_stream(self.tile.device).cuda_stream
        stats = self.r.render_rows_device(self.W, self.H, 0, self.tile.data_ptr(), stream=stream, seed=seed,
                                          stripe_rows=self.stripe_rows, stripe_count=self.world, stripe_index=self.rank,
                                          flags=self.flags)
        if self.world == 1:
            return self.tile[: len(self.my_rows)], stats
        dist.gather(self.tile, self.gathered if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank != 0:
            return None, stats
        for r in range(self.world):
            self.frame.index_copy_(0, self.index[r], self.gathered[r][: self.index[r].numel()])
        return self.frame, stats
