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


def source_rows(height, world, stripe_rows, max_rows):
    """For every image row: its index in the rank-major stack of tiles, ``rank * max_rows + position in the rank's tile``."""
    src = np.empty(height, dtype=np.int64)
    for r in range(world):
        ys = rank_rows(height, world, r, stripe_rows)
        src[ys] = r * max_rows + np.arange(len(ys))
    return src


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


class TileGatherer:
    """This rank's uint8 tile, the gather to rank 0 and the assembly of the frame there.

    Device-agnostic (``cuda`` with RCCL in production, ``cpu`` with gloo in the tests): whoever fills ``tile`` --
    the HIP renderer or, in the CPU tests, the oracle -- calls ``gather()`` afterwards.
    """

    def __init__(self, width, height, device, group=None, stripe_rows=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.W, self.H = int(width), int(height)
        self.stripe_rows = stripe_rows or stripe_rows_for(self.H, self.world)
        self.my_rows = rank_rows(self.H, self.world, self.rank, self.stripe_rows)
        self.max_rows = max_rows_per_rank(self.H, self.world, self.stripe_rows)
        self.tile = torch.zeros((self.max_rows, self.W, 3), dtype=torch.uint8, device=device)
        self.gathered = None
        self.frame = None
        # gloo moves host memory only: device tiles are staged through the host then (rehearsals of the multi-rank
        # path on a box with fewer GPUs than ranks; RCCL takes the device tensors as they are)
        self.stage_host = dist.is_initialized() and dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
        if self.rank == 0 and self.world > 1:
            # the tiles land in one buffer [world, max_rows, W, 3]; image row y is row source_row[y] of its flat view,
            # so the frame is assembled by ONE row gather instead of one scatter per rank
            self.buffer = torch.zeros((self.world,) + tuple(self.tile.shape), dtype=torch.uint8, device=device)
            self.gathered = [self.buffer[r] for r in range(self.world)]
            self.source_row = torch.as_tensor(source_rows(self.H, self.world, self.stripe_rows, self.max_rows),
                                              dtype=torch.long, device=device)
            self.frame = torch.empty((self.H, self.W, 3), dtype=torch.uint8, device=device)

    def gather(self):
        """Frame (H, W, 3) uint8 on rank 0, None elsewhere; with one rank, the tile itself."""
        if self.world == 1:
            return self.tile[: len(self.my_rows)]
        if self.stage_host:
            host = [self.torch.empty(self.tile.shape, dtype=self.torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(self.tile.cpu(), host, dst=0, group=self.group)
            if self.rank == 0:
                self.buffer.copy_(self.torch.stack(host))
        else:
            self.dist.gather(self.tile, self.gathered if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank != 0:
            return None
        self.torch.index_select(self.buffer.view(self.world * self.max_rows, self.W, 3), 0, self.source_row, out=self.frame)
        return self.frame


class DistributedRenderer(TileGatherer):
    """Per-rank renderer writing into the preallocated device tile, then the gather."""

    def __init__(self, renderer, width, height, group=None, stripe_rows=None, flags=0):
        import torch

        super().__init__(width, height, torch.device("cuda", renderer.device), group, stripe_rows)
        self.r = renderer
        self.flags = int(flags)

    def render(self, seed=0):
        """Render this rank's stripes and gather; returns (frame on rank 0 else None, this rank's stats)."""
        stream = self.torch.cuda.current_stream(self.tile.device).cuda_stream
        stats = self.r.render_rows_device(self.W, self.H, 0, self.tile.data_ptr(), stream=stream, seed=seed,
                                          stripe_rows=self.stripe_rows, stripe_count=self.world, stripe_index=self.rank,
                                          flags=self.flags)
        return self.gather(), stats

    def enqueue(self, seed=0):
        """Put one frame (render + gather + assembly) on the stream without waiting for it; returns the frame
        tensor on rank 0 (valid for later work on the same stream).  ``finish()`` collects the statistics."""
        stream = self.torch.cuda.current_stream(self.tile.device).cuda_stream
        self.r.enqueue_rows_device(self.W, self.H, 0, self.tile.data_ptr(), stream=stream, seed=seed,
                                   stripe_rows=self.stripe_rows, stripe_count=self.world, stripe_index=self.rank,
                                   flags=self.flags)
        return self.gather()

    def finish(self):
        """Wait for every enqueued frame of this rank; returns the summed stats dict."""
        return self.r.wait()
