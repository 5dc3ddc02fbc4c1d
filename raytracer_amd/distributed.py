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

import os
import time

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
            self.source_row32 = self.source_row.to(torch.int32)     # the same list for rt_deliver_rows
            self.frame = torch.empty((self.H, self.W, 3), dtype=torch.uint8, device=device)

    def gather(self, assemble=True):
        """Frame (H, W, 3) uint8 on rank 0, None elsewhere; with one rank, the tile itself.  ``assemble=False`` leaves
        the gathered tiles in ``buffer`` (rank 0 then reads them through ``source_row32``) and returns None."""
        if self.world == 1:
            return self.tile[: len(self.my_rows)]
        if self.stage_host:
            host = [self.torch.empty(self.tile.shape, dtype=self.torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
            self.dist.gather(self.tile.cpu(), host, dst=0, group=self.group)
            if self.rank == 0:
                self.buffer.copy_(self.torch.stack(host))
        else:
            self.dist.gather(self.tile, self.gathered if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank != 0 or not assemble:
            return None
        self.torch.index_select(self.buffer.view(self.world * self.max_rows, self.W, 3), 0, self.source_row, out=self.frame)
        return self.frame


def _overlap(torch, a, b, cycles=1_500_000):
    """True if spin kernels put on streams ``a`` and ``b`` run at the same time."""
    def run(streams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run([a, b])  # first use of either stream
    one = min(run([a]) for _ in range(3))
    two = min(run([a, b]) for _ in range(3))
    return two < 1.5 * one


def concurrent_streams(torch, device, n, candidates=12):
    """``n`` streams whose kernels run concurrently with one another.

    HIP binds a stream to one of a few hardware queues (4 per process by default) when the stream is first used, and
    two streams on one queue run their kernels in turn -- measured on MI355X: the first two side streams a process
    uses share a queue.  So the choice is made by measurement: a candidate stream is kept if a spin kernel on it
    overlaps one on every stream kept so far.
    """
    kept = []
    with torch.cuda.device(device):
        # the spin kernel of torch; without it -- or with RT_STREAM_PROBE=0, which keeps those launches out of a kernel
        # profile -- the first n streams are taken
        can_probe = hasattr(torch.cuda, "_sleep") and os.environ.get("RT_STREAM_PROBE") != "0"
        for _ in range(candidates):
            if len(kept) == n:
                break
            c = torch.cuda.Stream(device=device)
            if not can_probe or all(_overlap(torch, c, k) for k in kept):
                kept.append(c)
        while len(kept) < n:    # no more independent queues: share
            kept.append(torch.cuda.Stream(device=device))
    return kept


class _Pipeline:
    """One renderer handle, its stream and its gather buffers: frames of one pipeline run in order.

    Two buffer sets alternate from frame to frame, so the frame that is being copied to the host (``to_host``) is not
    the one the next frame of this pipeline renders and gathers into.
    """

    def __init__(self, renderer, stream, gatherers):
        self.r, self.stream, self.gs = renderer, stream, gatherers
        self.g = gatherers[0]
        self.used = False
        self.count = 0          # frames put on this pipeline
        self.host = None        # pinned host frames, one per buffer set (rank 0, to_host)
        self.copied = None      # events: the copy out of buffer set b has finished
        self.spans = []         # (first event of a queued frame on this stream, event behind its last kernel)


class DistributedRenderer:
    """Per-rank renderer writing into a preallocated device tile, then the gather to rank 0.

    ``pipelines`` > 1: queued frames (``enqueue``) alternate over that many independent renderer handles, each with
    its own stream, tile and gather buffers, so the first kernels of frame k+1 fill the GPU while the last, long
    work items of frame k drain (two pipelines: 10 % more frames per second on a full 4K frame, 41 % on an eighth
    of it; a third one did not help).  Frame k is complete when ``finish()`` returns.

    ``to_host``: on rank 0 every queued frame is delivered into pinned host memory (``host_frame``) by the library's
    row kernel (``rt_deliver_rows``) on a stream of its own, beside the kernels of the following frames -- the uint8
    frame the caller of the reference's ``render()`` ends up with (ray_tracer.py:436-439).  With several ranks that
    kernel reads the gathered tiles directly, so the frame is assembled on its way to the host and exists only there
    (``enqueue`` / ``render`` then return None for the device frame).  ``RT_DELIVER=memcpy`` selects the runtime's
    own copy instead (A/B measurements).
    """

    def __init__(self, renderer, width, height, group=None, stripe_rows=None, flags=0, pipelines=1, to_host=False):
        import torch

        from . import runtime

        if pipelines < 1:
            raise ValueError("pipelines must be >= 1")
        self.torch = torch
        self.r = renderer
        self.flags = int(flags)
        dev = torch.device("cuda", renderer.device)
        self.device = dev
        self.lanes = []
        self.to_host = bool(to_host)
        # the copy stream needs a hardware queue of its own as well: on a queue shared with a pipeline its copies would
        # wait behind that pipeline's kernels, and those behind the copy
        n_streams = pipelines + (1 if to_host else 0)
        streams = concurrent_streams(torch, dev, n_streams) if n_streams > 1 else [None]
        copy_stream = streams.pop() if to_host else None
        for i in range(pipelines):
            if i == 0:
                r = renderer
            else:
                if renderer._packed is None:
                    raise RuntimeError("DistributedRenderer: upload the scene (set_scene) before asking for pipelines")
                r = runtime.Renderer(renderer.device).set_scene(renderer._packed)
            sets = 2 if to_host else 1
            self.lanes.append(_Pipeline(r, streams[i], [TileGatherer(width, height, dev, group, stripe_rows) for _ in range(sets)]))
        g = self.lanes[0].g
        self.W, self.H, self.world, self.rank = g.W, g.H, g.world, g.rank
        self.stripe_rows, self.my_rows = g.stripe_rows, g.my_rows
        self.next_lane = 0
        self.copy_stream = None
        if self.to_host and self.rank == 0:
            self.copy_stream = copy_stream
            for lane in self.lanes:
                lane.host = [torch.empty((self.H, self.W, 3), dtype=torch.uint8).pin_memory() for _ in lane.gs]
                lane.copied = [torch.cuda.Event() for _ in lane.gs]
        self.last_host_frame = None
        self.copy_spans = []    # (event before, event behind) every delivery on the copy stream since the last report
        self.use_memcpy = os.environ.get("RT_DELIVER") == "memcpy"
        # footprint of the delivery kernel: it has to keep up with the frame rate and no more -- one GPU renders a 4K
        # frame in ~2.5 ms (8 workgroups deliver it in 0.8 ms), eight GPUs in a fraction of that (64: the PCIe rate)
        self.deliver_workgroups = min(64, 8 * self.world)
        self.deliver = True     # set False to leave the following frames on the device (same buffers, no host copy)

    def _frame(self, lane, seed, queued):
        torch = self.torch
        b = lane.count % len(lane.gs)
        g = lane.g = lane.gs[b]
        cur = torch.cuda.current_stream(self.device)
        if self.copy_stream is not None and lane.count >= len(lane.gs):
            cur.wait_event(lane.copied[b])      # this buffer set is still the source of an earlier copy
        lane.count += 1
        timing = queued and self.copy_stream is not None
        if timing:
            begin = torch.cuda.Event(enable_timing=True)
            begin.record(cur)
        call = lane.r.enqueue_rows_device if queued else lane.r.render_rows_device
        stats = call(self.W, self.H, 0, g.tile.data_ptr(), stream=cur.cuda_stream, seed=seed, stripe_rows=self.stripe_rows,
                     stripe_count=self.world, stripe_index=self.rank, flags=self.flags)
        to_host = self.copy_stream is not None and self.deliver
        frame = g.gather(assemble=not to_host or self.use_memcpy)
        if timing:
            done = torch.cuda.Event(enable_timing=True)
            done.record(cur)
            lane.spans.append((begin, done))
        if to_host:
            if not timing:
                done = torch.cuda.Event()
                done.record(cur)
            self.copy_stream.wait_event(done)
            if timing:
                c0 = torch.cuda.Event(enable_timing=True)
                c0.record(self.copy_stream)
            if self.use_memcpy:
                with torch.cuda.stream(self.copy_stream):
                    lane.host[b][: frame.shape[0]].copy_(frame, non_blocking=True)
            elif g.world == 1:
                lane.r.deliver_rows(lane.host[b].data_ptr(), g.tile.data_ptr(), len(g.my_rows), self.W * 3,
                                    workgroups=self.deliver_workgroups, stream=self.copy_stream.cuda_stream)
            else:
                lane.r.deliver_rows(lane.host[b].data_ptr(), g.buffer.data_ptr(), self.H, self.W * 3,
                                    src_row_ptr=g.source_row32.data_ptr(), workgroups=self.deliver_workgroups,
                                    stream=self.copy_stream.cuda_stream)
            lane.copied[b].record(self.copy_stream)
            if timing:
                c1 = torch.cuda.Event(enable_timing=True)
                c1.record(self.copy_stream)
                self.copy_spans.append((c0, c1))
            self.last_host_frame = lane.host[b]
        return frame, stats

    def prepare(self):
        """Render one frame on every pipeline that has not rendered yet (queue allocation, first-launch costs), so
        that none of it lands in a timed region.  Collective when world > 1: every rank calls it."""
        for i, lane in enumerate(self.lanes):
            if not lane.used:
                self.render(seed=0, pipeline=i)

    def render(self, seed=0, pipeline=0):
        """Render this rank's stripes and gather; returns (frame on rank 0 else None, this rank's stats)."""
        lane = self.lanes[pipeline]
        lane.used = True
        if lane.stream is None:
            out = self._frame(lane, seed, queued=False)
        else:
            with self.torch.cuda.stream(lane.stream):   # the handle's frames stay on one stream
                out = self._frame(lane, seed, queued=False)
            lane.stream.synchronize()
        if self.copy_stream is not None:
            self.copy_stream.synchronize()
        return out

    def enqueue(self, seed=0):
        """Put one frame (render + gather + assembly, and the copy to the host with ``to_host``) on the next pipeline's
        stream without waiting for it; returns the frame tensor on rank 0 (complete after ``finish()``; overwritten
        by a later frame of the same pipeline)."""
        lane = self.lanes[self.next_lane]
        lane.used = True
        self.next_lane = (self.next_lane + 1) % len(self.lanes)
        if lane.stream is None:
            return self._frame(lane, seed, queued=True)[0]
        with self.torch.cuda.stream(lane.stream):
            return self._frame(lane, seed, queued=True)[0]

    def finish(self):
        """Wait for every enqueued frame of this rank (kernels, gather, assembly, host copies); returns the summed
        stats dict.  A frame that outgrew the queues makes this raise ``runtime.RtError`` (code -5) after everything
        has drained: render that sequence again through ``render()``, which resizes the queues."""
        total = None
        error = None
        for lane in self.lanes:
            try:
                st = lane.r.wait()
            except Exception as e:      # keep draining the other pipelines: nothing may stay in flight
                error = error or e
                st = None
            if lane.stream is not None:
                lane.stream.synchronize()
            if st is None:
                continue
            if total is None:
                total = st
            else:
                for k, v in st.items():
                    total[k] = [a + b for a, b in zip(total[k], v)] if isinstance(v, list) else total[k] + v
        if self.copy_stream is not None:
            self.copy_stream.synchronize()
        if error is not None:
            raise error
        return total

    def delivery_report(self):
        """Where the deliveries of the frames queued since the last call ran, from events on the streams (call after
        ``finish()``): time the copy stream was busy per frame, and the share of that time during which a pipeline
        stream was rendering ANOTHER frame -- 1.0 means every copy ran beside kernels, 0.0 that the GPU did nothing
        else meanwhile.  None on ranks that deliver nothing."""
        spans, copies = [s for lane in self.lanes for s in lane.spans], self.copy_spans
        for lane in self.lanes:
            lane.spans = []
        self.copy_spans = []
        if not copies or not spans:
            return None
        t0 = spans[0][0]
        frames = sorted((t0.elapsed_time(a), t0.elapsed_time(b)) for a, b in spans)
        busy = hidden = 0.0
        for c0, c1 in copies:
            a, b = t0.elapsed_time(c0), t0.elapsed_time(c1)
            busy += b - a
            # union of the frames' render intervals, clipped to [a, b]; a frame whose kernels end at the copy's start is
            # the frame being copied (its interval ends where the copy begins and adds nothing)
            cover, edge = 0.0, a
            for fa, fb in frames:
                lo, hi = max(fa, edge), min(fb, b)
                if hi > lo:
                    cover += hi - lo
                    edge = hi
            hidden += cover
        n = len(copies)
        return {"frames": n, "copy_stream_busy_ms_per_frame": busy / n, "share_of_copy_time_beside_rendering": hidden / busy if busy > 0 else None,
                "copy": "memcpy" if self.use_memcpy else "rt_deliver_rows"}

    def host_frame(self):
        """Pinned host copy (numpy view) of the most recently queued frame on rank 0 (valid after ``finish()``)."""
        return None if self.last_host_frame is None else self.last_host_frame.numpy()
