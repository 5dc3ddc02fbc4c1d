"""Frame sequences delivered to HOST memory with the copy of frame k hidden behind the rendering of frame k+1.

``rt_render`` (``include/rt_amd.h``) hands over pageable host buffers, so every frame pays its device-to-host copy
after its kernels.  A caller that renders many frames (an animation, a parameter sweep) does not have to: the
frames are enqueued on the render stream (``rt_render_enqueue``), each into its own device buffer of a small ring,
and a second stream copies a finished frame into pinned host memory while the next one renders.  The reference has
no counterpart -- its caller converts one array per process (``ray_tracer.py:436-439``).

PyTorch is plumbing here: pinned and device buffers, two streams, events.
"""
from __future__ import annotations


def _add_stats(a, b):
    if a is None or b is None:
        return a if b is None else dict(b)
    return {k: ([x + y for x, y in zip(v, b[k])] if isinstance(v, list) else v + b[k]) for k, v in a.items()}


class FrameStream:
    """Ring of ``depth`` (device uint8 frame, pinned host frame) pairs.

        fs = FrameStream(renderer, width, height)
        for seed in seeds:
            done = fs.submit(seed)            # None while the ring fills, then the OLDEST finished frame
            if done is not None: use(done)    # numpy (H, W, 3) uint8 view of pinned memory, valid until the
        for done in fs.drain(): use(done)     # next ``submit`` (or the next step of ``drain``)
        stats = fs.stats()

    With the default ring of three, one frame renders, one is being copied and one is in the caller's hands.
    ``submit`` blocks only when the frame it returns has not reached the host yet.

    Queued frames cannot grow the library's ray / shade queues (``rt_render_enqueue``), and a frame that outgrew them
    would be incomplete.  The ray tree of a scene does not depend on the jitter seed, so the FIRST frame of every scene
    goes through the synchronous entry point, which sizes the queues (and raises if that fails); the frames queued
    after it have the same tree and fit.  ``stats()`` still reports an overflow the library saw, as an error.
    """

    def __init__(self, renderer, width, height, depth=3, flags=0):
        import torch

        if depth < 2:
            raise ValueError("FrameStream needs a ring of at least 2 frames")
        if depth > 8:
            raise ValueError("FrameStream: at most 8 frames in flight (each holds a device and a pinned host frame)")
        self.torch = torch
        self.r = renderer
        self.W, self.H, self.depth, self.flags = int(width), int(height), int(depth), int(flags)
        dev = torch.device("cuda", renderer.device)
        self.device = dev
        shape = (self.H, self.W, 3)
        self.dev_frames = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.host_frames = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.rendered = [torch.cuda.Event() for _ in range(depth)]
        self.copied = [torch.cuda.Event() for _ in range(depth)]
        self.submitted = 0
        self.delivered = 0
        self._sized_for = None      # scene the queues were last sized for by a synchronous frame
        self._sync_stats = None     # statistics of the synchronous frames since the last stats()

    def _deliver(self):
        slot = self.delivered % self.depth
        self.copied[slot].synchronize()
        self.delivered += 1
        return self.host_frames[slot].numpy()

    def submit(self, seed=0):
        """Enqueue one frame; returns the oldest finished frame once ``depth - 1`` frames are in flight, else None."""
        torch = self.torch
        out = None
        if self.submitted - self.delivered >= self.depth - 1:
            out = self._deliver()
        slot = self.submitted % self.depth
        render_stream = torch.cuda.current_stream(self.device)
        if self.submitted >= self.depth:
            # the device frame of this slot is still the source of an earlier copy
            render_stream.wait_event(self.copied[slot])
        if self._sized_for != self.r._scene_token:
            # first frame of this scene: the synchronous call grows the queues to what the scene's ray tree needs
            st = self.r.render_rows_device(self.W, self.H, 0, self.dev_frames[slot].data_ptr(), stream=render_stream.cuda_stream,
                                           seed=seed, flags=self.flags)
            self._sync_stats = _add_stats(self._sync_stats, st)
            self._sized_for = self.r._scene_token
        else:
            self.r.enqueue_rows_device(self.W, self.H, 0, self.dev_frames[slot].data_ptr(), stream=render_stream.cuda_stream,
                                       seed=seed, flags=self.flags)
        self.rendered[slot].record(render_stream)
        self.copy_stream.wait_event(self.rendered[slot])
        # the library's row kernel, not the runtime's copy: it runs beside the next frame's kernels (rt_deliver_rows)
        self.r.deliver_rows(self.host_frames[slot].data_ptr(), self.dev_frames[slot].data_ptr(), self.H, self.W * 3,
                            stream=self.copy_stream.cuda_stream)
        self.copied[slot].record(self.copy_stream)
        self.submitted += 1
        return out

    def drain(self):
        """Yield the frames still in flight, oldest first."""
        while self.delivered < self.submitted:
            yield self._deliver()

    def stats(self):
        """Summed statistics of the frames submitted since the last call (waits for them)."""
        st = _add_stats(self.r.wait(), self._sync_stats)
        self._sync_stats = None
        return st
