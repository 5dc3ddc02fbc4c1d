"""rt_render_device_noalloc (include/rt_amd.h): the entry point that allocates no device memory.

Its output has to equal rt_render_device's on the same input -- same kernels, the queues in the library's static
scratch array instead of a hipMalloc'd arena.  Two calls in a row, the second with a smaller frame, so that what the
first call left in the scratch (accumulators, queue records, counters) is shown not to leak into the second; then
the documented limits.  Collected last on purpose (the driver runs pytest -x).
"""
import numpy as np
import pytest

from helpers import scene
from raytracer_amd import runtime

pytestmark = pytest.mark.gpu


def _run(r, torch, call, w, h, seed, **kw):
    dev = torch.device("cuda", r.device)
    f32 = torch.full((h, w, 3), -1.0, dtype=torch.float32, device=dev)
    u8 = torch.full((h, w, 3), 77, dtype=torch.uint8, device=dev)
    st = call(w, h, f32.data_ptr(), u8.data_ptr(), seed=seed, **kw)
    torch.cuda.synchronize()
    n = st["rows_rendered"]                                # the selected rows come packed at the start of the buffers
    assert np.all(u8[n:].cpu().numpy() == 77)              # and nothing is written behind them
    return f32[:n].cpu().numpy(), u8[:n].cpu().numpy(), st


COUNTERS = ("primary_rays", "secondary_rays", "depth_cut", "shaded_points", "shadow_rays", "rows_rendered")


def _assert_same(a, b):
    (f_a, u_a, st_a), (f_b, u_b, st_b) = a, b
    assert np.array_equal(u_a, u_b)                       # bytes: bit-exact
    # float32 image: the accumulators are float64 sums whose order is not fixed, 1e-13 relative; after the cast to
    # float32 two runs can differ in the last bit of a value
    assert np.allclose(f_a, f_b, rtol=0, atol=1e-4)
    assert np.array_equal(np.uint8(f_b), u_b)
    assert {k: st_a[k] for k in COUNTERS} == {k: st_b[k] for k in COUNTERS}


@pytest.mark.parametrize("name", ["pool", "transparency_v7", "cubes"])
def test_noalloc_equals_render_device_twice_second_smaller(name):
    import torch
    r = runtime.Renderer(0)
    try:
        r.set_scene(scene(name))
        cases = [dict(w=320, h=200, seed=3), dict(w=96, h=54, seed=4),          # the second call is the smaller one
                 dict(w=200, h=120, seed=5, stripe_rows=8, stripe_count=3, stripe_index=1, row_begin=0, row_end=120)]
        mine = [_run(r, torch, r.render_rows_device_noalloc, **c) for c in cases]    # back to back in the scratch
        theirs = [_run(r, torch, r.render_rows_device, **c) for c in cases]
        for a, b in zip(theirs, mine):
            _assert_same(a, b)
        assert mine[0][2]["primary_rays"] == 320 * 200 and mine[1][2]["primary_rays"] == 96 * 54
    finally:
        r.close()


def test_noalloc_without_an_arena_and_with_two_handles():
    """A fresh handle that never rendered through an allocating entry point; and a second handle of the same GPU,
    used after the first (the scratch is shared: calls must not overlap, back to back is fine)."""
    import torch
    dev = torch.device("cuda", 0)
    r1, r2 = runtime.Renderer(0), runtime.Renderer(0)
    try:
        r1.set_scene(scene("pool"))
        r2.set_scene(scene("transparency_original"))
        outs = {}
        for tag, r in (("a", r1), ("b", r2), ("a2", r1)):
            u8 = torch.zeros((64, 80, 3), dtype=torch.uint8, device=dev)
            r.render_rows_device_noalloc(80, 64, 0, u8.data_ptr(), seed=1)
            torch.cuda.synchronize()
            outs[tag] = u8.cpu().numpy()
        assert np.array_equal(outs["a"], outs["a2"])
        for tag, r in (("a", r1), ("b", r2)):
            u8 = torch.zeros((64, 80, 3), dtype=torch.uint8, device=dev)
            r.render_rows_device(80, 64, 0, u8.data_ptr(), seed=1)
            torch.cuda.synchronize()
            assert np.array_equal(outs[tag], u8.cpu().numpy())
    finally:
        r1.close(); r2.close()


def test_noalloc_rejects_what_does_not_fit():
    import torch
    r = runtime.Renderer(0)
    try:
        r.set_scene(scene("pool"))
        u8 = torch.zeros((512, 512, 3), dtype=torch.uint8, device=torch.device("cuda", 0))
        with pytest.raises(runtime.RtError) as e:
            r.render_rows_device_noalloc(512, 512, 0, u8.data_ptr())      # 262144 pixels > RT_NOALLOC_MAX_PIXELS
        assert e.value.code == -6 and "RT_NOALLOC_MAX_PIXELS" in str(e.value)
        st = r.render_rows_device_noalloc(512, 512, 0, u8.data_ptr(), row_begin=0, row_end=256)   # half of it fits
        assert st["primary_rays"] == 512 * 256
        st = r.render_rows_device(512, 512, 0, u8.data_ptr())             # and the handle still works as before
        assert st["primary_rays"] == 512 * 512
    finally:
        r.close()


def _free_bytes(torch):
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


@pytest.mark.parametrize("name", ["pool", "transparency_v7"])   # v7: a material reflects AND transmits, so the ray tree's
def test_noalloc_moves_no_device_memory(name):                  # stack of parked children comes out of the scratch too
    """What the entry point exists for: the device's free memory is the same before and after the call (a hipMalloc or
    hipFree anywhere below it -- a queue that grows, a row list, an event pool -- would move it)."""
    import torch
    dev = torch.device("cuda", 0)
    r = runtime.Renderer(0)
    try:
        r.set_scene(scene(name))
        u8 = torch.zeros((200, 320, 3), dtype=torch.uint8, device=dev)
        f32 = torch.zeros((200, 320, 3), dtype=torch.float32, device=dev)
        want = torch.zeros_like(u8)
        r.render_rows_device(320, 200, 0, want.data_ptr(), seed=9)            # the handle is warm: arena, row list, events
        r.render_rows_device_noalloc(320, 200, f32.data_ptr(), u8.data_ptr(), seed=8)
        free0 = _free_bytes(torch)
        for k, (w, h) in enumerate([(320, 200), (96, 54), (320, 200)]):
            r.render_rows_device_noalloc(w, h, f32.data_ptr(), u8.data_ptr(), seed=9)
            assert _free_bytes(torch) == free0, (k, w, h)
        assert np.array_equal(u8.cpu().numpy(), want.cpu().numpy())
        # a fresh handle that never allocated an arena: the first call may not allocate either
        r2 = runtime.Renderer(0)
        try:
            r2.set_scene(scene(name))
            free1 = _free_bytes(torch)
            r2.render_rows_device_noalloc(320, 200, f32.data_ptr(), u8.data_ptr(), seed=9)
            assert _free_bytes(torch) == free1
            assert np.array_equal(u8.cpu().numpy(), want.cpu().numpy())
        finally:
            r2.close()
    finally:
        r.close()


def test_noalloc_reports_a_ray_tree_that_outgrows_the_scratch_instead_of_growing(tmp_path):
    """A deep tree of reflecting and transmitting hits needs more shade-queue slots than the scratch holds for the
    frame: the call fails with RT_ERR_QUEUE, allocates nothing, and the allocating entry point renders the frame."""
    import torch
    import raytracer_amd as R
    from helpers import SCENES, write_scene
    import os
    lines = open(os.path.join(SCENES, "transparency_v7.txt")).read().splitlines()
    deep = [("set " + " ".join(ln.split()[1:5]) + " 14") if ln.split()[:1] == ["set"] else ln for ln in lines]
    sc = R.parse_scene_file(write_scene(tmp_path, "\n".join(deep) + "\n"))
    dev = torch.device("cuda", 0)
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        u8 = torch.zeros((256, 480, 3), dtype=torch.uint8, device=dev)
        free0 = _free_bytes(torch)
        with pytest.raises(runtime.RtError) as e:
            r.render_rows_device_noalloc(480, 256, 0, u8.data_ptr(), seed=2)
        assert e.value.code == -5, str(e.value)                   # RT_ERR_QUEUE
        assert _free_bytes(torch) == free0
        st = r.render_rows_device(480, 256, 0, u8.data_ptr(), seed=2)
        assert st["shaded_points"] > 4 * 480 * 256                # more hits than the scratch layout has slots for
    finally:
        r.close()
