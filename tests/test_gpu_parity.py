"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle and the reference's goldens.

Bar (north star): with the keyed jitter stream identical on both sides, >= 99.9 % of
pixels within 1 LSB per 8-bit channel; the float64 geometry chain bit-identical;
ray counters exactly equal.  Free-running mode: PSNR >= 40 dB against the
reference's own Mersenne-Twister renders.
"""
import json
import os

import numpy as np
import pytest

import raytracer_amd as R
from helpers import THREADS, golden_json, golden_npz, golden_png, grid, scene, write_scene
from helpers import grid as grid_points
from raytracer_amd import compare_images as CI
from raytracer_amd import packing, runtime

pytestmark = pytest.mark.gpu

LSB_BAR = 0.999


def counters_of(stats):
    return {"primary": stats["primary_rays"], "secondary": stats["secondary_rays"], "depth_cut": stats["depth_cut"],
            "shaded": stats["shaded_points"], "shadow": stats["shadow_rays"]}


def assert_parity(u8, ref_f64, bar=LSB_BAR):
    m = CI.metrics(np.uint8(ref_f64), u8)
    assert m["within_1lsb"] >= bar, m
    return m


@pytest.mark.parametrize("name,w,h", [("pool", 256, 256), ("cubes", 320, 180), ("transparency_v7", 320, 180),
                                      ("transparency_original", 160, 160), ("pool_n16", 96, 54)])
def test_frames_match_oracle(renderer, oracle, name, w, h):
    sc = scene(name)
    renderer.set_scene(sc)
    f32, u8, st = renderer.render_rows(w, h, seed=0)
    ref, cnt = oracle.render_frame(sc, w, h, seed=0, threads=THREADS)
    assert_parity(u8, ref)
    assert counters_of(st) == cnt
    assert np.array_equal(np.uint8(f32), u8)              # float image truncates to the uint8 image
    assert f32.min() >= 0.0 and f32.max() <= 255.0


@pytest.mark.parametrize("name,w,h", [("pool", 500, 500), ("cubes", 480, 270), ("transparency_v7", 480, 270),
                                      ("transparency_original", 256, 256)])
def test_frames_match_reference_goldens(renderer, name, w, h):
    key = f"{name}_{w}x{h}_seed0"
    gold = golden_png(f"frames/{key}.png")
    meta = golden_json(f"frames/{key}.json")
    renderer.set_scene(scene(name))
    _, u8, st = renderer.render_rows(w, h, seed=0)
    m = CI.metrics(gold, u8)
    assert m["within_1lsb"] >= LSB_BAR, m
    assert m["pass_stochastic"]
    assert counters_of(st) == {k: meta["counters"][k] for k in ("primary", "secondary", "depth_cut", "shaded", "shadow")}


@pytest.mark.parametrize("key", ["cubes_1920x1080", "transparency_v7_1920x1080", "pool_3840x2160", "pool_n16_3840x2160",
                                 "synthetic_10k_1k_3840x2160"])
def test_benchmark_configs_at_full_size_against_reference_pixels(renderer, key):
    """The benchmark configurations at their real resolution: the reference's colours at sparse pixels."""
    z = golden_npz(f"sparse/{key}.npz")
    name = key.rsplit("_", 1)[0]
    w, h = int(z["width"]), int(z["height"])
    renderer.set_scene(scene(name))
    _, u8, st = renderer.render_rows(w, h, seed=int(z["seed"]), want_f32=False)
    got = u8[z["y"], z["x"]]
    m = CI.metrics(np.uint8(z["rgb"]), got)
    assert m["within_1lsb"] >= LSB_BAR - 0.002, m   # 384..2048 pixels: allow 2 extra per 1000 for the small sample
    assert st["primary_rays"] == w * h


@pytest.mark.parametrize("name,w,h", [("pool", 64, 64), ("cubes", 96, 54), ("transparency_v7", 96, 54),
                                      ("transparency_original", 64, 64)])
def test_geometry_chain_is_bit_identical(renderer, oracle, name, w, h):
    """Every nearest-hit query of the ray tree: surface, t, hit point, normal -- all 64 bits."""
    sc = scene(name)
    renderer.set_scene(sc)
    hb = renderer.debug_hits(w, h)
    xs, ys = grid(w, h)
    _, _, oh = oracle.render_pixels(sc, w, h, xs, ys, seed=0, record_hits=True)
    assert hb["count"] == len(oh)
    want = {(o[0], o[1]): o for o in oh}
    for i in range(hb["count"]):
        o = want[(int(hb["pixel"][i]), int(hb["node"][i]))]
        assert o[2] == hb["surf"][i]
        assert o[3] == hb["t"][i]
        assert o[4] == tuple(hb["P"][i]) and o[5] == tuple(hb["n"][i])


def test_stripes_reproduce_the_full_frame_exactly(renderer):
    """Multi-GPU partition: any row-stripe split gives the single-GPU image bit for bit."""
    w, h = 200, 117
    renderer.set_scene(scene("transparency_v7"))
    _, full, st_full = renderer.render_rows(w, h, seed=9)
    total = 0
    for world, rows in ((3, 5), (8, 4)):
        out = np.zeros_like(full)
        total = 0
        for r in range(world):
            _, part, st = renderer.render_rows(w, h, seed=9, stripe_rows=rows, stripe_count=world, stripe_index=r)
            ys = packing.selected_rows(w, h, 0, h, rows, world, r)
            assert part.shape[0] == len(ys)
            out[ys] = part
            total += st["total_rays"]
        np.testing.assert_array_equal(out, full)
        assert total == st_full["total_rays"]
    _, band, _ = renderer.render_rows(w, h, seed=9, row_begin=40, row_end=77)
    np.testing.assert_array_equal(band, full[40:77])


def test_determinism_and_seed_sensitivity(renderer):
    w, h = 240, 135
    renderer.set_scene(scene("pool"))
    _, a, _ = renderer.render_rows(w, h, seed=1)
    _, b, _ = renderer.render_rows(w, h, seed=1)
    _, c, _ = renderer.render_rows(w, h, seed=2)
    m = CI.metrics(a, b)
    assert m["max_abs_diff"] <= 1 and m["within_1lsb"] == 1.0   # float64 atomics: order effects stay below 1 LSB
    m2 = CI.metrics(a, c)
    assert m2["max_abs_diff"] > 1 and m2["psnr"] > 40.0          # different jitter, same image statistically


@pytest.mark.parametrize("name,w,h", [("pool", 160, 160), ("cubes", 160, 90), ("transparency_v7", 160, 90)])
def test_free_running_mode_psnr_against_reference_mt_renders(renderer, name, w, h):
    """Stochastic mode: any seed of ours vs the reference's own np.random stream: PSNR >= 40 dB, MSE < 0.001."""
    gold = golden_png(f"frames_mt/{name}_{w}x{h}_mt1234.png")
    renderer.set_scene(scene(name))
    for seed in (0, 12345):
        _, u8, _ = renderer.render_rows(w, h, seed=seed)
        m = CI.metrics(gold, u8)
        assert m["pass_stochastic"], m


def test_general_kernel_agrees_with_register_resident_kernel(renderer, oracle):
    """RT_FLAG_GENERAL_SHADE forces the any-size shading kernel on a small scene."""
    import ctypes as C
    w, h = 128, 72
    for name in ("cubes", "transparency_v7"):
        sc = scene(name)
        renderer.set_scene(sc)
        p = packing.render_params(w, h, seed=0, flags=1)
        u8 = np.empty((h, w, 3), np.uint8)
        st = packing.RtStats()
        rc = renderer._lib.rt_render(renderer._h, C.byref(p), None, u8.ctypes.data, C.byref(st))
        assert rc == 0
        ref, _ = oracle.render_frame(sc, w, h, seed=0, threads=THREADS)
        assert_parity(u8, ref)


EDGE_HEADER = "cam 0 1 6  0 0.5 0  0 1 0  1.2 1.6\n"
EDGE_BODY = ("mtl 0.8 0.3 0.2  1 1 1  0.3 0.3 0.3  25 0\nmtl 0.2 0.5 0.9  0.5 0.5 0.5  0 0 0  8 0.6\n"
             "mtl 0.5 0.5 0.5  0.2 0.2 0.2  0.1 0.1 0.1  5 0\n"
             "pln 0 1 0 0 3\nsph -0.8 0.6 0 0.6 1\nbox 0.9 0.5 0.3 1.0 2\nsph 0.9 0.5 0.3 0.3 1\n")
EDGE_CASES = {
    "single_shadow_ray": EDGE_HEADER + "set 0.2 0.3 0.4 1 3\n" + EDGE_BODY + "lgt 2 4 3  1 1 1  1 0.7 1\nlgt -3 5 1  0.6 0.6 0.6  0.5 0.9 2\n",
    "point_light_no_shadow": EDGE_HEADER + "set 0.2 0.3 0.4 4 3\n" + EDGE_BODY + "lgt 2 4 3  1 1 1  1 0 1\nlgt -3 5 1  0.6 0.6 0.6  0.5 0.9 2\n",
    "depth_zero": EDGE_HEADER + "set 0.2 0.3 0.4 3 0\n" + EDGE_BODY + "lgt 2 4 3  1 1 1  1 0.7 1\n",
    "depth_one": EDGE_HEADER + "set 0.2 0.3 0.4 3 1\n" + EDGE_BODY + "lgt 2 4 3  1 1 1  1 0.7 1\n",
    "fractional_depth": EDGE_HEADER + "set 0.2 0.3 0.4 2 2.5\n" + EDGE_BODY + "lgt 2 4 3  1 1 1  1 0.7 1\n",
    "no_lights": EDGE_HEADER + "set 0.2 0.3 0.4 3 3\n" + EDGE_BODY,
    "no_surfaces": EDGE_HEADER + "set 0.2 0.3 0.4 3 3\nmtl 1 1 1 1 1 1 0 0 0 1 0\nlgt 2 4 3  1 1 1  1 0.7 1\n",
    "zero_shininess_and_tilted_plane": EDGE_HEADER + "set 0 0 0 3 4\nmtl 0.8 0.3 0.2  1 1 1  0.5 0.5 0.5  0 0\n"
                                       "mtl 0.3 0.3 0.9  1 1 1  0 0 0  12 0.5\npln 0.2 1 0.1 -0.3 1\nsph 0 0.8 0 0.7 2\nsph 0 0.8 0 0.3 1\n"
                                       "lgt 2 4 3  1 1 1  1 0.7 1.5\n",
    "nine_surfaces_general_kernel": EDGE_HEADER + "set 0.2 0.3 0.4 3 3\n" + EDGE_BODY +
                                    "sph -2 0.4 -1 0.4 2\nsph 2.2 0.4 -1 0.4 1\nbox 0 0.3 -2 0.6 1\nsph -0.3 0.25 1.5 0.25 3\nbox -1.6 0.3 1.2 0.6 2\n"
                                    "lgt 2 4 3  1 1 1  1 0.7 1\n",
}


@pytest.mark.parametrize("case", sorted(EDGE_CASES))
@pytest.mark.parametrize("w,h", [(97, 61), (1, 1)])
def test_edge_cases(renderer, oracle, tmp_path, case, w, h):
    sc = R.parse_scene_file(write_scene(tmp_path, EDGE_CASES[case]))
    renderer.set_scene(sc)
    _, u8, st = renderer.render_rows(w, h, seed=4)
    ref, cnt = oracle.render_frame(sc, w, h, seed=4, threads=4)
    m = CI.metrics(np.uint8(ref), u8)
    assert m["within_1lsb"] >= (LSB_BAR if w * h > 1000 else 1.0), (case, m)
    assert counters_of(st) == cnt, case


def test_render_is_a_drop_in(renderer, oracle, capsys):
    """render(scene_settings, width, height) -> (H, W, 3) float64 in [0, 255]; np.uint8() gives the image."""
    sc = scene("cubes")
    img = R.render(sc, 80, 45, seed=0)
    assert img.shape == (45, 80, 3) and img.dtype == np.float64
    assert img.min() >= 0 and img.max() <= 255
    ref, _ = oracle.render_frame(sc, 80, 45, seed=0, threads=4)
    assert_parity(np.uint8(img), ref)
    out = capsys.readouterr().out
    assert "Render time:" in out and "seconds" in out


def test_cli_writes_png(tmp_path, oracle):
    from PIL import Image
    from raytracer_amd import ray_tracer
    out = str(tmp_path / "out.png")
    ray_tracer.main([os.path.join(os.path.dirname(os.path.dirname(__file__)), "scenes", "pool.txt"), out, "--width", "64", "--height", "48"])
    got = np.array(Image.open(out))
    assert got.shape == (48, 64, 3)
    ref, _ = oracle.render_frame(scene("pool"), 64, 48, seed=0, threads=4)
    assert_parity(got, ref)


def test_invalid_arguments_are_reported(renderer):
    renderer.set_scene(scene("pool"))
    with pytest.raises(runtime.RtError):
        renderer.render_rows(0, 10)
    with pytest.raises(runtime.RtError):
        renderer.render_rows(10, 10, row_begin=5, row_end=20)
    with pytest.raises(runtime.RtError):
        renderer.render_rows(10, 10, stripe_count=2, stripe_index=2)
    fresh = runtime.Renderer(0)
    with pytest.raises(runtime.RtError) as e:
        fresh.render_rows(8, 8)
    assert e.value.code == -2
    fresh.close()


def test_headline_frame_properties(renderer):
    """pool N=16 at 3840x2160 (the bench workload): size-independent properties."""
    w, h = 3840, 2160
    renderer.set_scene(scene("pool_n16"))
    _, a, st = renderer.render_rows(w, h, seed=0, want_f32=False)
    assert st["primary_rays"] == w * h
    assert st["shadow_rays"] == st["shaded_points"] * 5 * 256
    assert st["shaded_points"] <= st["primary_rays"] + st["secondary_rays"]
    assert st["depth_cut"] > 0 and st["secondary_rays"] > 0
    # the image is mirror-symmetric in the scene's symmetry plane only statistically; check seeds agree instead
    _, b, _ = renderer.render_rows(w, h, seed=1, want_f32=False, row_begin=1000, row_end=1128)
    m = CI.metrics(a[1000:1128], b)
    assert m["psnr"] > 45.0   # N = 16: jitter noise is small


def test_row_bands_and_queue_growth(oracle, tmp_path, monkeypatch):
    """Frames larger than a band are rendered in row bands; a ray tree that outgrows the queues makes the
    library retry with fewer rows and, at one row, with larger queues -- the image does not change."""
    monkeypatch.setenv("RT_MAX_BAND_PIXELS", "2048")   # read by rt_create: bands of 2048 pixels
    monkeypatch.setenv("RT_TAIL_FROM", "0")            # one launch per level, so every level goes through the ray queues
    r = runtime.Renderer(0)
    try:
        sc = scene("transparency_v7")
        r.set_scene(sc)
        w, h = 160, 90
        _, u8, st = r.render_rows(w, h, seed=2)
        assert st["bands"] >= (w * h) // 2048
        ref, cnt = oracle.render_frame(sc, w, h, seed=2, threads=THREADS)
        assert_parity(u8, ref)
        assert counters_of(st) == cnt
        # binary ray tree: every hit spawns a reflection AND a transparency child, 12 levels deep
        text = ("cam 0 0 6  0 0 0  0 1 0  1 1\nset 0.1 0.1 0.1 1 12\nmtl 0.5 0.5 0.5  1 1 1  0.5 0.5 0.5  10 0.5\n"
                "sph 0 0 0 1 1\nsph 0 0 -3 1.5 1\nsph 2.2 0 0 1 1\nsph -2.2 0 0 1 1\npln 0 1 0 -1.2 1\npln 0 0 1 -6 1\n"
                "lgt 3 4 5  1 1 1  1 0.5 1\n")
        sc2 = R.parse_scene_file(write_scene(tmp_path, text))
        r.set_scene(sc2)
        w, h = 256, 8
        _, u8, st = r.render_rows(w, h, seed=0)
        ref, cnt = oracle.render_frame(sc2, w, h, seed=0, threads=THREADS)
        assert cnt["secondary"] > 200 * w * h        # really a deep tree
        assert counters_of(st) == cnt
        assert_parity(u8, ref, bar=0.995)            # 2048 pixels
    finally:
        r.close()


BINARY_TREE = ("cam 0 0 6  0 0 0  0 1 0  1 1\nset 0.1 0.1 0.1 2 9\nmtl 0.5 0.5 0.5  1 1 1  0.5 0.5 0.5  10 0.5\n"
               "mtl 0.7 0.2 0.2  1 1 1  0 0 0  10 0\nmtl 0.2 0.7 0.2  1 1 1  0.4 0.4 0.4  10 0\n"
               "sph 0 0 0 1 1\nsph 0 0 -3 1.5 1\nsph 2.2 0 0 1 3\nsph -2.2 0 0 1 1\npln 0 1 0 -1.2 2\npln 0 0 1 -6 1\nbox 0 1.8 0 0.8 1\n"
               "lgt 3 4 5  1 1 1  1 0.5 1\n")


@pytest.mark.parametrize("tail_from", ["0", "1", "3", "auto"])
def test_run_to_completion_kernel_matches_level_launches(oracle, tmp_path, monkeypatch, tail_from):
    """The deep levels of the ray tree run either as one launch per level or inside one launch that follows every
    ray to the recursion limit (RT_TAIL_FROM picks the depth of the switch; default: the camera rays already).  Same
    queries bit for bit, same counters, same image -- on a chain scene and on a binary tree that parks children."""
    if tail_from == "auto":
        monkeypatch.delenv("RT_TAIL_FROM", raising=False)
    else:
        monkeypatch.setenv("RT_TAIL_FROM", tail_from)
    r = runtime.Renderer(0)
    try:
        for sc, w, h in ((scene("pool"), 96, 54), (R.parse_scene_file(write_scene(tmp_path, BINARY_TREE)), 64, 36)):
            r.set_scene(sc)
            _, u8, st = r.render_rows(w, h, seed=3)
            ref, cnt = oracle.render_frame(sc, w, h, seed=3, threads=THREADS)
            assert counters_of(st) == cnt
            assert_parity(u8, ref, bar=0.999 if w * h > 4000 else 0.995)
            xs, ys = grid(w, h)
            _, _, oh = oracle.render_pixels(sc, w, h, xs, ys, seed=3, record_hits=True)
            hb = r.debug_hits(w, h, capacity=len(oh) + 64)
            assert hb["count"] == len(oh)
            want = {(o[0], o[1]): o for o in oh}
            for i in range(hb["count"]):
                o = want[(int(hb["pixel"][i]), int(hb["node"][i]))]
                assert (o[2], o[3]) == (hb["surf"][i], hb["t"][i])
                assert o[4] == tuple(hb["P"][i]) and o[5] == tuple(hb["n"][i])
    finally:
        r.close()


def _crowded_scene_text(n_spheres=150, n_boxes=50, seed=7):
    """More than 64 surfaces (so the library builds its uniform grid) packed densely enough that objects
    overlap, touch several cells each, and shadow rays cross several transparent ones."""
    state = [seed * 2654435761 % 2**32]

    def u():
        state[0] = (state[0] * 1664525 + 1013904223) % 2**32
        return (state[0] >> 8) / 16777216.0

    def q(lo, hi):
        return round((lo + (hi - lo) * u()) * 1024.0) / 1024.0

    lines = ["cam 0 3 7  0 1 -3  0 1 0  1.2 1.8", "set 0.3 0.4 0.6 2 3",
             "mtl 0.8 0.3 0.2  0.5 0.5 0.5  0 0 0  20 0", "mtl 0.2 0.7 0.3  0.5 0.5 0.5  0.4 0.4 0.4  10 0",
             "mtl 0.3 0.3 0.9  0.5 0.5 0.5  0 0 0  30 0.5", "mtl 0.7 0.7 0.2  1 1 1  0.3 0.3 0.3  15 0.4",
             "mtl 0.5 0.5 0.5  0.1 0.1 0.1  0 0 0  5 0", "pln 0 1 0 0 5", "pln 0 0 1 -9 1"]
    for _ in range(n_spheres):
        r = q(0.1, 0.5)
        lines.append(f"sph {q(-4, 4)!r} {round((r + q(0, 2.5)) * 1024) / 1024!r} {q(-8, 1)!r} {r!r} {1 + int(u() * 4)}")
    for _ in range(n_boxes):
        e = q(0.2, 0.8)
        lines.append(f"box {q(-4, 4)!r} {round((e / 2 + q(0, 2.5)) * 1024) / 1024!r} {q(-8, 1)!r} {e!r} {1 + int(u() * 4)}")
    lines += ["lgt 2 6 3  1 1 1  1 0.8 1.5", "lgt -3 5 -2  0.6 0.6 0.7  0.5 0.6 1"]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("grid", ["auto", "0"])
def test_grid_walk_matches_brute_force_on_a_crowded_scene(oracle, tmp_path, monkeypatch, grid):
    """202 surfaces: nearest hits and shadow rays go through the uniform grid (RT_GRID=0: every surface is tested,
    as the oracle does).  Same queries bit for bit, same counters, same image, with transparent surfaces that a
    shadow ray meets in several cells."""
    if grid == "auto":
        monkeypatch.delenv("RT_GRID", raising=False)
    else:
        monkeypatch.setenv("RT_GRID", grid)
    sc = R.parse_scene_file(write_scene(tmp_path, _crowded_scene_text()))
    assert len(sc.surfaces) == 202
    w, h = 128, 72
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        _, u8, st = r.render_rows(w, h, seed=1)
        ref, cnt = oracle.render_frame(sc, w, h, seed=1, threads=THREADS)
        assert counters_of(st) == cnt
        assert_parity(u8, ref)
        xs, ys = grid_points(w, h)
        _, _, oh = oracle.render_pixels(sc, w, h, xs, ys, seed=1, record_hits=True)
        hb = r.debug_hits(w, h, capacity=len(oh) + 64)
        assert hb["count"] == len(oh)
        want = {(o[0], o[1]): o for o in oh}
        for i in range(hb["count"]):
            o = want[(int(hb["pixel"][i]), int(hb["node"][i]))]
            assert (o[2], o[3]) == (hb["surf"][i], hb["t"][i])
            assert o[4] == tuple(hb["P"][i]) and o[5] == tuple(hb["n"][i])
    finally:
        r.close()


def _huge_scene_text(n_spheres=66000, n_boxes=4000, seed=11):
    """More than 65 536 surfaces (surface indices beyond 16 bits in every table that holds one): small spheres and boxes
    over a ground plane, a few transparent or reflecting; coordinates on a 2^-10 lattice so that the text round-trips."""
    state = [seed * 2654435761 % 2**32]

    def u():
        state[0] = (state[0] * 1664525 + 1013904223) % 2**32
        return (state[0] >> 8) / 16777216.0

    def q(lo, hi):
        return round((lo + (hi - lo) * u()) * 1024.0) / 1024.0

    lines = ["cam 0 6 8  0 0 -15  0 1 0  1.2 1.8", "set 0.2 0.3 0.5 2 2",
             "mtl 0.8 0.3 0.2  0.5 0.5 0.5  0 0 0  20 0", "mtl 0.2 0.7 0.3  0.5 0.5 0.5  0.3 0.3 0.3  10 0",
             "mtl 0.3 0.3 0.9  0.5 0.5 0.5  0 0 0  30 0.5", "mtl 0.6 0.6 0.6  0.1 0.1 0.1  0 0 0  5 0", "pln 0 1 0 0 4"]
    for _ in range(n_spheres):
        r = q(0.05, 0.15)
        lines.append(f"sph {q(-20, 20)!r} {round((r + q(0, 3)) * 1024) / 1024!r} {q(-45, 0)!r} {r!r} {1 + int(u() * 3)}")
    for _ in range(n_boxes):
        e = q(0.1, 0.3)
        lines.append(f"box {q(-20, 20)!r} {round((e / 2 + q(0, 3)) * 1024) / 1024!r} {q(-45, 0)!r} {e!r} {1 + int(u() * 3)}")
    lines += ["lgt 5 30 10  1 1 1  1 0.8 2", "lgt -8 25 -30  0.7 0.7 0.8  0.5 0.7 1.5"]
    return "\n".join(lines) + "\n"


def test_more_than_65536_surfaces(oracle, tmp_path):
    """70 001 surfaces: the surface index leaves 16 bits everywhere it is stored (shade records, cell lists, the
    per-light lists, the debug probe).  Queries bit for bit, counters and image against the oracle's file-order loop;
    a 2 x 2 shadow grid per light, so the any-size shading kernel walks the grid with batched shadow rays too."""
    sc = R.parse_scene_file(write_scene(tmp_path, _huge_scene_text()))
    assert len(sc.surfaces) == 70001
    w, h = 96, 54
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        _, u8, st = r.render_rows(w, h, seed=3)
        ref, cnt = oracle.render_frame(sc, w, h, seed=3, threads=THREADS)
        assert counters_of(st) == cnt
        assert_parity(u8, ref)
        xs, ys = grid_points(w, h)
        _, _, oh = oracle.render_pixels(sc, w, h, xs, ys, seed=3, record_hits=True)
        hb = r.debug_hits(w, h, capacity=len(oh) + 64)
        assert hb["count"] == len(oh)
        want = {(o[0], o[1]): o for o in oh}
        for i in range(hb["count"]):
            o = want[(int(hb["pixel"][i]), int(hb["node"][i]))]
            assert (o[2], o[3]) == (hb["surf"][i], hb["t"][i])
            assert o[4] == tuple(hb["P"][i]) and o[5] == tuple(hb["n"][i])
        assert int(hb["surf"].max()) >= 65536      # the frame does see surfaces beyond 16 bits
    finally:
        r.close()


@pytest.mark.parametrize("name", ["cubes", "transparency_v7"])
def test_grid_forced_on_small_scenes(oracle, monkeypatch, name):
    """RT_GRID=1 builds the grid for any scene; with RT_FLAG_GENERAL_SHADE the shadow rays walk it too."""
    import ctypes as C
    monkeypatch.setenv("RT_GRID", "1")
    sc = scene(name)
    w, h = 128, 72
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        ref, cnt = oracle.render_frame(sc, w, h, seed=0, threads=THREADS)
        for flags in (0, 1):
            p = packing.render_params(w, h, seed=0, flags=flags)
            u8 = np.empty((h, w, 3), np.uint8)
            st = packing.RtStats()
            assert r._lib.rt_render(r._h, C.byref(p), None, u8.ctypes.data, C.byref(st)) == 0
            assert_parity(u8, ref)
            assert (st.primary_rays, st.secondary_rays, st.shaded_points) == (cnt["primary"], cnt["secondary"], cnt["shaded"])
    finally:
        r.close()


def test_enqueued_frames_equal_synchronous_frames(renderer):
    """rt_render_enqueue / rt_render_wait: frames queued back to back (more than the 8 kept in flight) give the
    images and, summed, the statistics of the same frames rendered one call at a time."""
    import torch
    w, h = 160, 90
    renderer.set_scene(scene("transparency_v7"))
    seeds = list(range(11))
    want, total = [], None
    for s in seeds:
        _, u8, st = renderer.render_rows(w, h, seed=s, want_f32=False)
        want.append(u8)
        keys = ("primary_rays", "secondary_rays", "depth_cut", "shaded_points", "shadow_rays", "rows_rendered", "bands")
        total = {k: (total[k] if total else 0) + st[k] for k in keys}
    bufs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in seeds]
    stream = torch.cuda.current_stream().cuda_stream
    for s, b in zip(seeds, bufs):
        renderer.enqueue_rows_device(w, h, 0, b.data_ptr(), stream=stream, seed=s)
    st = renderer.wait()
    for b, u8 in zip(bufs, want):
        m = CI.metrics(u8, b.cpu().numpy())
        assert m["max_abs_diff"] <= 1 and m["within_1lsb"] == 1.0     # float64 atomics: order effects stay below 1 LSB
    assert {k: st[k] for k in total} == total
    assert st["ms_shade"] > 0 and st["ms_trace"] > 0
    assert renderer.wait()["primary_rays"] == 0                       # nothing left in flight
    # a frame that needs several row bands is refused, the synchronous call handles it
    with pytest.raises(runtime.RtError):
        renderer.enqueue_rows_device(3840, 2161, 0, bufs[0].data_ptr(), stream=stream)


def test_frame_stream_delivers_every_frame_to_the_host_in_order(renderer):
    """frames.FrameStream: copies overlapped with the next frame's kernels; every frame arrives, in order, equal to
    the synchronous frame of the same seed; ring sizes 2, 3 and 8 behave alike."""
    from raytracer_amd.frames import FrameStream
    w, h = 160, 90
    renderer.set_scene(scene("transparency_v7"))
    seeds = list(range(13))
    want = [renderer.render_rows(w, h, seed=s, want_f32=False)[1] for s in seeds]
    for depth in (2, 3, 8):
        fs = FrameStream(renderer, w, h, depth=depth)
        got = []
        for s in seeds:
            done = fs.submit(s)
            if done is not None:
                got.append(done.copy())
        got += [d.copy() for d in fs.drain()]
        st = fs.stats()
        assert len(got) == len(seeds) and st["primary_rays"] == len(seeds) * w * h
        for a, b in zip(want, got):
            m = CI.metrics(a, b)
            assert m["max_abs_diff"] <= 1 and m["within_1lsb"] == 1.0
        # consecutive seeds differ (soft-shadow jitter), so an out-of-order delivery would not pass unnoticed
        assert any(CI.metrics(want[i], want[i + 1])["max_abs_diff"] > 1 for i in range(len(seeds) - 1))
    with pytest.raises(ValueError):
        FrameStream(renderer, w, h, depth=9)


def test_deliver_rows_copies_and_gathers_into_pinned_memory(renderer):
    """rt_deliver_rows: plain copy and row gather, 16-byte / 4-byte / 1-byte row widths, any workgroup count; memory
    that is not pinned is refused (the kernel would fault on it)."""
    import torch
    dev = torch.device("cuda", renderer.device)
    g = torch.Generator().manual_seed(5)
    for rows, row_bytes in ((37, 11520), (50, 3 * 500), (9, 3 * 333), (1, 16), (2160, 3 * 3840)):
        src = torch.randint(0, 256, (rows + 5, row_bytes), dtype=torch.uint8, generator=g).to(dev)
        host = torch.zeros((rows, row_bytes), dtype=torch.uint8).pin_memory()
        for wgs in (0, 1, 64):
            host.zero_()
            renderer.deliver_rows(host.data_ptr(), src.data_ptr(), rows, row_bytes, workgroups=wgs)
            torch.cuda.synchronize()
            assert torch.equal(host, src[:rows].cpu())
        order = torch.randperm(rows + 5, generator=g)[:rows].to(torch.int32)
        host.zero_()
        renderer.deliver_rows(host.data_ptr(), src.data_ptr(), rows, row_bytes, src_row_ptr=order.to(dev).data_ptr(), workgroups=7)
        torch.cuda.synchronize()
        assert torch.equal(host, src.cpu()[order.long()])
    pageable = torch.zeros((4, 64), dtype=torch.uint8)
    with pytest.raises(runtime.RtError):
        renderer.deliver_rows(pageable.data_ptr(), src.data_ptr(), 4, 64)
    with pytest.raises(runtime.RtError):
        renderer.deliver_rows(host.data_ptr(), src.data_ptr(), 4, 64, workgroups=-1)
    renderer.deliver_rows(0, 0, 0, 64)          # nothing to do is not an error


def test_frame_stream_on_a_fresh_handle_with_a_deep_ray_tree(oracle, tmp_path):
    """A handle that never rendered synchronously, and a scene whose ray tree outgrows the initial queues (transparency
    scene with 9 levels): queued frames cannot grow the queues, so FrameStream sizes them with a synchronous first
    frame -- every delivered frame is complete (counters and bytes equal the synchronous frames, which the oracle pins)."""
    from raytracer_amd.frames import FrameStream
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "transparency_v7.txt")).read()
    lines = [("set " + " ".join(l.split()[1:5]) + " 9") if l.startswith("set") else l for l in text.splitlines()]
    sc = R.parse_scene_file(write_scene(tmp_path, "\n".join(lines) + "\n"))
    w, h = 288, 288                                    # more than the queues' floor of 65536 pixels
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        fs = FrameStream(r, w, h, depth=3)
        got = []
        for s in range(4):
            done = fs.submit(s)
            if done is not None:
                got.append(done.copy())
        got += [d.copy() for d in fs.drain()]
        st = fs.stats()
        want = [r.render_rows(w, h, seed=s, want_f32=False) for s in range(4)]
        assert st["shaded_points"] == sum(x[2]["shaded_points"] for x in want) and st["shaded_points"] > 4 * 4 * w * h
        for (_, a, _), b in zip(want, got):
            assert np.array_equal(a, b)
        ref, cnt = oracle.render_frame(sc, w, h, seed=0, threads=THREADS)
        assert_parity(got[0], ref)
        assert counters_of(want[0][2]) == cnt
    finally:
        r.close()


def test_two_pipelines_give_the_frames_of_one(renderer):
    """distributed.DistributedRenderer with two pipelines (two handles, two streams): queued frames alternate over
    them, overlap at their ends, and equal the frames of the synchronous call; statistics add up."""
    from raytracer_amd.distributed import DistributedRenderer
    w, h = 320, 180
    renderer.set_scene(scene("pool"))
    seeds = list(range(7))
    want, rays = [], 0
    for s in seeds:
        _, u8, st = renderer.render_rows(w, h, seed=s, want_f32=False)
        want.append(u8); rays += st["total_rays"]
    dr = DistributedRenderer(renderer, w, h, pipelines=2)
    first, st0 = dr.render(seed=0)
    assert CI.metrics(want[0], first.cpu().numpy())["within_1lsb"] == 1.0 and st0["total_rays"] > 0
    got = []
    for i in range(0, len(seeds), 2):        # a frame tensor is reused two frames later: collect in pairs
        frames = [dr.enqueue(seed=s) for s in seeds[i:i + 2]]
        st = dr.finish()
        got += [f.cpu().numpy() for f in frames]
        rays -= st["total_rays"]
    assert rays == 0
    assert len(got) == len(seeds)
    for u8, g in zip(want, got):
        m = CI.metrics(u8, g)
        assert m["max_abs_diff"] <= 1 and m["within_1lsb"] == 1.0
    a, b = dr.lanes[0].stream, dr.lanes[1].stream
    assert a is not None and b is not None and a.cuda_stream != b.cuda_stream
    with pytest.raises(ValueError):
        DistributedRenderer(renderer, w, h, pipelines=0)


def _random_scene_text(seed, family=0):
    """Small random scenes: 3..14 surfaces of all three kinds (some touching or overlapping), reflective and
    transparent materials, 1..3 lights with and without soft shadows, N in 1..4, depth 1..5.
    family 1: mostly boxes, N in 4..8, wider lights."""
    if family in (3, 4):
        return _edge_scene_text(seed, lights_on_surfaces=family == 4)
    rng = np.random.default_rng(seed)
    p_sphere, n_lo, n_hi, r_hi = (0.6, 1, 5, 2.5) if family == 0 else ((0.25, 4, 9, 4.0) if family == 1 else (0.6, 2, 5, 2.5))
    s_lo, s_hi = (2, 13) if family < 2 else (15, 41)   # family 2: 16..42 surfaces (the any-size kernel's reach mask)

    def q(lo, hi):
        return round(float(rng.uniform(lo, hi)) * 256.0) / 256.0

    n_mat = int(rng.integers(2, 6))
    lines = [f"cam {q(-1, 1)!r} {q(1, 3)!r} {q(5, 8)!r}  {q(-0.5, 0.5)!r} {q(0, 1)!r} 0  0 1 0  {q(1, 1.6)!r} {q(1.2, 2)!r}",
             f"set {q(0, 1)!r} {q(0, 1)!r} {q(0, 1)!r} {int(rng.integers(n_lo, n_hi))} {int(rng.integers(1, 6))}"]
    for _ in range(n_mat):
        refl = q(0.1, 0.6) if rng.random() < 0.4 else 0.0
        transp = q(0.2, 0.8) if rng.random() < 0.35 else 0.0
        shin = [0.0, 1.0, 8.0, 30.0, 100.0][int(rng.integers(0, 5))]
        lines.append(f"mtl {q(0.1, 1)!r} {q(0.1, 1)!r} {q(0.1, 1)!r}  {q(0, 1)!r} {q(0, 1)!r} {q(0, 1)!r}  "
                     f"{refl!r} {refl!r} {refl!r}  {shin!r} {transp!r}")
    lines.append(f"pln 0 1 0 {q(-1, 0)!r} {int(rng.integers(1, n_mat + 1))}")
    if rng.random() < 0.5:
        lines.append(f"pln {q(-0.3, 0.3)!r} {q(-0.2, 0.2)!r} 1 {q(-8, -4)!r} {int(rng.integers(1, n_mat + 1))}")
    for _ in range(int(rng.integers(s_lo, s_hi))):
        m = int(rng.integers(1, n_mat + 1))
        if rng.random() < p_sphere:
            lines.append(f"sph {q(-2.5, 2.5)!r} {q(-0.5, 2)!r} {q(-3, 2)!r} {q(0.2, 1.0)!r} {m}")
        else:
            lines.append(f"box {q(-2.5, 2.5)!r} {q(-0.5, 2)!r} {q(-3, 2)!r} {q(0.3, 1.4)!r} {m}")
    for _ in range(int(rng.integers(1, 4))):
        shadow = 0.0 if rng.random() < 0.2 else q(0.3, 1)
        lines.append(f"lgt {q(-4, 4)!r} {q(2, 6)!r} {q(-2, 5)!r}  {q(0.3, 1)!r} {q(0.3, 1)!r} {q(0.3, 1)!r}  {q(0, 1)!r} {shadow!r} {q(0.2, r_hi)!r}")
    return "\n".join(lines) + "\n"


def _edge_scene_text(seed, lights_on_surfaces=False):
    """Scenes built from the awkward cases: camera inside a transparent sphere or box, lights inside objects or
    exactly on a surface, point lights under a sample grid (radius 0), tangent / coincident / duplicated surfaces,
    raw plane normals that are not unit vectors, tilted crossing planes, very small and very large spheres, extreme
    shininess, zero colours, N up to 10 and depth up to 10."""
    rng = np.random.default_rng(seed)

    def q(lo, hi):
        return round(float(rng.uniform(lo, hi)) * 256.0) / 256.0

    def pick(*v):
        return v[int(rng.integers(0, len(v)))]

    cam = (q(-1, 1), q(0.5, 2.5), q(4, 7))
    lines = [f"cam {cam[0]!r} {cam[1]!r} {cam[2]!r}  {q(-0.5, 0.5)!r} {q(0, 1)!r} 0  {pick(0.0, 0.125)!r} 1 0  {q(1, 1.6)!r} {q(1.2, 2)!r}",
             f"set {q(0, 1)!r} {q(0, 1)!r} {q(0, 1)!r} {pick(1, 2, 3, 5, 10)} {pick(1, 2, 4, 10)}"]
    n_mat = 5
    for k in range(n_mat):
        refl = pick(0.0, 0.0, 0.25, 1.0)
        transp = pick(0.0, 0.0, 0.5, 1.0) if k else 0.5          # material 1 is always transparent
        shin = pick(0.0, 1.0, 2.0, 500.0)
        kd = pick(0.0, q(0.1, 1))
        lines.append(f"mtl {kd!r} {q(0, 1)!r} {q(0, 1)!r}  {pick(0.0, 1.0)!r} {q(0, 1)!r} {q(0, 1)!r}  "
                     f"{refl!r} {pick(0.0, refl)!r} {refl!r}  {shin!r} {transp!r}")
    m = lambda: int(rng.integers(1, n_mat + 1))
    # ground: a raw normal that is not a unit vector (the offset is NOT rescaled by the reference)
    lines.append(f"pln 0 {pick(1.0, 2.0, 0.5)!r} 0 {q(-1, 0)!r} {m()}")
    if rng.random() < 0.5:
        lines.append(f"pln {pick(1.0, -1.0, 0.5)!r} {pick(0.0, 0.25)!r} {pick(1.0, 3.0)!r} {q(-9, -5)!r} {m()}")
    shape = pick("sphere", "box", "none")
    if shape == "sphere":      # the camera sits inside a transparent sphere
        lines.append(f"sph {cam[0]!r} {cam[1]!r} {cam[2]!r} {pick(0.5, 1.0)!r} 1")
    elif shape == "box":
        lines.append(f"box {cam[0]!r} {cam[1]!r} {cam[2]!r} {pick(1.0, 2.5)!r} 1")
    cx, cy, cz, rr = q(-1, 1), q(0.5, 1.5), q(-2, 1), pick(0.25, 0.5, 0.75)
    lines.append(f"sph {cx!r} {cy!r} {cz!r} {rr!r} {m()}")
    extra = pick("duplicate", "tangent", "concentric", "face", "tiny", "huge")
    if extra == "duplicate":
        lines.append(f"sph {cx!r} {cy!r} {cz!r} {rr!r} {m()}")
    elif extra == "tangent":   # touches the first sphere in exactly one point
        lines.append(f"sph {cx + 2 * rr!r} {cy!r} {cz!r} {rr!r} {m()}")
    elif extra == "concentric":
        lines.append(f"sph {cx!r} {cy!r} {cz!r} {rr / 2!r} {m()}")
    elif extra == "face":      # two boxes sharing a face
        lines.append(f"box {cx!r} {cy + 2!r} {cz!r} 1.0 {m()}")
        lines.append(f"box {cx + 1.0!r} {cy + 2!r} {cz!r} 1.0 {m()}")
    elif extra == "tiny":
        lines.append(f"sph {q(-1, 1)!r} {q(0.5, 2)!r} {q(0, 3)!r} 0.015625 {m()}")
    else:
        lines.append(f"sph {q(-5, 5)!r} {q(0, 3)!r} -70 50 {m()}")
    bx, by, bz, be = q(-2, 2), q(0, 1.5), q(-2, 1), pick(0.5, 1.0, 1.5)
    lines.append(f"box {bx!r} {by!r} {bz!r} {be!r} {m()}")
    for _ in range(int(rng.integers(0, 4))):
        if rng.random() < 0.5:
            lines.append(f"sph {q(-2.5, 2.5)!r} {q(-0.5, 2)!r} {q(-3, 2)!r} {q(0.2, 1.0)!r} {m()}")
        else:
            lines.append(f"box {q(-2.5, 2.5)!r} {q(-0.5, 2)!r} {q(-3, 2)!r} {q(0.3, 1.4)!r} {m()}")
    for _ in range(int(rng.integers(1, 4))):
        # lights lying exactly ON a surface are a family of their own (4): whether a shadow ray towards one counts
        # that surface (t < dist with t == dist up to rounding) is decided by the last bit of the reference's float64
        # arithmetic -- its own image is salt-and-pepper noise there, and only a float64 evaluation reproduces it
        where = pick("free", "free", "in_sphere", "in_box", "on_box" if lights_on_surfaces else "free", "far")
        pos = {"free": (q(-4, 4), q(2, 6), q(-2, 5)), "in_sphere": (cx, cy, cz), "in_box": (bx, by, bz),
               "on_box": (bx, by + be / 2, bz), "far": (q(-400, 400), 900.0, q(-400, 400))}[where]
        shadow = pick(0.0, 0.5, 1.0, 1.0)
        radius = pick(0.0, 0.25, 1.0, 4.0)
        col = (pick(0.0, 1.0), q(0.3, 1), q(0.3, 1))
        lines.append(f"lgt {pos[0]!r} {pos[1]!r} {pos[2]!r}  {col[0]!r} {col[1]!r} {col[2]!r}  {pick(0.0, 0.5, 1.0)!r} {shadow!r} {radius!r}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_match_the_oracle(renderer, oracle, tmp_path, seed):
    """Geometry chain bit for bit, counters exactly, image within 1 LSB -- on scenes nobody tuned the kernels for."""
    sc = R.parse_scene_file(write_scene(tmp_path, _random_scene_text(1000 + seed)))
    w, h = 72, 48
    renderer.set_scene(sc)
    _, u8, st = renderer.render_rows(w, h, seed=seed)
    ref, cnt = oracle.render_frame(sc, w, h, seed=seed, threads=THREADS)
    assert counters_of(st) == cnt
    m = CI.metrics(np.uint8(ref), u8)
    assert m["pixels_beyond_1lsb"] <= 3, m            # 3456 pixels: 99.9 %
    xs, ys = grid(w, h)
    _, _, oh = oracle.render_pixels(sc, w, h, xs, ys, seed=seed, record_hits=True)
    hb = renderer.debug_hits(w, h, capacity=len(oh) + 64)
    assert hb["count"] == len(oh)
    want = {(o[0], o[1]): o for o in oh}
    for i in range(hb["count"]):
        o = want[(int(hb["pixel"][i]), int(hb["node"][i]))]
        assert (o[2], o[3]) == (hb["surf"][i], hb["t"][i])
        assert o[4] == tuple(hb["P"][i]) and o[5] == tuple(hb["n"][i])


@pytest.mark.parametrize("seed", range(16))
def test_edge_case_scenes_match_the_oracle(renderer, oracle, tmp_path, seed):
    """Scenes made of awkward cases (camera inside a glass sphere or box, lights inside objects, tangent / duplicate /
    concentric surfaces, raw non-unit plane normals, tiny and huge spheres, extreme shininess, N up to 10)."""
    sc = R.parse_scene_file(write_scene(tmp_path, _random_scene_text(7000 + seed, family=3)))
    w, h = 72, 48
    renderer.set_scene(sc)
    _, u8, st = renderer.render_rows(w, h, seed=seed)
    ref, cnt = oracle.render_frame(sc, w, h, seed=seed, threads=THREADS)
    assert counters_of(st) == cnt
    m = CI.metrics(np.uint8(ref), u8)
    assert m["pixels_beyond_1lsb"] <= 3, m


def test_far_side_of_a_glass_ball_is_decided_in_float64(renderer, oracle, tmp_path, monkeypatch):
    """Exit hits of a transparent sphere (and a camera inside one): shadow rays start on the sphere with c > 0 of a
    quantum or two and point back into it; whether the reference still resolves the near root depends on the last
    bits of its float64 quadratic (sphere.py:110-119).  Those samples are evaluated in float64 (k_shade_exact, listed
    in wave_light[7]); the image then agrees with the oracle everywhere, and the pair list grows on demand."""
    text = ("cam 0 1.2 5  0 1 0  0 1 0  1.4 1.6\nset 0.2 0.3 0.5 4 4\n"
            "mtl 0.6 0.7 0.8  1 1 1  0 0 0  8 0.6\nmtl 0.7 0.6 0.4  0.5 0.5 0.5  0.2 0.2 0.2  30 0\n"
            "pln 0 1 0 -0.5 2\nsph 0 1.2 5 1.0 1\nsph 0.4 1.0 0.5 1.1 1\nsph -1.8 0.7 -1 0.6 2\n"
            "lgt 2 4 6  1 0.9 0.8  1 0.8 1.5\nlgt -3 3 2  0.6 0.7 1  0.8 1 0.7\nlgt 0 2 -4  0.9 0.9 0.9  1 0.6 2\n")
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    renderer.set_scene(sc)
    w, h = 96, 64
    ref, cnt = oracle.render_frame(sc, w, h, seed=4, threads=THREADS)
    for flags in (0, 1):
        _, u8, st = renderer.render_rows(w, h, seed=4, flags=flags)
        assert counters_of(st) == cnt
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] == 0, (flags, m)
        if flags == 0:
            pairs, fast_u8 = st["wave_light"][7], u8
            assert pairs > 1000                         # pairs handed to the float64 kernel
    # a frame that needs more pairs than the list holds: the call enlarges the list and renders again, same image
    monkeypatch.setenv("RT_EXACT_LIST", "500")                 # read by rt_create
    small = runtime.Renderer(0)
    small.set_scene(sc)
    _, again, st2 = small.render_rows(w, h, seed=4)
    assert st2["wave_light"][7] == pairs > 500 and counters_of(st2) == cnt
    m = CI.metrics(fast_u8, again)
    assert m["max_abs_diff"] <= 1 and m["within_1lsb"] == 1.0
    # queued frames cannot be rendered again by the library: the overflow is reported at the wait
    import torch
    fresh = runtime.Renderer(0)
    fresh.set_scene(sc)
    buf = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    fresh.enqueue_rows_device(w, h, 0, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, seed=4)
    with pytest.raises(runtime.RtError):
        fresh.wait()
    small.close(); fresh.close()


def test_repeated_surface_lines_share_the_own_surface_rule(renderer, oracle, tmp_path):
    """A sphere, a box and a plane each listed twice (with different transparent materials): a shadow ray that starts
    on one starts on its twin too, and the reference's float64 arithmetic treats both alike; so does the library
    (twin groups), in the register-resident kernels and in the any-size one."""
    text = ("cam 0.2 1.3 5.5  0.3 0.8 0  0 1 0  1.3 1.5\nset 0.1 0.5 0.3 4 3\n"
            "mtl 0.2 0.6 0.3  0.4 0.3 0.2  0 0 0  10 0.5\nmtl 0.9 0.3 0.5  0.8 0.5 0.5  0 0 0  2 0.25\n"
            "mtl 0.5 0.5 0.5  1 1 1  0.3 0.3 0.3  50 0\n"
            "pln 0 1 0 -0.3 3\npln 0 1 0 -0.3 1\nsph -0.6 1.1 0.1 0.75 1\nsph -0.6 1.1 0.1 0.75 2\n"
            "box 1.2 0.6 -0.8 1.1 2\nbox 1.2 0.6 -0.8 1.1 1\n"
            "lgt -3.5 3.6 0.6  1 0.6 0.7  1 1 3\nlgt 2 4 3  0.5 0.7 1  0.5 0.8 1\n")
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    renderer.set_scene(sc)
    w, h = 96, 64
    ref, cnt = oracle.render_frame(sc, w, h, seed=9, threads=THREADS)
    for flags in (0, 1):
        _, u8, st = renderer.render_rows(w, h, seed=9, flags=flags)
        assert counters_of(st) == cnt
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] == 0, (flags, m)


@pytest.mark.parametrize("n,rec,lights,w,h", [(17, 4, 2, 64, 40), (7, 31, 1, 48, 32), (300, 2, 1, 16, 10), (3, 3, 12, 64, 40),
                                              (255, 2, 1, 16, 10), (1, 10, 3, 96, 64)])
def test_unusual_parameters(renderer, oracle, tmp_path, n, rec, lights, w, h):
    """Shadow grids that are not a power of two, beyond the register-resident kernel's 255, the recursion limit of 31
    with mirrors facing each other, a dozen lights, single-ray shadows: counters exact, image within 1 LSB everywhere."""
    rng = np.random.default_rng(5)
    lgt = "".join(f"lgt {rng.uniform(-4, 4):.3f} {rng.uniform(2, 5):.3f} {rng.uniform(-2, 5):.3f}  1 0.9 0.8  1 "
                  f"{rng.uniform(0.3, 1):.2f} {rng.uniform(0.3, 2):.2f}\n" for _ in range(lights))
    text = (f"cam 0 1.2 5  0 0.8 0  0 1 0  1.4 1.6\nset 0.2 0.3 0.5 {n} {rec}\n"
            "mtl 0.6 0.7 0.8  1 1 1  0.9 0.9 0.9  8 0.5\nmtl 0.7 0.6 0.4  0.5 0.5 0.5  0.8 0.8 0.8  30 0\n"
            "pln 0 1 0 -0.5 2\npln 0 0 1 -4 2\nsph 0.4 1.0 0.5 1.1 1\nsph -1.8 0.7 -1 0.6 2\nbox 1.5 0.2 1.5 1.0 2\n" + lgt)
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    renderer.set_scene(sc)
    ref, cnt = oracle.render_frame(sc, w, h, seed=2, threads=THREADS)
    for flags in (0, 1):
        _, u8, st = renderer.render_rows(w, h, seed=2, flags=flags)
        assert counters_of(st) == cnt
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] == 0, (flags, m)


def test_twelve_surfaces_use_the_wide_register_resident_kernels(renderer, oracle, tmp_path):
    """9..16 surfaces: the register-resident kernels in their 16-slot instantiation (classified, brute force) and the
    any-size kernel give the oracle's image and counters."""
    sc = R.parse_scene_file(write_scene(tmp_path, _crowded_scene_text(8, 2, seed=3).replace("set 0.3 0.4 0.6 2 3", "set 0.3 0.4 0.6 4 3")))
    assert len(sc.surfaces) == 12
    w, h = 128, 72
    renderer.set_scene(sc)
    ref, cnt = oracle.render_frame(sc, w, h, seed=2, threads=THREADS)
    classes = {}
    for flags in (0, 2, 1):
        _, u8, st = renderer.render_rows(w, h, seed=2, flags=flags)
        assert counters_of(st) == cnt
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] == 0, (flags, m)
        classes[flags] = st["wave_light"]
    assert sum(classes[0][:5]) > 0 and sum(classes[1][:5]) > 0      # classified by the fast kernel / by the any-size kernel
    assert sum(classes[1][:6]) == sum(classes[0][:6])               # ... over the same (wave, light) pairs
    assert sum(classes[2][:5]) == 0 and classes[2][5] > 0           # brute force: every pair through the generic loop


@pytest.mark.parametrize("seed", [5000, 5008, 5061, 5074, 5080, 5086, 5105, 5108])
def test_lights_lying_on_box_faces(renderer, oracle, tmp_path, seed):
    """A shadow ray towards a light that lies ON a surface meets that surface at t = dist up to rounding; the reference
    decides `t < dist` (ray_tracer.py:43, :71) by the last bit of its float64 arithmetic, so its image is noise there.
    Pairs with such a light are shaded with those surfaces tested in float64, the reference's operations repeated."""
    sc = R.parse_scene_file(write_scene(tmp_path, _random_scene_text(seed, family=4)))
    w, h = 96, 64
    renderer.set_scene(sc)
    ref, cnt = oracle.render_frame(sc, w, h, seed=seed - 5000, threads=THREADS)
    _, u8, st = renderer.render_rows(w, h, seed=seed - 5000)
    assert counters_of(st) == cnt
    m = CI.metrics(np.uint8(ref), u8)
    assert m["pixels_beyond_1lsb"] <= 2, m


@pytest.mark.parametrize("n,radius", [(1, 0.5), (3, 0.0), (4, 1.5)])
def test_lights_on_a_ceiling_a_ball_and_a_box(renderer, oracle, tmp_path, n, radius):
    """Lamps placed exactly on the ceiling plane, on a sphere and on a box face, single-ray and sampled: register-
    resident kernels, any-size kernel and brute force all hand these pairs to the float64 path."""
    text = (f"cam 0 1.5 6  0 1.2 0  0 1 0  1.4 1.6\nset 0.2 0.3 0.5 {n} 3\n"
            "mtl 0.7 0.7 0.7  0.3 0.3 0.3  0 0 0  10 0\nmtl 0.8 0.4 0.3  1 1 1  0.3 0.3 0.3  40 0\n"
            "mtl 0.3 0.5 0.8  1 1 1  0 0 0  20 0.5\n"
            "pln 0 1 0 0 1\npln 0 -1 0 -4 1\npln 0 0 1 -5 1\n"
            "sph -1.5 1 0 1 2\nsph 1.6 0.8 0.5 0.8 3\nbox 0 0.5 -2 1 2\n"
            f"lgt 0.5 4 1  1 0.9 0.8  1 0.8 {radius}\n"          # on the ceiling plane y = 4
            f"lgt -1.5 2 0  0.5 0.6 1  0.7 1 {radius}\n"          # on top of the first sphere
            f"lgt 0 1 -2  0.9 0.5 0.5  1 0.6 {radius}\n")         # on the top face of the box
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    renderer.set_scene(sc)
    w, h = 96, 64
    ref, cnt = oracle.render_frame(sc, w, h, seed=6, threads=THREADS)
    for flags in (0, 1, 2):
        _, u8, st = renderer.render_rows(w, h, seed=6, flags=flags)
        assert counters_of(st) == cnt
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] <= 1, (flags, m)
        assert st["wave_light"][7] >= 3 * st["shaded_points"] * 0.99     # every (hit, light) pair went the float64 way


@pytest.mark.parametrize("n,radius,grid", [(1, 0.5, "auto"), (3, 1.0, "auto"), (3, 1.0, "0")])
def test_lights_on_surfaces_of_a_crowded_scene(oracle, tmp_path, monkeypatch, n, radius, grid):
    """The same with more than 64 surfaces (uniform grid, or every surface tested with RT_GRID=0): a lamp on the back
    wall, one on a sphere in the crowd and one on a box face; the other light of the scene takes the ordinary path."""
    monkeypatch.setenv("RT_GRID", grid)
    text = _crowded_scene_text(60, 20, seed=9).replace("set 0.3 0.4 0.6 2 3", f"set 0.3 0.4 0.6 {n} 3")
    text += ("sph 1.5 1 2 0.5 2\nbox -2 0.5 2 1 1\n"
             f"lgt 0.5 3 -9  1 0.9 0.8  1 0.8 {radius}\n"         # on the wall z = -9
             f"lgt 1.5 1.5 2  0.5 0.6 1  0.7 1 {radius}\n"         # on top of the added sphere
             f"lgt -2 1 2  0.9 0.5 0.5  1 0.6 {radius}\n")         # on the top face of the added box
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    assert len(sc.surfaces) == 84 and len(sc.lights) == 5
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        w, h = 96, 64
        ref, cnt = oracle.render_frame(sc, w, h, seed=6, threads=THREADS)
        _, u8, st = r.render_rows(w, h, seed=6)
        assert counters_of(st) == cnt
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] <= 1, m
        assert st["wave_light"][7] >= 3 * st["shaded_points"] * 0.99     # three of the five lights went the float64 way
    finally:
        r.close()


def test_pair_list_grows_when_every_hit_is_listed(oracle, tmp_path, monkeypatch):
    """A lamp on the ceiling plane lists every (hit, light) pair; with a list far too short the call counts the
    demand, rebuilds the list and renders again: the oracle's image, the same counters."""
    text = ("cam 0 1.5 6  0 1.2 0  0 1 0  1.4 1.6\nset 0.2 0.3 0.5 3 3\n"
            "mtl 0.7 0.7 0.7  0.3 0.3 0.3  0 0 0  10 0\nmtl 0.3 0.5 0.8  1 1 1  0.2 0.2 0.2  20 0.5\n"
            "pln 0 1 0 0 1\npln 0 -1 0 -4 1\nsph -1 1 0 1 2\nbox 1 0.5 -1 1 1\n"
            "lgt 0.5 4 1  1 0.9 0.8  1 0.8 1.5\nlgt -3 3 3  0.5 0.6 1  0.7 1 1\n")
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    monkeypatch.setenv("RT_EXACT_LIST", "300")
    r = runtime.Renderer(0)
    try:
        r.set_scene(sc)
        w, h = 96, 64
        ref, cnt = oracle.render_frame(sc, w, h, seed=1, threads=THREADS)
        _, u8, st = r.render_rows(w, h, seed=1)
        assert counters_of(st) == cnt
        assert st["wave_light"][7] >= st["shaded_points"] > 300
        m = CI.metrics(np.uint8(ref), u8)
        assert m["pixels_beyond_1lsb"] == 0, m
    finally:
        r.close()


def test_ground_plane_towards_the_horizon(renderer, oracle, tmp_path):
    """Hit points tens of thousands of units away (a level camera over an infinite plane): their shadow rays
    graze the objects near the origin from afar, where the literal FP32 sphere quadratic is pure cancellation."""
    text = ("cam 0 1.7 6  0.4 1.7172 0  0 1 0  1.6 1.5\nset 0.6 0.7 0.9 2 3\n"
            "mtl 0.6 0.5 0.7  0.8 0.7 0.8  0.2 0.2 0.2  30 0.5\nmtl 0.5 0.7 0.4  0.8 0.1 0.9  0.5 0.5 0.5  0 0\n"
            "pln 0 1 0 -0.25 2\nsph 1.2 1.4 -2.8 0.65 1\nsph 1.7 0.75 1.7 0.28 2\nbox -1.9 -0.4 -1.6 0.35 1\n"
            "sph -1.25 1.3 -2.3 0.66 1\nbox -0.35 1.8 -1.3 0.5 2\n"
            "lgt -1.8 4.2 0.75  0.8 0.5 0.35  0.5 0.5 0.9\nlgt -0.8 5.3 4.2  0.5 0.8 0.4  0.65 0.7 1.5\n")
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    w, h = 160, 96
    renderer.set_scene(sc)
    ref, cnt = oracle.render_frame(sc, w, h, seed=0, threads=THREADS)
    _, _, oh = oracle.render_pixels(sc, w, h, *grid(w, h), seed=0, record_hits=True)
    assert sum(o[2] == 0 and o[3] > 1e4 for o in oh) > 50    # a row of the plane is hit 30 000 units away
    for flags in (0, 1):
        p_u8 = renderer.render_rows(w, h, seed=0, flags=flags)[1]
        assert_parity(p_u8, ref)


def test_two_rank_bench_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver types it, with no launcher in the command line: the script starts
    its two ranks itself.  Rehearsed on one GPU: both ranks share the device and gather over gloo
    (RT_BENCH_REHEARSE=1).  Stripes, gather, assembly, host delivery, the reductions and the JSON line are the code
    the multi-GPU run executes; the ray count must be the whole frame's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RT_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["unit"] == "Mrays/s" and d["scaling"] == "strong"
    assert d["config"]["rays_per_frame"] == 14358303438.0      # both halves of the 4K pool N = 16 frame
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["host_frame_checked"] is True           # the assembled frame reached pinned host memory
    assert d["device_resident"]["ms_per_step"] > 0
    assert 0 < d["config"]["shadow_samples_generated_per_frame"] < d["config"]["rays_per_frame"]


def test_mid_size_scene_uses_the_reach_mask(renderer, oracle, tmp_path):
    """32 surfaces: no grid, no register-resident kernel -- the any-size kernel tests, per (wave, light), only the
    surfaces some hit of the wave can reach.  Same counters and image as the oracle, and as RT_FLAG_NO_CULL."""
    sc = R.parse_scene_file(write_scene(tmp_path, _crowded_scene_text(22, 8, seed=3).replace("set 0.3 0.4 0.6 2 3", "set 0.3 0.4 0.6 4 3")))
    assert len(sc.surfaces) == 32
    w, h = 128, 72
    renderer.set_scene(sc)
    ref, cnt = oracle.render_frame(sc, w, h, seed=2, threads=THREADS)
    _, masked, st = renderer.render_rows(w, h, seed=2)
    _, brute, _ = renderer.render_rows(w, h, seed=2, flags=2)
    assert counters_of(st) == cnt
    assert_parity(masked, ref)
    assert CI.metrics(brute, masked)["max_abs_diff"] <= 1


@pytest.mark.parametrize("opaque", [False, True])
def test_mid_size_scene_takes_linear_rows_and_skips_sample_tiles(renderer, oracle, tmp_path, opaque):
    """40 surfaces, spheres and planes only, N = 10: most (wave, light) pairs reach a handful of surfaces, all of them
    clear of the hit point, so the any-size kernel evaluates them as linear rows and decides whole sample tiles at once
    (fewer samples generated than N * N per pair).  Image and counters as the oracle's and as RT_FLAG_NO_CULL's."""
    text = _crowded_scene_text(38, 0, seed=11).replace("set 0.3 0.4 0.6 2 3", "set 0.3 0.4 0.6 10 3")
    if opaque:
        text = text.replace("  30 0.5", "  30 0").replace("  15 0.4", "  15 0")
    sc = R.parse_scene_file(write_scene(tmp_path, text))
    assert len(sc.surfaces) == 40
    w, h = 128, 72
    renderer.set_scene(sc)
    ref, cnt = oracle.render_frame(sc, w, h, seed=4, threads=THREADS)
    _, linear, st = renderer.render_rows(w, h, seed=4)
    _, brute, st_b = renderer.render_rows(w, h, seed=4, flags=2)
    assert counters_of(st) == cnt
    assert_parity(linear, ref)
    assert CI.metrics(brute, linear)["max_abs_diff"] <= 1
    print("classes", st["wave_light"], st["shadow_samples"], st_b["shadow_samples"])
    assert 0 < st["shadow_samples"] < 0.75 * st_b["shadow_samples"]
