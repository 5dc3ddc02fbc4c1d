"""Pins oracle/rt_oracle.c against vectors produced by the unmodified reference (tests/golden/).

The golden files were written by oracle/make_goldens.py in the build container
(reference + keyed jitter stream injected through wrappers).  The restatement is
expected to reproduce them exactly: float64 colours, every (t, P, n) of the ray
tree, ray counters.  A tiny allowance (<= 1e-9 on [0,255] colours, <= 0.01 % of
uint8 pixels off by one) absorbs a libm whose pow() rounds differently.
"""
import ctypes
import json
import math

import numpy as np
import pytest

from helpers import THREADS, golden_json, golden_npz, golden_png, grid, scene

FRAMES = [("pool", 500, 500), ("cubes", 480, 270), ("transparency_v7", 480, 270), ("transparency_original", 256, 256)]


@pytest.mark.parametrize("name,w,h", FRAMES)
def test_full_frames_match_reference(oracle, name, w, h):
    key = f"{name}_{w}x{h}_seed0"
    ref_u8 = golden_png(f"frames/{key}.png")
    meta = golden_json(f"frames/{key}.json")
    img, counters = oracle.render_frame(scene(name), w, h, seed=0, threads=THREADS)
    got = np.uint8(img)
    diff = np.abs(got.astype(np.int16) - ref_u8.astype(np.int16))
    assert diff.max() <= 1
    assert np.mean(diff.reshape(-1, 3).max(axis=1) > 0) <= 1e-4
    for k in ("primary", "secondary", "depth_cut", "shaded", "shadow"):
        assert counters[k] == meta["counters"][k], k
    assert abs(float(img.sum()) - meta["sum_f64"]) <= 1e-6 * max(1.0, abs(meta["sum_f64"]))
    rows = golden_npz(f"frames/{key}_rows.npz")
    np.testing.assert_allclose(img[rows["rows"]], rows["rgb"], rtol=0, atol=1e-9)


SPARSE = ["cubes_1920x1080", "transparency_v7_1920x1080", "pool_3840x2160", "pool_n16_3840x2160", "synthetic_10k_1k_3840x2160"]


@pytest.mark.parametrize("key", SPARSE)
def test_sparse_pixels_of_the_benchmark_configs(oracle, key):
    z = golden_npz(f"sparse/{key}.npz")
    name = key.rsplit("_", 1)[0]
    w, h = int(z["width"]), int(z["height"])
    rgb, counters, _ = oracle.render_pixels(scene(name), w, h, z["x"], z["y"], seed=int(z["seed"]), threads=THREADS)
    np.testing.assert_allclose(rgb, z["rgb"], rtol=0, atol=1e-9)
    want = json.loads(str(z["counters"]))
    for k in ("primary", "secondary", "depth_cut", "shaded", "shadow"):
        assert counters[k] == want[k], k


CHAINS = [("pool", 48, 48), ("cubes", 48, 27), ("transparency_v7", 48, 27), ("transparency_original", 32, 32)]


@pytest.mark.parametrize("name,w,h", CHAINS)
def test_every_nearest_hit_is_bit_identical(oracle, name, w, h):
    z = golden_npz(f"chain/{name}_{w}x{h}_hits.npz")
    xs, ys = grid(w, h)
    rgb, _, hits = oracle.render_pixels(scene(name), w, h, xs, ys, seed=0, record_hits=True)
    assert len(hits) == len(z["pixel"])
    got = {(h_[0], h_[1]): h_ for h_ in hits}
    for i in range(len(z["pixel"])):
        g = got[(int(z["pixel"][i]), int(z["node"][i]))]
        assert g[2] == int(z["surf"][i])
        assert g[3] == float(z["t"][i])                      # t: all 64 bits
        assert g[4] == tuple(float(v) for v in z["P"][i])    # hit point
        assert g[5] == tuple(float(v) for v in z["n"][i])    # normal
    np.testing.assert_array_equal(rgb, z["rgb"])             # colours: all 64 bits


def test_known_answer_counters(oracle):
    for key, row in golden_json("counters.json").items():
        w, h = row["width"], row["height"]
        img, counters = oracle.render_frame(scene(row["scene"]), w, h, seed=0, threads=THREADS)
        for k in ("primary", "secondary", "depth_cut", "shaded", "shadow"):
            assert counters[k] == row["counters"][k], (key, k)
        assert abs(float(img.sum()) - row["keyed_seed0_sum_f64"]) < 1e-6


def test_primitive_tables(oracle):
    """(o, d) -> (t, n) of Sphere / InfinitePlane / Cube .intersect and .intersect_batch, incl. on-surface origins."""
    z = golden_npz("kat/primitives.npz")
    kind_to_type = {0: 0, 1: 3, 2: 2}  # plane rows carry the RAW normal -> RT_SURF_PLANE_RAW
    for i in range(len(z["kind"])):
        ty = kind_to_type[int(z["kind"][i])]
        t, n = oracle.intersect_scalar(ty, z["param"][i], z["o"][i], z["d"][i])
        if int(z["hit"][i]):
            assert t == float(z["t"][i]), i
            assert tuple(n) == tuple(z["n"][i]), i
        else:
            assert t is None, i
        tb = oracle.intersect_batch(ty, z["param"][i], z["o"][i], z["d"][i])
        want = float(z["t_batch"][i])
        assert (math.isinf(tb) and math.isinf(want)) or tb == want, i


def test_numeric_conventions_of_the_reference_are_the_ones_restated():
    """np.dot / np.linalg.norm == fma chain; np.sum(axis=1) == rounded products summed left to right."""
    z = golden_npz("kat/numerics.npz")
    libm = ctypes.CDLL("libm.so.6")
    libm.fma.restype = ctypes.c_double
    libm.fma.argtypes = [ctypes.c_double] * 3
    for a, b, d, nrm, s, cr in zip(z["a"], z["b"], z["dot"], z["norm"], z["sum"], z["cross"]):
        a = [float(v) for v in a]; b = [float(v) for v in b]
        assert libm.fma(a[2], b[2], libm.fma(a[1], b[1], a[0] * b[0])) == float(d)
        assert math.sqrt(libm.fma(a[2], a[2], libm.fma(a[1], a[1], a[0] * a[0]))) == float(nrm)
        assert (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2] == float(s)
        assert (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]) == tuple(float(v) for v in cr)


def test_camera_basis_known_answer(oracle):
    # pool camera: look straight down-ish; basis must be orthonormal and right-handed as camera.py:14-16 builds it
    b = oracle.camera_basis(scene("pool"))
    f, r, u = b
    assert abs(np.dot(f, r)) < 1e-15 and abs(np.dot(f, u)) < 1e-15 and abs(np.dot(r, u)) < 1e-15
    np.testing.assert_allclose(np.cross(r, f), u, atol=1e-15)


def test_reference_cpu_baseline_fixture_is_consistent():
    """tests/golden/ref_cpu_baseline.json (oracle/time_reference.py: the unmodified reference timed in the build
    container): every run's ray count is the sum of its counters and its rate follows from rays and seconds."""
    d = golden_json("ref_cpu_baseline.json")
    assert len(d["runs"]) == 5
    for name, run in d["runs"].items():
        for leg, secs in (("one_process", "seconds"), ("processes", "seconds_slowest_worker")):
            r = run[leg]
            c = r["counters"]
            assert r["rays"] == c["primary"] + c["secondary"] + c["shadow"], (name, leg)
            assert abs(r["krays_per_s"] - r["rays"] / r[secs] / 1e3) <= 0.02 * r["krays_per_s"] + 0.01, (name, leg)
            assert c["primary"] > 0 and r[secs] > 1.0
