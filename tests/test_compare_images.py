"""The image checker: the reference tool's thresholds (MSE < 0.001 and PSNR > 40 dB) plus the 1-LSB share."""
import numpy as np
import pytest
from PIL import Image

from raytracer_amd import compare_images as CI


def test_identical_images():
    a = np.random.default_rng(0).integers(0, 256, (32, 48, 3), dtype=np.uint8)
    m = CI.metrics(a, a)
    assert m["mse"] == 0 and m["psnr"] == float("inf") and m["within_1lsb"] == 1.0
    assert m["pass_seeded"] and m["pass_stochastic"]


def test_one_lsb_everywhere_is_within_tolerance():
    a = np.full((16, 16, 3), 100, np.uint8)
    b = a + 1
    m = CI.metrics(a, b)
    assert m["within_1lsb"] == 1.0 and m["max_abs_diff"] == 1 and m["pass_seeded"]
    assert abs(m["mse"] - (1 / 255) ** 2) < 1e-12


def test_share_counts_pixels_not_channels():
    a = np.zeros((10, 10, 3), np.uint8)
    b = a.copy()
    b[0, 0, 1] = 2          # one channel of one pixel beyond 1 LSB
    b[1, 1] = 1             # within
    m = CI.metrics(a, b)
    assert m["pixels_beyond_1lsb"] == 1 and m["within_1lsb"] == 0.99 and not m["pass_seeded"]


def test_psnr_formula_and_threshold():
    a = np.zeros((8, 8, 3), np.uint8)
    b = np.full((8, 8, 3), 3, np.uint8)   # mse = (3/255)^2 -> psnr = 38.6 dB
    m = CI.metrics(a, b)
    assert abs(m["psnr"] - 20 * np.log10(255 / 3)) < 1e-9 and not m["pass_stochastic"]
    b = np.full((8, 8, 3), 2, np.uint8)   # 42.1 dB
    assert CI.metrics(a, b)["pass_stochastic"]


def test_shape_mismatch_raises():
    with pytest.raises(ValueError):
        CI.metrics(np.zeros((4, 4, 3), np.uint8), np.zeros((4, 5, 3), np.uint8))


def test_cli_exit_codes(tmp_path):
    a = np.zeros((8, 8, 3), np.uint8)
    pa, pb, pc = (str(tmp_path / n) for n in ("a.png", "b.png", "c.png"))
    Image.fromarray(a).save(pa); Image.fromarray(a + 1).save(pb); Image.fromarray(a + 40).save(pc)
    for args, code in (([pa, pb, "--quiet"], 0), ([pa, pc, "--quiet"], 1), ([pa, str(tmp_path / "missing.png"), "--quiet"], 2),
                       ([pa, pb, "--quiet", "--lsb"], 0), ([pa, pc, "--quiet", "--lsb", "--save-diff", str(tmp_path / "d.png")], 1)):
        with pytest.raises(SystemExit) as e:
            CI.main(args)
        assert e.value.code == code, args


def test_tile_psnr_localises_a_defect():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, size=(100, 150, 3), dtype=np.uint8)
    b = a.copy()
    b[70:90, 130:150] = 255 - b[70:90, 130:150]          # one damaged corner block
    grid, worst = CI.tile_psnr(a, b, tile=64)
    assert grid.shape == (2, 3) and worst == (1, 2)
    assert np.isinf(grid[0, 0]) and grid[1, 2] < 15.0
    assert CI.metrics(a, b)["psnr"] > grid[1, 2]          # the whole-frame figure averages the defect away
    with pytest.raises(ValueError):
        CI.tile_psnr(a, b[:50])
