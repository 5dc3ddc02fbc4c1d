#!/usr/bin/env python3
"""Image checker: MSE / PSNR with the reference tool's thresholds, plus the 1-LSB metric.

Counterpart of /root/reference/optimizations/compare_images.py (thresholds :91-92,
exit codes :169/:174): an image pair passes the *stochastic* check when
``MSE < 0.001`` and ``PSNR > 40 dB`` on [0,1]-normalised values.  The reference
tool has no per-pixel metric, so the *seeded* check is added here: the share of
pixels whose three 8-bit channels all differ by at most 1 LSB (target >= 99.9 %).

    python -m raytracer_amd.compare_images ref.png test.png [--save-diff d.png] [--lsb] [--tiles 64]

Exit codes: 0 match, 1 differ, 2 error.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

MSE_THRESHOLD = 0.001
PSNR_THRESHOLD = 40.0
LSB_SHARE_THRESHOLD = 0.999


def metrics(ref_u8, test_u8):
    """All metrics of two uint8 images (any shape ending in the channel axis)."""
    a = np.asarray(ref_u8)
    b = np.asarray(test_u8)
    if a.shape != b.shape:
        raise ValueError(f"Image shapes don't match: {a.shape} vs {b.shape}")
    fa = a.astype(np.float64) / 255.0
    fb = b.astype(np.float64) / 255.0
    mse = float(np.mean((fa - fb) ** 2))
    psnr = float("inf") if mse == 0 else float(10.0 * np.log10(1.0 / mse))
    diff = np.abs(a.astype(np.int16) - b.astype(np.int16))
    per_pixel = diff.reshape(-1, a.shape[-1]).max(axis=1) if a.ndim >= 2 else diff
    within = float(np.mean(per_pixel <= 1))
    return {
        "mse": mse, "psnr": psnr,
        "within_1lsb": within, "pixels_beyond_1lsb": int(np.sum(per_pixel > 1)), "max_abs_diff": int(diff.max(initial=0)),
        "pass_stochastic": bool(mse < MSE_THRESHOLD and psnr > PSNR_THRESHOLD),
        "pass_seeded": bool(within >= LSB_SHARE_THRESHOLD),
    }


def tile_psnr(ref_u8, test_u8, tile=64):
    """PSNR of every ``tile`` x ``tile`` block (edge blocks are smaller): a localised defect that a whole-frame
    PSNR averages away shows up as one bad tile.  Returns (grid of PSNR values, (row, col) of the worst tile)."""
    a = np.asarray(ref_u8, dtype=np.float64) / 255.0
    b = np.asarray(test_u8, dtype=np.float64) / 255.0
    if a.shape != b.shape:
        raise ValueError(f"Image shapes don't match: {a.shape} vs {b.shape}")
    h, w = a.shape[:2]
    ny, nx = (h + tile - 1) // tile, (w + tile - 1) // tile
    grid = np.empty((ny, nx))
    for ty in range(ny):
        for tx in range(nx):
            d = a[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile] - b[ty * tile:(ty + 1) * tile, tx * tile:(tx + 1) * tile]
            mse = float(np.mean(d * d))
            grid[ty, tx] = np.inf if mse == 0 else 10.0 * np.log10(1.0 / mse)
    worst = np.unravel_index(int(np.argmin(grid)), grid.shape)
    return grid, (int(worst[0]), int(worst[1]))


def difference_map(ref_u8, test_u8, scale=10.0):
    d = np.abs(np.asarray(ref_u8, dtype=np.float64) - np.asarray(test_u8, dtype=np.float64)) / 255.0
    return np.uint8(np.clip(d * scale, 0.0, 1.0) * 255)


def compare_images(ref_path, test_path, save_diff=None, verbose=True, seeded=False, tile=0):
    from PIL import Image

    ref = np.array(Image.open(ref_path))
    test = np.array(Image.open(test_path))
    m = metrics(ref, test)
    m["pass"] = m["pass_seeded"] if seeded else m["pass_stochastic"]
    if verbose:
        print(f"Reference: {ref_path}\nTest:      {test_path}")
        print(f"MSE:       {m['mse']:.6f}  (threshold: < {MSE_THRESHOLD})")
        print(f"PSNR:      {m['psnr']:.2f} dB  (threshold: > {PSNR_THRESHOLD} dB)")
        print(f"<=1 LSB:   {100 * m['within_1lsb']:.4f} % of pixels  ({m['pixels_beyond_1lsb']} beyond, max diff {m['max_abs_diff']}; "
              f"seeded threshold: >= {100 * LSB_SHARE_THRESHOLD} %)")
        if tile:
            grid, (ty, tx) = tile_psnr(ref, test, tile)
            m["worst_tile_psnr"] = float(grid[ty, tx])
            print(f"Worst {tile}x{tile} tile: {grid[ty, tx]:.2f} dB at rows {ty * tile}.., columns {tx * tile}..")
        print("PASS" if m["pass"] else "FAIL")
    if save_diff:
        Image.fromarray(difference_map(ref, test)).save(save_diff)
    return m


def main(argv=None):
    ap = argparse.ArgumentParser(description="Compare two ray tracer output images")
    ap.add_argument("reference")
    ap.add_argument("test")
    ap.add_argument("--save-diff", metavar="PATH")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--lsb", action="store_true", help="judge by the share of pixels within 1 LSB (seeded renders)")
    ap.add_argument("--tiles", type=int, default=0, metavar="SIZE", help="also report the worst SIZE x SIZE tile's PSNR")
    a = ap.parse_args(argv)
    try:
        r = compare_images(a.reference, a.test, a.save_diff, not a.quiet, a.lsb, a.tiles)
    except Exception as e:  # noqa: BLE001
        if not a.quiet:
            print(f"Error: {e}", file=sys.stderr)
        sys.exit(2)
    sys.exit(0 if r["pass"] else 1)


if __name__ == "__main__":
    main()
