#!/usr/bin/env python3
"""Generates tests/golden/ by running the UNMODIFIED reference (build container only).

TEST INFRASTRUCTURE.  Usage:  python -m oracle.make_goldens [--jobs 7] [--only NAME ...]

Everything written is data: images / colour tables / hit tuples / counters that
the reference produced for the scene files under scenes/ (whose numbers equal the
reference's own scene files) with the keyed jitter stream of oracle/jitter.py
injected through the wrappers of oracle/ref_harness.py, plus Mersenne-Twister
seeded renders of the reference's own frame loop for the free-running mode.
No reference source is copied; the reference tree must be present to run this.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")

from oracle import jitter  # noqa: E402


def scene_path(name):
    return os.path.join(SCENES, name + ".txt")


# ------------------------------------------------------------------ workers
_cache = {}


PERTURB = 1e-13   # camera nudge of the "perturbed" renders (every component of the position)


def _scene(name):
    """Parsed scene; ``name#p`` = the same scene with the camera position nudged by PERTURB on every component (the
    reference's own Camera constructor rebuilds the basis)."""
    from oracle import ref_harness as H
    if name not in _cache:
        base = name.split("#")[0]
        sc = H.parse(scene_path(base))
        if name.endswith("#p"):
            cam = sc.camera
            sc.camera = type(cam)(np.asarray(cam.position, dtype=float) + PERTURB, cam.look_at, cam.up_vector,
                                  cam.screen_distance, cam.screen_width)
        _cache[name] = sc
    return _cache[name]


def _work_pixels(args):
    from oracle import ref_harness as H
    name, W, Hh, xs, ys, seed, want_hits = args
    rgb, counters, hits = H.render_pixels(_scene(name), W, Hh, xs, ys, seed, record_hits=want_hits)
    return rgb, counters, hits


def _work_mt(args):
    from oracle import ref_harness as H
    name, W, Hh, mt_seed = args
    return H.render_frame_mt(_scene(name), W, Hh, mt_seed)


def parallel_pixels(pool, name, W, Hh, xs, ys, seed, chunk=256, want_hits=False):
    xs = np.asarray(xs)
    ys = np.asarray(ys)
    jobs = [(name, W, Hh, xs[i:i + chunk], ys[i:i + chunk], seed, want_hits) for i in range(0, len(xs), chunk)]
    rgbs, hits = [], []
    total = None
    for rgb, counters, h in pool.imap(_work_pixels, jobs):
        rgbs.append(rgb)
        if h:
            hits.extend(h)
        if total is None:
            total = counters
        else:
            for k, v in counters.items():
                if k == "depth_hist":
                    for d, c in v.items():
                        total[k][d] = total[k].get(d, 0) + c
                else:
                    total[k] += v
    return np.concatenate(rgbs), total, hits


def sparse_pixels(W, Hh, n, salt):
    """Deterministic stratified-ish pixel sample: n distinct pixels chosen by hash."""
    chosen, seen = [], set()
    i = 0
    while len(chosen) < n:
        h = jitter.mix32(i * 0x9E3779B9 + salt)
        p = h % (W * Hh)
        i += 1
        if p in seen:
            continue
        seen.add(p)
        chosen.append(p)
    chosen = np.array(sorted(chosen))
    return chosen % W, chosen // W


# ------------------------------------------------------------------ items
def item_frames(pool, only):
    from PIL import Image
    os.makedirs(os.path.join(GOLD, "frames"), exist_ok=True)
    specs = [("pool", 500, 500), ("cubes", 480, 270), ("transparency_v7", 480, 270), ("transparency_original", 256, 256)]
    meta = {}
    for name, W, Hh in specs:
        key = f"{name}_{W}x{Hh}_seed0"
        if only and key not in only and "frames" not in only:
            continue
        t0 = time.time()
        xs = np.tile(np.arange(W), Hh)
        ys = np.repeat(np.arange(Hh), W)
        rgb, counters, _ = parallel_pixels(pool, name, W, Hh, xs, ys, 0)
        img = rgb.reshape(Hh, W, 3)
        Image.fromarray(np.uint8(img)).save(os.path.join(GOLD, "frames", key + ".png"))
        # float64 values of every 16th row pin the un-truncated colours
        rows = np.arange(0, Hh, 16)
        np.savez_compressed(os.path.join(GOLD, "frames", key + "_rows.npz"), rows=rows, rgb=img[rows])
        meta[key] = {"scene": name, "width": W, "height": Hh, "seed": 0, "counters": counters,
                     "sum_f64": float(img.sum()), "seconds": round(time.time() - t0, 1)}
        print("frame", key, meta[key], flush=True)
        with open(os.path.join(GOLD, "frames", key + ".json"), "w") as f:
            json.dump(meta[key], f, indent=1)


def item_sparse(pool, only):
    os.makedirs(os.path.join(GOLD, "sparse"), exist_ok=True)
    specs = [("cubes", 1920, 1080, 2048), ("transparency_v7", 1920, 1080, 2048), ("pool", 3840, 2160, 1024),
             ("pool_n16", 3840, 2160, 768), ("synthetic_10k_1k", 3840, 2160, 384)]
    for name, W, Hh, n in specs:
        key = f"{name}_{W}x{Hh}"
        if only and key not in only and "sparse" not in only:
            continue
        t0 = time.time()
        xs, ys = sparse_pixels(W, Hh, n, salt=len(name) * 7919 + W)
        rgb, counters, _ = parallel_pixels(pool, name, W, Hh, xs, ys, 0, chunk=16 if "synthetic" in name else 64)
        np.savez_compressed(os.path.join(GOLD, "sparse", key + ".npz"), x=xs.astype(np.int32), y=ys.astype(np.int32),
                            rgb=rgb, width=W, height=Hh, seed=0,
                            counters=json.dumps(counters))
        print("sparse", key, counters, round(time.time() - t0, 1), "s", flush=True)


def item_perturbed(pool, only):
    """The oracle against itself: the frames and sparse pixel sets above once more, identical jitter, camera position
    nudged by 1e-13.  Where the two renders differ the reference's own image hinges on float64 rounding noise (shadow
    rays that start on the hit sphere: sign of |P-C|^2 - r^2), so this is the ceiling of any parity figure and the
    mask of pixels on which a comparison is meaningful.  Stored as uint8 (np.uint8 truncation, as the goldens)."""
    if only and "perturbed" not in only:
        return
    os.makedirs(os.path.join(GOLD, "perturbed"), exist_ok=True)
    for name, W, Hh in [("pool", 500, 500), ("cubes", 480, 270), ("transparency_v7", 480, 270), ("transparency_original", 256, 256)]:
        t0 = time.time()
        xs = np.tile(np.arange(W), Hh)
        ys = np.repeat(np.arange(Hh), W)
        rgb, _, _ = parallel_pixels(pool, name + "#p", W, Hh, xs, ys, 0)
        np.savez_compressed(os.path.join(GOLD, "perturbed", f"{name}_{W}x{Hh}_seed0.npz"), u8=np.uint8(rgb.reshape(Hh, W, 3)),
                            nudge=PERTURB)
        print("perturbed frame", name, round(time.time() - t0, 1), "s", flush=True)
    for name, W, Hh, n in [("cubes", 1920, 1080, 2048), ("transparency_v7", 1920, 1080, 2048), ("pool", 3840, 2160, 1024),
                           ("pool_n16", 3840, 2160, 768), ("synthetic_10k_1k", 3840, 2160, 384)]:
        t0 = time.time()
        xs, ys = sparse_pixels(W, Hh, n, salt=len(name) * 7919 + W)
        rgb, _, _ = parallel_pixels(pool, name + "#p", W, Hh, xs, ys, 0, chunk=16 if "synthetic" in name else 64)
        np.savez_compressed(os.path.join(GOLD, "perturbed", f"{name}_{W}x{Hh}.npz"), x=xs.astype(np.int32), y=ys.astype(np.int32),
                            u8=np.uint8(rgb), nudge=PERTURB)
        print("perturbed sparse", name, round(time.time() - t0, 1), "s", flush=True)


def item_chain(pool, only):
    os.makedirs(os.path.join(GOLD, "chain"), exist_ok=True)
    specs = [("pool", 48, 48), ("cubes", 48, 27), ("transparency_v7", 48, 27), ("transparency_original", 32, 32)]
    for name, W, Hh in specs:
        key = f"{name}_{W}x{Hh}_hits"
        if only and key not in only and "chain" not in only:
            continue
        xs = np.tile(np.arange(W), Hh)
        ys = np.repeat(np.arange(Hh), W)
        rgb, counters, hits = parallel_pixels(pool, name, W, Hh, xs, ys, 0, want_hits=True)
        np.savez_compressed(
            os.path.join(GOLD, "chain", key + ".npz"),
            pixel=np.array([h[0] for h in hits], dtype=np.uint32), node=np.array([h[1] for h in hits], dtype=np.uint32),
            surf=np.array([h[2] for h in hits], dtype=np.int32), t=np.array([h[3] for h in hits]),
            P=np.array([h[4] for h in hits]), n=np.array([h[5] for h in hits]), rgb=rgb, width=W, height=Hh,
            counters=json.dumps(counters))
        print("chain", key, len(hits), counters, flush=True)


def item_counters(pool, only):
    """Known-answer ray-tree counters (independent of the jitter stream) + MT checksums (tripwire)."""
    if only and "counters" not in only:
        return
    out = {}
    for name, W, Hh in [("pool", 32, 32), ("cubes", 64, 36), ("transparency_v7", 64, 36), ("transparency_original", 48, 48)]:
        xs = np.tile(np.arange(W), Hh)
        ys = np.repeat(np.arange(Hh), W)
        rgb, counters, _ = parallel_pixels(pool, name, W, Hh, xs, ys, 0)
        img = pool.apply(_work_mt, ((name, W, Hh, 1234),))
        out[f"{name}_{W}x{Hh}"] = {"scene": name, "width": W, "height": Hh, "counters": counters,
                                   "keyed_seed0_sum_f64": float(rgb.sum()), "mt1234_sum_f64": float(img.sum())}
        print("counters", name, out[f"{name}_{W}x{Hh}"], flush=True)
    with open(os.path.join(GOLD, "counters.json"), "w") as f:
        json.dump(out, f, indent=1)


def item_mt(pool, only):
    """Free-running mode: the reference's own render() under np.random.seed (PSNR >= 40 dB target)."""
    from PIL import Image
    if only and "mt" not in only:
        return
    os.makedirs(os.path.join(GOLD, "frames_mt"), exist_ok=True)
    specs = [("pool", 160, 160, 1234), ("cubes", 160, 90, 1234), ("transparency_v7", 160, 90, 1234)]
    for (name, W, Hh, sd), img in zip(specs, pool.map(_work_mt, specs)):
        Image.fromarray(np.uint8(img)).save(os.path.join(GOLD, "frames_mt", f"{name}_{W}x{Hh}_mt{sd}.png"))
        print("mt", name, float(img.sum()), flush=True)


def item_kat(pool, only):
    """Per-primitive (o,d)->(t,n) tables from the reference's own classes, dot/norm vectors, jitter table."""
    if only and "kat" not in only:
        return
    from oracle import ref_harness as H
    m = H.load()
    sph_mod, pln_mod, box_mod = m["sphere"], m["plane"], m["cube"]
    os.makedirs(os.path.join(GOLD, "kat"), exist_ok=True)
    rng = np.random.default_rng(20260101)
    n = 600
    kinds, params, O, D = [], [], [], []
    for k in range(n):
        kind = k % 3
        if kind == 0:
            p = np.array([*rng.uniform(-3, 3, 3), rng.uniform(0.2, 2.0)])
        elif kind == 1:
            nrm = rng.standard_normal(3)
            if k % 2:
                nrm = np.eye(3)[k % 3] * (1 if k % 4 else -1)
            p = np.array([*nrm, rng.uniform(-2, 2)])
        else:
            p = np.array([*rng.uniform(-3, 3, 3), rng.uniform(0.3, 2.5)])
        mode = (k // 3) % 4
        o = rng.uniform(-6, 6, 3)
        d = rng.standard_normal(3)
        if mode == 1:  # aim at the primitive
            target = p[:3] if kind != 1 else o - p[:3] / np.linalg.norm(p[:3])
            d = (target + rng.standard_normal(3) * 0.3 * abs(p[3])) - o
        elif mode == 2 and kind != 1:  # start inside
            o = p[:3] + rng.uniform(-0.4, 0.4, 3) * p[3] * (1.0 if kind == 0 else 0.9)
        elif mode == 3 and kind == 2:  # axis-parallel ray
            d[k % 3] = 0.0
        d = d / np.linalg.norm(d)
        kinds.append(kind); params.append(p); O.append(o); D.append(d)
    # epsilon cases: origin exactly on the surface (as shadow rays are), both directions
    for k in range(120):
        kind = k % 3
        if kind == 0:
            c = rng.uniform(-2, 2, 3); r = [1.0, 0.7, 0.45][k % 3 if k % 2 else 0]
            v = rng.standard_normal(3); v /= np.linalg.norm(v)
            o0 = rng.uniform(-6, 6, 3); dd = (c + v * r * 0.5) - o0; dd /= np.linalg.norm(dd)
            sp = sph_mod.Sphere(list(c), r, 1)
            t, _ = sp.intersect(o0, dd)
            o = o0 + t * dd
            p = np.array([*c, r])
        elif kind == 1:
            p = np.array([0.0, 1.0, 0.0, -1.0]); o0 = rng.uniform(-6, 6, 3); o0[1] = abs(o0[1]) + 1
            dd = rng.standard_normal(3); dd[1] = -abs(dd[1]); dd /= np.linalg.norm(dd)
            pl = pln_mod.InfinitePlane([0, 1, 0], -1.0, 1)
            t, _ = pl.intersect(o0, dd); o = o0 + t * dd
        else:
            c = rng.uniform(-2, 2, 3); e = 1.5
            o0 = rng.uniform(-6, 6, 3); dd = (c + rng.uniform(-0.3, 0.3, 3)) - o0; dd /= np.linalg.norm(dd)
            bx = box_mod.Cube(list(c), e, 1)
            t, _ = bx.intersect(o0, dd); o = o0 + t * dd
            p = np.array([*c, e])
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        kinds.append(kind); params.append(p); O.append(o); D.append(d)
    kinds = np.array(kinds, dtype=np.int32); params = np.array(params); O = np.array(O); D = np.array(D)
    hit = np.zeros(len(kinds), dtype=np.int32); T = np.zeros(len(kinds)); Nn = np.zeros((len(kinds), 3)); TB = np.zeros(len(kinds))
    for i, (kind, p, o, d) in enumerate(zip(kinds, params, O, D)):
        if kind == 0:
            s = sph_mod.Sphere(list(p[:3]), float(p[3]), 1)
        elif kind == 1:
            s = pln_mod.InfinitePlane(list(p[:3]), float(p[3]), 1)
        else:
            s = box_mod.Cube(list(p[:3]), float(p[3]), 1)
        t, nn = s.intersect(o, d)
        if t is not None:
            hit[i] = 1; T[i] = t; Nn[i] = nn
        TB[i] = s.intersect_batch(o[None, :], d[None, :])[0]
    # kind 1 tables use the RAW normal (the oracle / library normalise it themselves)
    np.savez_compressed(os.path.join(GOLD, "kat", "primitives.npz"), kind=kinds, param=params, o=O, d=D, hit=hit, t=T, n=Nn,
                        t_batch=TB)
    A = rng.standard_normal((512, 3)) * 10.0 ** rng.uniform(-3, 3, (512, 1))
    B = rng.standard_normal((512, 3))
    dots = np.array([np.dot(a, b) for a, b in zip(A, B)])
    norms = np.array([np.linalg.norm(a) for a in A])
    unit = np.array([m["rt"].normalize(a) for a in A])
    sums = np.sum(A * B, axis=1)
    cross = np.array([np.cross(a, b) for a, b in zip(A, B)])
    np.savez_compressed(os.path.join(GOLD, "kat", "numerics.npz"), a=A, b=B, dot=dots, norm=norms, unit=unit, sum=sums, cross=cross)
    table = []
    for seed, pixel, node, light, n, s in [(0, 0, 1, 0, 5, 0), (0, 1, 1, 0, 5, 1), (0, 123456, 1, 4, 5, 24), (7, 8294399, 2, 1, 16, 255),
                                           (0xFFFFFFFF, 42, 1023, 0, 2, 3), (1, 2, 3, 4, 3, 5), (0, 250000, 5, 2, 5, 17), (99, 77, 6, 0, 1, 0)]:
        u, v = jitter.sample_bits(jitter.stream_key(seed, pixel, node, light), n, s)
        table.append({"seed": seed, "pixel": pixel, "node": node, "light": light, "n": n, "s": s, "u": u, "v": v})
    with open(os.path.join(GOLD, "kat", "jitter_kat.json"), "w") as f:
        json.dump(table, f, indent=1)
    print("kat done", len(kinds), flush=True)


ITEMS = {"kat": item_kat, "counters": item_counters, "chain": item_chain, "mt": item_mt, "sparse": item_sparse, "frames": item_frames,
         "perturbed": item_perturbed}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=7)
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    with mp.Pool(a.jobs) as pool:
        for name, fn in ITEMS.items():
            t0 = time.time()
            fn(pool, a.only)
            print(f"[{name}] {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
