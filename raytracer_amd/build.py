"""Builds ``raytracer_amd/librt_amd.so`` (HIP kernels + C ABI) for gfx950 with hipcc.

    python -m raytracer_amd.build [--force]

The float64 ray-tree kernel and the host code that derives scene constants are
compiled with floating-point contraction off (their fused multiply-adds are
explicit); the FP32 shading kernels are compiled with contraction on, except for
the float64 functions of rt_exact.h, which switch it off by pragma (hence
fast-honor-pragmas: plain "fast" ignores the pragma).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "librt_amd.so")
ARCH = "gfx950"

SOURCES = [
    # (file, extra flags)
    ("rt_trace.hip", ["-ffp-contract=off"]),
    ("rt_shade.hip", ["-ffp-contract=fast-honor-pragmas"]),  # FP32 code fuses; rt_exact.h switches it off for its float64 chain
    ("rt_api.hip", ["-ffp-contract=off"]),
    ("rt_deliver.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
HEADERS = [os.path.join(CSRC, "rt_device.h"), os.path.join(CSRC, "rt_exact.h"), os.path.join(HERE, "..", "include", "rt_amd.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the renderer has no CPU fallback")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    objs = []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, __file__] + HEADERS):
            cmd = [cc] + COMMON + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
