#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + secondary) on pool.txt at 3840x2160, N=16.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no launcher in the environment the script starts its N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`, one rank per GPU over RCCL) before
anything touches the GPU, relays rank 0's JSON line and exits with the launcher's status; started by a
launcher it checks that WORLD_SIZE equals --gpus and refuses a mismatch.

A step is one full frame of scenes/pool_n16.txt (pool geometry, 16x16 soft-shadow grid, max_recursions 10)
with the scene already resident on the GPU, delivered as the uint8 frame in (pinned) host memory on rank 0 --
what the caller of the reference's render() ends up with (ray_tracer.py:436-439).  The frame is split into
interleaved row stripes over the ranks, the uint8 tiles are gathered to rank 0 over RCCL, and the copy to
the host runs on a stream of its own behind the following frames' kernels; all of it is inside the timed
region.  The K timed frames are queued and alternate over --pipelines renderer handles (streams) per rank,
so consecutive frames overlap at their ends.  The total frame is fixed, so scaling is strong.  Rank 0 prints
ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X FP32 vector peak (256 CU x 4 SIMD x 2.4 GHz x 64 flop/clk/SIMD: one packed FMA per 4 cycles)

WORKLOADS = {
    "pool_n16_4k": ("pool_n16", 3840, 2160),
    "pool_4k": ("pool", 3840, 2160),
    "cubes_1080p": ("cubes", 1920, 1080),
    "v7_1080p": ("transparency_v7", 1920, 1080),
    "synthetic_4k": ("synthetic_10k_1k", 3840, 2160),
    "spheres40_1080p": ("spheres_40", 1920, 1080),
}


def kernel_source_hash():
    """Identifies the kernel sources a counter summary under profiles/ was collected on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(scene_name, width, height, target_seconds=12.0):
    """The float64 CPU restatement (oracle/rt_oracle.c) timed on this host's cores on a
    bounded pixel sample of the SAME workload.  A reported baseline, not a target."""
    import numpy as np
    from oracle import rt_oracle
    import raytracer_amd as R

    rt_oracle.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))
    scene = R.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".txt"))

    def sample(n):
        k = np.arange(n, dtype=np.int64)
        pix = (k * 2654435761) % (width * height)
        return (pix % width).astype(np.int32), (pix // width).astype(np.int32)

    xs, ys = sample(cores * 64)
    t0 = time.perf_counter()
    _, c, _ = rt_oracle.render_pixels(scene, width, height, xs, ys, seed=0, threads=cores)
    dt = max(time.perf_counter() - t0, 1e-6)
    n = int(min(width * height, max(len(xs), len(xs) * target_seconds / dt)))
    xs, ys = sample(n)
    t0 = time.perf_counter()
    _, c, _ = rt_oracle.render_pixels(scene, width, height, xs, ys, seed=0, threads=cores)
    dt = time.perf_counter() - t0
    rays = c["primary"] + c["secondary"] + c["shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{n} pixels of {scene_name} {width}x{height} ({rays} rays, {dt:.1f} s, float64, OpenMP)"}


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as children of this process (which has not touched the
    GPU) and pass rank 0's output through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pool_n16_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--pipelines", type=int, default=0,
                    help="renderer handles (each with a stream of its own) the queued frames alternate over; "
                         "default: 2 on one GPU, 4 when the frame is split over several (measured on one rank's share "
                         "of the headline frame: an eighth takes 0.53 / 0.47 / 0.43 ms with 2 / 3 / 4, the whole frame the same)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-brute-force", action="store_true",
                    help="skip the RT_FLAG_NO_CULL frames after the timed region (profiling runs: keeps per-kernel averages clean)")
    ap.add_argument("--no-device-resident", action="store_true",
                    help="skip the second timed loop that leaves the frames on the device (profiling runs)")
    ap.add_argument("--no-pool-4k", action="store_true",
                    help="skip the extra leg on scenes/pool.txt itself (N = 5) at 3840x2160 after the timed region")
    ap.add_argument("--no-cull", action="store_true",
                    help="brute force: every shadow ray against every surface (RT_FLAG_NO_CULL), for A/B numbers")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); refusing to report a mismatched run",
                  file=sys.stderr)
        sys.exit(2)

    if world > 1:
        # HIP gives a process 4 hardware queues by default; four pipelines and the copy stream want one each
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    import raytracer_amd as R
    from raytracer_amd import runtime
    from raytracer_amd.distributed import DistributedRenderer

    # RT_BENCH_REHEARSE=1: rehearsal of the multi-rank path on a box with fewer GPUs than ranks -- the ranks share the
    # visible GPUs and gather over gloo (RCCL refuses two ranks on one device); the numbers mean nothing, the code path
    # (stripes, gather, assembly, reductions, the JSON line) is the real one
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif world > 1 and torch.cuda.device_count() < world:
        if rank == 0:
            print(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    if args.pipelines <= 0:
        args.pipelines = 2 if world == 1 else 4
    scene_name, W, H = WORKLOADS[args.workload]
    scene = R.parse_scene_file(os.path.join(ROOT, "scenes", scene_name + ".txt"))
    renderer = runtime.Renderer(local_rank)
    renderer.set_scene(scene)
    flags = 2 if args.no_cull else 0
    dr = DistributedRenderer(renderer, W, H, flags=flags, pipelines=args.pipelines, to_host=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):   # progress lines of a rehearsal (a hang then names its place)
        if rehearse:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    def timed(d, first_seed, retry=True):
        """K queued frames between two barriers; returns (seconds, summed stats of this rank)."""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            d.enqueue(seed=first_seed + i)
        overflow, st = 0.0, None
        try:
            st = d.finish()
        except runtime.RtError as e:
            if e.code != -5:
                raise
            overflow = 1.0
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:    # every rank has to take the same branch below: the retry contains collectives
            flag = torch.tensor([overflow], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            overflow = float(flag[0])
        if overflow:
            # a queued frame outgrew the ray / shade queues (they cannot grow while frames are in flight): size them
            # with one synchronous frame per pipeline, which grows them, and time the K frames again -- once
            if not retry:
                if rank == 0:
                    print("bench.py: queued frames still outgrow the queues after resizing them", file=sys.stderr)
                sys.exit(3)
            for k in range(len(d.lanes)):
                d.render(seed=first_seed, pipeline=k)
            return timed(d, first_seed, retry=False)
        return dt, st

    note("renderer ready")
    isolated = None   # kernel times of a frame rendered alone (warm-up frames run one at a time on one stream)
    for i in range(args.warmup):
        _, isolated = dr.render(seed=i)
    dr.prepare()      # every pipeline has rendered once before the clock starts
    if isolated is None:
        _, isolated = dr.render(seed=0)
    note("warm-up done")
    # ---- the timed region: K frames, each delivered to host memory on rank 0 ---------------------------------
    dt, st = timed(dr, 1000)
    note("timed frames done")
    rays = st["total_rays"]
    wave_light = list(st["wave_light"])
    ms_shade, ms_main, flop_shade, ms_trace, ms_resolve = st["ms_shade"], st["ms_shade_main"], st["flop_shade"], st["ms_trace"], st["ms_resolve"]
    samples = st["shadow_samples"]
    delivery = dr.delivery_report()
    # the frame that reached the host last is compared with a synchronous render of the same seed through the plain
    # entry point (rt_render: one stream, the runtime's own copy): every channel of every pixel within 1 LSB
    host_ok = None
    if rank == 0:
        import numpy as np
        hf = dr.host_frame()
        _, want, _ = renderer.render_rows(W, H, seed=1000 + args.steps - 1, want_f32=False, flags=flags)
        host_ok = bool(hf is not None and hf.shape == want.shape and
                       int(np.abs(hf.astype(np.int16) - want.astype(np.int16)).max()) <= 1 and int(want.max()) > 0)

    # ---- the same K frames left on the device (no copy to the host): reported beside the headline ------------
    dt_dev = None
    if not args.no_device_resident:
        dr.deliver = False
        dt_dev, _ = timed(dr, 3000)
        dr.deliver = True

    # the scene file the metric names, as it is (pool.txt: N = 5), at the same resolution and delivered the same way:
    # reported as an extra key beside the headline (whose workload is the N = 16 configuration of the same geometry)
    pool_4k = None
    if world == 1 and args.workload == "pool_n16_4k" and not args.no_pool_4k and not args.no_cull:
        r5 = runtime.Renderer(local_rank)
        r5.set_scene(R.parse_scene_file(os.path.join(ROOT, "scenes", "pool.txt")))
        d5 = DistributedRenderer(r5, W, H, pipelines=args.pipelines, to_host=True)
        for i in range(max(args.warmup, 1)):
            d5.render(seed=i)
        d5.prepare()
        dt5, st5 = timed(d5, 5000)
        gen5 = st5["total_rays"] - st5["shadow_rays"] + st5["shadow_samples"]
        pool_4k = {"workload": "scenes/pool.txt at 3840x2160 (N = 5, max_recursions 10): the scene file the metric names, as it is",
                   "ms_per_step": dt5 / max(args.steps, 1) * 1e3, "value": st5["total_rays"] / dt5 / 1e6, "unit": "Mrays/s",
                   "rays_per_frame": st5["total_rays"] / max(args.steps, 1),
                   "value_generated_samples_only": gen5 / dt5 / 1e6, "steps": args.steps,
                   "delivery": "uint8 frame in pinned host memory, as the headline"}
        del d5
        r5.close()

    # the same frame with RT_FLAG_NO_CULL (every shadow ray against every surface): what the straightforward
    # kernel reaches of the vector peak; outside the timed region, rank 0 of a single-GPU run only
    brute = None
    if world == 1 and not args.no_cull and not args.no_brute_force and args.workload == "pool_n16_4k":
        drb = DistributedRenderer(renderer, W, H, flags=2)
        drb.render(seed=0)
        bms = bfl = 0.0
        for i in range(2):
            _, stb = drb.render(seed=2000 + i)
            bms += stb["ms_shade"]; bfl += stb["flop_shade"]
        brute = {"shade_ms_per_frame": bms / 2, "algorithmic_tflops": (bfl / 1e12) / (bms / 1e3),
                 "algorithmic_over_peak": (bfl / 1e12) / (bms / 1e3) / FP32_VECTOR_PEAK_TFLOPS}

    if world > 1:
        t = torch.tensor([dt, float(rays), float(samples), dt_dev or 0.0, float(st["shadow_rays"])], dtype=torch.float64,
                         device="cpu" if rehearse else "cuda")
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); rays_total = float(tsum[1]); samples_total = float(tsum[2])
        dt_dev = float(tmax[3]) if dt_dev is not None else None
        shadow_total = float(tsum[4])
    else:
        rays_total = float(rays); samples_total = float(samples); shadow_total = float(st["shadow_rays"])

    if rank == 0:
        steps = max(args.steps, 1)
        # ---- roofline of the dominant kernel ------------------------------------------------------------------
        # executed FP32 flop per launch: rocprofv3 per-class VALU counters of this build (tools/profile.sh,
        # profiles/shade_roofline.json); time: the same launch measured live with HIP events on its own stream,
        # running alone (the last warm-up frame) -- in the timed region two pipelines overlap and a launch's
        # event-to-event time includes time shared with the other stream's kernels
        pmc = {}
        tpath = os.path.join(ROOT, "profiles", "shade_roofline.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                pmc = json.load(f).get(args.workload, {})
        iso_ms = isolated["ms_shade_main"] if isolated else 0.0
        timed_ms = ms_main / steps
        executed = pmc.get("executed_fp32_flop_per_launch")
        achieved = (executed / 1e12) / (iso_ms / 1e3) if executed and iso_ms > 0 and world == 1 and not args.no_cull else None
        algorithmic_iso = (isolated["flop_shade"] / 1e12) / (isolated["ms_shade"] / 1e3) if isolated and isolated["ms_shade"] > 0 else None
        desc = f"scenes/{scene_name}.txt at {W}x{H}"
        if args.workload == "pool_n16_4k":
            desc += " (pool geometry, 16x16 soft-shadow grid, max_recursions 10)"
        out = {
            "metric": "Mrays/s (primary+shadow+secondary) at 3840x2160 pool.txt; frame time ms",
            "metric_note": "rays are REFERENCE-EQUIVALENT rays (what the reference casts for this frame, config.rays_definition); "
                           "value_generated_samples_only counts only the shadow samples the kernels really generated",
            "value": rays_total / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc,
                       "rays_per_frame": rays_total / steps,
                       "rays_definition": "the reference's count: primary + traced secondary + N*N shadow rays per (shaded hit, "
                                          "light) (1 for single-ray lights); shadow_samples_generated_per_frame is what the "
                                          "kernels really sampled after deciding whole (wavefront, light) pairs and sample tiles without sampling",
                       "shadow_samples_generated_per_frame": samples_total / steps,
                       "partition": f"row stripes x{world}, uint8 gather to rank 0",
                       "delivery": "uint8 frame in pinned host memory on rank 0, copy on its own stream inside the timed region",
                       "host_frame_checked": host_ok,
                       "host_frame_check": "last delivered frame against a synchronous rt_render of the same seed: every channel within 1 LSB",
                       "delivery_overlap": delivery,
                       "pipelines": args.pipelines,
                       "geometry_chain": "f64", "shadows_and_shading": "f32"},
            "device_resident": None if dt_dev is None else {
                "ms_per_step": dt_dev / steps * 1e3, "value": rays_total / dt_dev / 1e6, "unit": "Mrays/s",
                "note": "the same K frames with the uint8 frame left on rank 0's device (no copy to the host)"},
            "value_generated_samples_only": (rays_total - shadow_total + samples_total) / dt / 1e6,
            "pool_4k": pool_4k,
            "wave_light_classes_rank0": dict(zip(["none", "k1", "k2", "k3", "k4", "generic", "point", "float64_pairs"], wave_light)),
            "kernel_ms_per_frame_rank0": {"trace": ms_trace / steps, "shade": ms_shade / steps, "shade_main_kernel": timed_ms,
                                          "resolve": ms_resolve / steps,
                                          "isolated": None if not isolated else {k: isolated[k] for k in ("ms_trace", "ms_shade", "ms_shade_main", "ms_resolve")}},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": None if achieved is None else achieved / FP32_VECTOR_PEAK_TFLOPS,
                         "traffic": pmc.get("hbm_bytes_per_launch"),
                         "kernel": pmc.get("kernel", "k_shade_fast") + " (Phong + in-register N x N soft shadows), FP32 vector ALU; "
                                   "no dense contraction, so neither HBM nor MFMA binds it",
                         "executed_fp32_flop_per_launch": executed,
                         "launch_ms_isolated_live": iso_ms, "launch_ms_in_timed_region_overlapped": timed_ms,
                         "launch_ms_under_pmc": pmc.get("ms_under_pmc"),
                         "counters": pmc.get("counters"),
                         "pmc_profile": pmc.get("profile"), "pmc_kernel_source_hash": pmc.get("kernel_source_hash"),
                         "pmc_matches_this_build": (pmc.get("kernel_source_hash") == kernel_source_hash()) if pmc else None,
                         "valu_issue": pmc.get("valu_issue"),
                         "algorithmic_over_peak": None if algorithmic_iso is None else {
                             "tflops": algorithmic_iso, "ratio": algorithmic_iso / FP32_VECTOR_PEAK_TFLOPS,
                             "note": "reference-equivalent flop (every N*N shadow ray against every surface, per-test costs fixed "
                                     "by convention) / isolated shading time: a speed-up over brute force, NOT a roofline fraction"},
                         "brute_force": brute,
                         "note": "achieved = FP32 flop the kernel EXECUTED per launch (rocprofv3 SQ_INSTS_VALU_FLOPS_FP32 x 64 lanes: add / mul / "
                                 "transcendental 1, fma 2, packed fma 4; counters collected on this build, see pmc_*) / the launch's duration measured "
                                 "live with HIP events on its own stream, running alone"},
        }
        # the float64 ray-tree kernel: float64 flop executed per launch (profiles/trace_roofline.json, tools/profile.sh pass
        # pmc_f64) / the kernel's duration measured live on its own stream, against the vector float64 rate
        fpath = os.path.join(ROOT, "profiles", "trace_roofline.json")
        if os.path.exists(fpath) and isolated and isolated["ms_trace"] > 0 and world == 1:
            with open(fpath) as f:
                t64 = json.load(f).get(args.workload)
            if t64 and t64.get("executed_fp64_flop_per_launch"):
                a64 = t64["executed_fp64_flop_per_launch"] / 1e12 / (isolated["ms_trace"] / 1e3)
                out["roofline_ray_tree"] = {"bound": "valu f64 (latency-bound in practice)", "achieved": a64, "peak": t64["peak_tflops"], "unit": "TFLOP/s",
                                            "frac": a64 / t64["peak_tflops"], "kernel": t64["kernel"], "launch_ms_isolated_live": isolated["ms_trace"],
                                            "executed_fp64_flop_per_launch": t64["executed_fp64_flop_per_launch"],
                                            "fp64_share_of_valu_insts": t64.get("fp64_share_of_valu_insts"),
                                            "pmc_matches_this_build": t64.get("kernel_source_hash") == kernel_source_hash()}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene_name, W, H)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
