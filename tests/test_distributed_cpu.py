"""Row-stripe partition + gather, rehearsed on CPU with gloo (world_size 2 and 3).

Each rank 'renders' its stripes with the CPU oracle (the GPU path is covered by the
gpu tests) into the tile of raytracer_amd.distributed.TileGatherer -- the object the
GPU path uses on RCCL -- which gathers the uint8 tiles to rank 0, and rank 0's assembled frame must equal the
single-process frame bit for bit (the keyed jitter stream makes pixels independent
of the partition).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import scene
from raytracer_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, stripe_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import rt_oracle
    sc = scene("transparency_v7")
    # the production gather / assembly object, on CPU tensors; the oracle stands in for the HIP renderer
    tg = D.TileGatherer(w, h, torch.device("cpu"), stripe_rows=stripe_rows)
    assert tg.my_rows == D.rank_rows(h, world, rank, stripe_rows)
    for seed in (4, 5):   # two frames through the same buffers
        img, _ = rt_oracle.render_rows(sc, w, h, tg.my_rows, seed=seed, threads=1)
        tg.tile[: len(tg.my_rows)] = torch.from_numpy(np.uint8(img))
        frame = tg.gather()
    assert (frame is not None) == (rank == 0)
    if rank == 0:
        assert np.array_equal(frame.numpy(), D.assemble([g.numpy() for g in tg.gathered], h, w, world, stripe_rows))
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,stripe_rows", [(2, 4), (3, 5)])
def test_gather_reassembles_the_single_process_frame(tmp_path, oracle, world, stripe_rows):
    w, h = 40, 27   # 27 rows: uneven stripes, ranks hold different row counts
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, stripe_rows, out), nprocs=world, join=True)
    full, _ = oracle.render_frame(scene("transparency_v7"), w, h, seed=5, threads=1)
    np.testing.assert_array_equal(np.load(out), np.uint8(full))


def test_stripe_helpers():
    assert D.stripe_rows_for(2160, 8) == 16
    assert D.stripe_rows_for(16, 8) == 1
    for world in (1, 2, 3, 8):
        rows = sorted(sum((D.rank_rows(100, world, r, 7) for r in range(world)), []))
        assert rows == list(range(100))
    assert D.max_rows_per_rank(27, 2, 4) == 15


def test_bench_refuses_a_world_that_is_not_gpus():
    """Started by a launcher whose WORLD_SIZE differs from --gpus, bench.py exits non-zero before touching a GPU
    (a run reported as N GPUs must have used N ranks)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "refusing" in out.stderr and "{" not in out.stdout
