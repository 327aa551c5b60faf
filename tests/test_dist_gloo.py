"""Index-range sharding + all-gather, world_size 2 on the gloo backend (CPU).

The per-rank compute is injected (the oracle) because no GPU exists here; what
is under test is the exchange, the ownership ranges and the host concatenation.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, k, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature, gather_to_rank0, shard_range
    import pct_oracle as oracle

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def compute(full, lo, hi, kk, eps):
        r = oracle.pipeline_batched(full, kk, rows=np.arange(lo, hi), workers=1)
        return r["K"], r["H"]

    lo, hi = shard_range(n, rank, world)
    local = shapes.torus_random(n, seed=21, lo=lo, hi=hi)
    sc = ShardedCurvature(n, k, rank, world, compute=compute)
    K, H = sc.step(local)
    assert len(K) == hi - lo
    Kall, Hall = gather_to_rank0(K, H, n, rank, world)
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), K=Kall, H=Hall)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [3000, 3001])
def test_two_rank_sharding_matches_single_process(tmp_path, n):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pct_oracle as oracle
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes

    k, port = 20, _free_port()
    mp.spawn(_worker, args=(2, port, n, k, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = oracle.pipeline_batched(shapes.torus_random(n, seed=21), k, workers=1)
    assert np.array_equal(got["K"], ref["K"]) and np.array_equal(got["H"], ref["H"])   # independent of G


def test_shard_ranges_partition_the_cloud(built):
    from point_cloud_toolbox_amd.dist import shard_range
    for n in (1, 7, 1000, 1_000_003):
        for g in (1, 2, 4, 8):
            r = [shard_range(n, i, g) for i in range(g)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(g - 1))


def test_sharded_path_requires_a_device_or_checker(built):
    from point_cloud_toolbox_amd.dist import ShardedCurvature
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ShardedCurvature(100, 5, 0, 1)


def _pipeline_worker(rank, world, port, n, steps, out_dir):
    """bench.py's multi-GPU step pattern on CPU tensors: the exchange of cloud i+1 is started before cloud i is
    consumed, two gather buffers alternate."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature, shard_range

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    sc = ShardedCurvature(n, 5, rank, world, compute=lambda *a: None)
    bufs = [torch.empty((n, 3), dtype=torch.float32) for _ in range(2)]
    clouds = [torch.from_numpy(shapes.torus_scan_order(n, world, rank, seed=100 + i)) for i in range(steps + 1)]
    seen = []
    ticket = sc.begin_exchange(clouds[0], bufs[0])
    for i in range(steps):
        cur = sc.end_exchange(ticket, bufs[i % 2])
        ticket = sc.begin_exchange(clouds[i + 1], bufs[(i + 1) % 2])      # runs while `cur` is being consumed
        seen.append(cur.clone().numpy())
    sc.end_exchange(ticket, bufs[steps % 2])
    if rank == 0:
        np.savez(os.path.join(out_dir, "pipe.npz"), *seen)
    dist.barrier()
    dist.destroy_process_group()


def test_double_buffered_exchange_delivers_every_cloud_intact(tmp_path):
    import torch.multiprocessing as mp
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes

    world, n, steps, port = 2, 4000, 5, _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "pipe.npz"))
    for i in range(steps):
        want = np.concatenate([shapes.torus_scan_order(n, world, r, seed=100 + i) for r in range(world)])
        assert np.array_equal(got[f"arr_{i}"], want), f"cloud {i}"
