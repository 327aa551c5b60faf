"""Index-range sharding: ownership, exchange pattern, concatenation and the rendezvous -- on the CPU.

The product's collective is RCCL behind the C ABI (pct_comm_*), which needs GPUs.  What runs here, world size 2 on
torch.distributed's gloo backend, is everything around it: ``ShardedCurvature`` drives an INJECTED exchange object (same
four methods as ``RcclExchange``) and an injected handle whose "device buffers" are host arrays, the per-rank compute is
the oracle.  The TCP rendezvous that distributes the RCCL unique id runs as is, with 2, 4 and 8 processes.
"""
import multiprocessing as mp
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


class GlooExchange:
    """Stand-in for RcclExchange on host tensors: one broadcast per rank (shards may differ in size), asynchronous."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def begin(self, send, recv, counts):
        import torch.distributed as dist
        off, works = 0, []
        for r, c in enumerate(int(c) for c in counts):
            part = recv[off:off + c]
            if r == self.rank:
                part.copy_(send[:c])
            works.append(dist.broadcast(part, src=r, async_op=True))
            off += c
        return works, recv

    def end(self, ticket):
        works, recv = ticket
        for w in works:
            w.wait()
        return recv

    def allgather_host(self, local, counts):
        import torch
        recv = torch.empty(int(np.sum(counts)), dtype=torch.float32)
        self.end(self.begin(torch.from_numpy(np.ascontiguousarray(local, np.float32)).reshape(-1), recv, counts))
        return recv.numpy().reshape(-1, 3)

    def barrier(self):
        import torch.distributed as dist
        dist.barrier()


class HostHandle:
    """The handle calls ShardedCurvature makes, on host memory; ``curvature`` records what the HIP path would see."""

    def __init__(self):
        self.seen = []

    def device_alloc(self, nbytes):
        import torch
        return torch.empty(nbytes // 4, dtype=torch.float32)

    def device_free(self, buf):
        pass

    def device_upload(self, buf, host):
        import torch
        buf[:host.size].copy_(torch.from_numpy(host.reshape(-1)))

    def comm_synchronize(self):
        pass

    def synchronize(self):
        pass

    def use_points_device(self, buf, n):
        self.cloud = buf[:3 * n].numpy().reshape(n, 3)

    def set_query_range(self, lo, hi):
        self.range = (lo, hi)

    def curvature(self, k, eps):
        self.seen.append((self.cloud.copy(), self.range, k, eps))

    # ownership by slab: the cut here is by rank of the x coordinate (any rule every rank applies alike will do);
    # "K" of a point is twice its public index, "H" its negative, so that a misplaced record shows
    def set_query_slab(self, part, parts):
        self.range = ("slab", part, parts)
        order = np.argsort(self.cloud[:, 0], kind="stable")
        cuts = [len(order) * p // parts for p in range(parts + 1)]
        self.counts = [cuts[p + 1] - cuts[p] for p in range(parts)]
        self.mine = np.sort(order[cuts[part]:cuts[part + 1]])

    def slab_counts(self, parts):
        return list(self.counts)

    def slab_records(self, buf, capacity_rows):
        import torch
        assert len(self.mine) <= capacity_rows
        rec = np.empty((len(self.mine), 3), np.float32)
        rec[:, 0] = self.mine.astype(np.int32).view(np.float32)
        rec[:, 1], rec[:, 2] = 2.0 * self.mine, -1.0 * self.mine
        buf[:rec.size].copy_(torch.from_numpy(rec.reshape(-1)))
        return len(self.mine)

    def scatter_records(self, records, n_records, lo, hi, K, H):
        import torch
        from point_cloud_toolbox_amd.dist import scatter_records_host
        k, h = scatter_records_host(records[:3 * n_records].numpy(), lo, hi)
        K[:hi - lo].copy_(torch.from_numpy(k))
        H[:hi - lo].copy_(torch.from_numpy(h))

    def device_download(self, buf, host):
        host[...] = buf[:host.size].numpy().reshape(host.shape)


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import pointCloudToolbox  # noqa: F401
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    return dist


def _worker(rank, world, port, n, k, out_dir):
    dist = _init(rank, world, port)
    from point_cloud_toolbox_amd import shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature, shard_range
    import pct_oracle as oracle

    def compute(full, lo, hi, kk, eps):
        r = oracle.pipeline_batched(full, kk, rows=np.arange(lo, hi), workers=1)
        return r["K"], r["H"]

    lo, hi = shard_range(n, rank, world)
    local = shapes.torus_random(n, seed=21, lo=lo, hi=hi)
    sc = ShardedCurvature(n, k, rank, world, exchange=GlooExchange(rank, world), compute=compute)
    K, H = sc.step(local)
    assert len(K) == hi - lo
    np.savez(os.path.join(out_dir, f"out_{rank}.npz"), K=K, H=H)       # no collective on outputs: the host concatenates
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [3000, 3001])
def test_two_rank_sharding_matches_single_process(tmp_path, n):
    import torch.multiprocessing as tmp_mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pct_oracle as oracle
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes

    k, port = 20, _free_port()
    tmp_mp.spawn(_worker, args=(2, port, n, k, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"out_{r}.npz") for r in range(2)]
    K, H = np.concatenate([p["K"] for p in parts]), np.concatenate([p["H"] for p in parts])
    ref = oracle.pipeline_batched(shapes.torus_random(n, seed=21), k, workers=1)
    assert np.array_equal(K, ref["K"]) and np.array_equal(H, ref["H"])   # independent of G


def test_shard_ranges_partition_the_cloud(built):
    from point_cloud_toolbox_amd.dist import shard_range, shard_sizes
    for n in (1, 7, 1000, 1_000_003, 20_022_479):
        for g in (1, 2, 4, 8):
            r = [shard_range(n, i, g) for i in range(g)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(g - 1))
            assert shard_sizes(n, g) == [b - a for a, b in r] and sum(shard_sizes(n, g)) == n


def test_sharded_path_requires_a_device_or_checker(built):
    from point_cloud_toolbox_amd.dist import ShardedCurvature
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ShardedCurvature(100, 5, 0, 1)
    with pytest.raises(RuntimeError, match="needs an exchange"):
        ShardedCurvature(100, 5, 0, 2, compute=lambda *a: None)


def _pipeline_worker(rank, world, port, n, steps, out_dir):
    """bench.py's multi-GPU step pattern through ShardedCurvature's own buffer logic: the exchange of cloud i+1 is
    started before cloud i is consumed, two gather buffers alternate, shards differ in size (n is odd)."""
    dist = _init(rank, world, port)
    from point_cloud_toolbox_amd import shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature

    h = HostHandle()
    sc = ShardedCurvature(n, 5, rank, world, eps=0.25, handle=h, exchange=GlooExchange(rank, world))
    clouds = [shapes.torus_scan_order(n, world, rank, seed=100 + i) for i in range(steps + 1)]
    sc.upload_shard(clouds[0])
    ticket = sc.begin_exchange(0)
    for i in range(steps):
        cur = sc.end_exchange(ticket)
        sc.upload_shard(clouds[i + 1])                       # the send buffer is free once the gather has completed
        ticket = sc.begin_exchange(i + 1)                    # runs while `cur` is being consumed
        sc.run_device(cur)
    sc.end_exchange(ticket)
    assert all(s[1] == (sc.lo, sc.hi) and s[2] == 5 and s[3] == 0.25 for s in h.seen)
    if rank == 0:
        np.savez(os.path.join(out_dir, "pipe.npz"), *[s[0] for s in h.seen])
    dist.barrier()
    sc.close()
    dist.destroy_process_group()


def test_double_buffered_exchange_delivers_every_cloud_intact(tmp_path):
    import torch.multiprocessing as tmp_mp
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes

    world, n, steps, port = 2, 4001, 5, _free_port()
    tmp_mp.spawn(_pipeline_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "pipe.npz"))
    for i in range(steps):
        want = np.concatenate([shapes.torus_scan_order(n, world, r, seed=100 + i) for r in range(world)])
        assert np.array_equal(got[f"arr_{i}"], want), f"cloud {i}"


def _slab_pipeline_worker(rank, world, port, n, steps, out_dir):
    """The pipeline above with ownership by slab: after every pass the rank's records go out in a SECOND exchange --
    queued behind the gather of the next cloud, which is in flight -- and come back scattered into its index range."""
    dist = _init(rank, world, port)
    from point_cloud_toolbox_amd import shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature

    h = HostHandle()
    sc = ShardedCurvature(n, 5, rank, world, eps=0.25, handle=h, exchange=GlooExchange(rank, world), ownership="slab")
    assert sc.slab and sc.collective
    clouds = [shapes.torus_random(n, seed=200 + i, lo=sc.lo, hi=sc.hi) for i in range(steps + 1)]
    sc.upload_shard(clouds[0])
    ticket = sc.begin_exchange(0)
    for i in range(steps):
        cur = sc.end_exchange(ticket)
        sc.upload_shard(clouds[i + 1])
        ticket = sc.begin_exchange(i + 1)
        sc.run_device(cur)                                   # ends that exchange itself before the records go out
        K, H = sc.download()
        rows = np.arange(sc.lo, sc.hi, dtype=np.float32)
        assert np.array_equal(K, 2.0 * rows) and np.array_equal(H, -rows), (rank, i)
        assert h.seen[-1][1] == ("slab", rank, world)
    assert sc.end_exchange(ticket) is not None               # (asked twice: the same buffer, no second wait)
    if rank == 0:
        np.savez(os.path.join(out_dir, "slab_pipe.npz"), *[s_[0] for s_ in h.seen])
    dist.barrier()
    sc.close()
    dist.destroy_process_group()


def test_slab_ownership_sends_the_rows_back_in_a_second_exchange(tmp_path):
    import torch.multiprocessing as tmp_mp
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes

    world, n, steps, port = 2, 5001, 4, _free_port()
    tmp_mp.spawn(_slab_pipeline_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "slab_pipe.npz"))
    for i in range(steps):
        assert np.array_equal(got[f"arr_{i}"], shapes.torus_random(n, seed=200 + i)), f"cloud {i}"


def _slab_oracle_worker(rank, world, port, n, k, out_dir):
    dist = _init(rank, world, port)
    from point_cloud_toolbox_amd import shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature, shard_range
    import pct_oracle as oracle

    def compute(full, part, parts, kk, eps):                 # a slab checker: cut by the rank of the x coordinate
        order = np.argsort(full[:, 0], kind="stable")
        cuts = [len(order) * p // parts for p in range(parts + 1)]
        rows = np.sort(order[cuts[part]:cuts[part + 1]])
        r = oracle.pipeline_batched(full, kk, rows=rows, workers=1)
        return rows, r["K"], r["H"], [cuts[p + 1] - cuts[p] for p in range(parts)]

    lo, hi = shard_range(n, rank, world)
    local = shapes.torus_random(n, seed=22, lo=lo, hi=hi)
    sc = ShardedCurvature(n, k, rank, world, exchange=GlooExchange(rank, world), compute=compute, ownership="slab")
    K, H = sc.step(local)
    assert len(K) == hi - lo
    np.savez(os.path.join(out_dir, f"slab_out_{rank}.npz"), K=K, H=H)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_ownership_matches_single_process(tmp_path):
    import torch.multiprocessing as tmp_mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pct_oracle as oracle
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd import shapes

    n, k, port = 4101, 20, _free_port()
    tmp_mp.spawn(_slab_oracle_worker, args=(2, port, n, k, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(tmp_path / f"slab_out_{r}.npz") for r in range(2)]
    K, H = np.concatenate([p["K"] for p in parts]), np.concatenate([p["H"] for p in parts])
    ref = oracle.pipeline_batched(shapes.torus_random(n, seed=22), k, workers=1)
    assert np.array_equal(K, ref["K"]) and np.array_equal(H, ref["H"])


def test_records_that_do_not_partition_the_rows_are_refused(built):
    from point_cloud_toolbox_amd.dist import scatter_records_host
    rec = np.zeros((6, 3), np.float32)
    rec[:, 0] = np.array([4, 2, 3, 5, 0, 1], np.int32).view(np.float32)
    rec[:, 1] = np.arange(6)
    K, H = scatter_records_host(rec, 2, 5)
    assert K.tolist() == [1.0, 2.0, 0.0]
    with pytest.raises(RuntimeError, match="partition"):
        scatter_records_host(rec[1:], 2, 5)                  # row 4 is missing
    rec[1, 0] = rec[2, 0]
    with pytest.raises(RuntimeError, match="partition"):
        scatter_records_host(rec, 2, 5)                      # row 3 twice, row 2 missing


def _rdzv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import pointCloudToolbox  # noqa: F401
    from point_cloud_toolbox_amd.dist import rendezvous_unique_id
    calls = []

    def make_id():
        calls.append(1)
        return bytes((7 * i + world) % 256 for i in range(128))

    uid = rendezvous_unique_id(rank, world, make_id, addr="127.0.0.1", port=port, timeout=60.0)
    q.put((rank, uid, len(calls)))


def test_rendezvous_steps_past_an_occupied_port():
    """torchrun's own store sits on MASTER_PORT; should MASTER_PORT+1 be taken as well, rank 0 listens on the next free
    candidate and the peers find it (a foreign listener does not answer with the magic)."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    squatter = socket.socket()
    squatter.bind(("127.0.0.1", port))
    squatter.listen(4)
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import threading

    def drop():                                   # the foreign service accepts and says nothing useful
        squatter.settimeout(0.2)
        end = __import__("time").monotonic() + 20
        while __import__("time").monotonic() < end and any(p.is_alive() for p in procs):
            try:
                c, _ = squatter.accept()
                c.close()
            except socket.timeout:
                pass
    t = threading.Thread(target=drop)
    t.start()
    got = [q.get(timeout=60) for _ in range(2)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    t.join()
    squatter.close()
    assert got[0][1] == got[1][1] and len(got[0][1]) == 128


@pytest.mark.parametrize("world", [2, 4, 8])
def test_unique_id_rendezvous_reaches_every_rank(world):
    """What distributes the RCCL unique id: rank 0 generates it once, every rank ends up with the same 128 bytes --
    whatever the order in which the processes come up."""
    ctx = mp.get_context("spawn")
    port, q = _free_port(), ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in reversed(procs):                     # rank 0 last: the others must keep trying until it listens
        p.start()
    got = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = bytes((7 * i + world) % 256 for i in range(128))
    assert sorted(r for r, _, _ in got) == list(range(world))
    assert all(uid == want for _, uid, _ in got)
    assert sum(c for _, _, c in got) == 1 and [c for r, _, c in got if r == 0] == [1]
