"""Point-index-range sharding of the curvature path over up to 8 GPUs.

The reference is single-process (nothing to mirror); queries are
independent, so rank r owns the contiguous index range
``[r*N/G, (r+1)*N/G)`` of the cloud.  The only exchange is one all-gather of the
float32 coordinates (12 B/point) so that every GPU holds the whole candidate
set -- for an unsorted cloud the "halo" of an index range is the whole cloud.
Outputs need no collective: each rank keeps its K/H rows, neighbour indices are
global, and results are independent of G.

``torch.distributed`` is plumbing here (process group, RCCL all-gather when the
backend is "nccl", gloo on CPU); the compute goes through the C ABI.

Load order: PyTorch-ROCm bundles its own ``libamdhip64.so.7``.  A process that
uses both must ``import torch`` BEFORE the first ``_capi.load()`` so that the
dynamic loader binds ``libpct_hip.so`` to the HIP runtime that is already
loaded; two HIP runtimes in one process cannot both open the device.
"""
from __future__ import annotations

import numpy as np

__all__ = ["shard_range", "allgather_points", "ShardedCurvature"]


def shard_range(n_total, rank, world):
    """Global index range owned by ``rank``."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def allgather_points(local_pts, n_total, rank, world, device=None):
    """All-gather the per-rank coordinate shards into the full (N,3) float32 cloud.

    Returns a torch tensor on ``device`` (RCCL over xGMI for backend "nccl") or
    on the CPU (gloo).  Shards may differ in size by one row; they are padded to
    the common maximum for the collective and compacted afterwards.
    """
    import torch
    import torch.distributed as dist

    lo, hi = shard_range(n_total, rank, world)
    local_pts = np.ascontiguousarray(local_pts, dtype=np.float32)
    if local_pts.shape != (hi - lo, 3):
        raise ValueError(f"rank {rank} must hold rows [{lo},{hi}) of the cloud, got {local_pts.shape}")
    dev = torch.device("cpu") if device is None else torch.device(device)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    m = max(sizes)
    pad = torch.zeros((m, 3), dtype=torch.float32, device=dev)
    pad[: hi - lo] = torch.from_numpy(local_pts).to(dev)
    if world == 1:
        return pad[: hi - lo]
    gathered = torch.empty((world, m, 3), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(gathered.view(world * m, 3), pad)
    if all(s == m for s in sizes):
        return gathered.view(world * m, 3)
    return torch.cat([gathered[r, : sizes[r]] for r in range(world)], 0).contiguous()


class ShardedCurvature:
    """Per-rank driver: all-gather -> neighbour sweep + fit on the owned range.

    ``compute`` is the per-rank kernel entry.  On a GPU rank it is left ``None``
    and the HIP path is used through ``handle``; CPU tests of the sharding logic
    inject a checker with the same signature
    ``compute(full_points(np.ndarray), lo, hi, k, eps) -> (K, H)``.
    """

    def __init__(self, n_total, k, rank, world, eps=None, handle=None, device=None, compute=None):
        self.n_total, self.k, self.rank, self.world, self.eps = int(n_total), int(k), rank, world, eps
        self.handle, self.device, self.compute = handle, device, compute
        self.lo, self.hi = shard_range(self.n_total, rank, world)
        if handle is None and compute is None:
            raise RuntimeError("ShardedCurvature needs a device handle (HIP path); there is no CPU fallback")

    def step(self, local_pts):
        """One pass: exchange + compute.  Returns (K, H) for rows [lo, hi)."""
        full = allgather_points(local_pts, self.n_total, self.rank, self.world, self.device)
        if self.compute is not None:
            return self.compute(full.cpu().numpy(), self.lo, self.hi, self.k, self.eps)
        self.run_device(full)
        return self.download()

    def run_device(self, full):
        """Hand the gathered device buffer to the HIP path in place (no copy): ``full`` must stay unchanged
        until the next call."""
        import torch
        torch.cuda.current_stream(full.device).synchronize()     # the collective ran on torch's stream
        h = self.handle
        h.use_points_device(full.data_ptr(), self.n_total)
        h.set_query_range(self.lo, self.hi)
        h.curvature(self.k, self.eps or 0.0)

    # A stream of clouds: the all-gather of the next cloud runs (RCCL's own stream) while the kernels of the
    # current one run on the handle's stream.  Two gather buffers alternate.
    def begin_exchange(self, local_dev, out):
        """Start the all-gather of ``local_dev`` (this rank's (N/G,3) device shard) into ``out`` ((N,3) device)."""
        import torch.distributed as dist
        return dist.all_gather_into_tensor(out, local_dev, async_op=True)

    def end_exchange(self, ticket, out):
        """Block until the exchange started by ``begin_exchange`` has filled ``out``."""
        import torch
        ticket.wait()
        if out.device.type == "cuda":
            torch.cuda.current_stream(out.device).synchronize()
        return out

    def download(self):
        _, K, H, _ = self.handle.get_fit(self.lo, self.hi, coefs=False, H2=False)
        return K, H


def gather_to_rank0(K, H, n_total, rank, world):
    """Host-side concatenation of the per-rank rows (outputs need no device collective)."""
    import torch.distributed as dist
    if world == 1:
        return K, H
    parts = [None] * world if rank == 0 else None
    dist.gather_object((np.asarray(K), np.asarray(H)), parts, dst=0)
    if rank != 0:
        return None, None
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
