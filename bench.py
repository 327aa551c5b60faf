"""Headline bench: points/s of Gaussian+mean curvature (BASELINE.json metric).

Workload (config.workload): synthetic torus R=1, r=1/3, random (theta, phi), seed 1234, float32,
1 000 000 points PER GPU at k=50 (BASELINE configs[2]; weak scaling: N ranks hold an N-million-point
torus in scan order, rank r owns index range r = the r-th major-angle wedge).  One "step" = one full pass of the hot path over the resident
cloud: [N>1: RCCL all-gather of the coordinate shards] -> cell-list build -> k-NN sweep -> fused
plane-align / quadric fit / K,H.  Inputs are in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def measured_traffic(n_points, k):
    """HBM bytes per sweep launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the file)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_o_knn_traffic.json")) as f:
            d = json.load(f)
        if n_points == 1_000_000 and k == 50:
            return d["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(pts, k, seconds_target=15.0):
    """1-core reference-faithful port (oracle loop) on a bounded sample of the same cloud."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pct_oracle as oracle
    from scipy.spatial import cKDTree
    n = len(pts)
    rng = np.random.default_rng(7)
    tree = cKDTree(np.array(pts, dtype=np.float32))           # build is excluded, as the GPU's upload is
    probe = rng.choice(n, 2000, replace=False)
    t0 = time.perf_counter()
    _run_loop(oracle, tree, pts, probe, k)
    rate = 2000 / (time.perf_counter() - t0)
    m = int(min(n, max(2000, rate * seconds_target)))
    rows = rng.choice(n, m, replace=False)
    t0 = time.perf_counter()
    _run_loop(oracle, tree, pts, rows, k)
    dt = time.perf_counter() - t0
    out = {"value": m / dt, "unit": "points/s", "cores": 1, "kind": "port",
           "sample": f"{m} random rows of the same {n}-point torus, k={k}, per-point loop "
                     f"(cKDTree.query + cov/svd + lstsq, oracle/pct_oracle.py), tree build excluded, {dt:.1f} s"}
    # the vectorised restatement on every host core (threaded cKDTree.query + batched eigh / solve)
    try:
        mb = int(min(n, 40_000))
        t0 = time.perf_counter()
        oracle.pipeline_batched(pts, k, rows=np.sort(rng.choice(n, mb, replace=False)), workers=-1, tree=tree)
        rate_b = mb / (time.perf_counter() - t0)
        mb = int(min(n, max(mb, rate_b * 10.0)))                  # about 10 s of work
        rows_b = np.sort(rng.choice(n, mb, replace=False))
        t0 = time.perf_counter()
        oracle.pipeline_batched(pts, k, rows=rows_b, workers=-1, tree=tree)
        dtb = time.perf_counter() - t0
        out["all_cores"] = {"value": mb / dtb, "unit": "points/s", "cores": os.cpu_count(),
                            "sample": f"{mb} random rows, vectorised restatement (oracle.pipeline_batched), {dtb:.1f} s"}
    except Exception as e:      # the headline baseline above stands on its own
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def _run_loop(oracle, tree, pts, rows, k):
    for i in rows:
        d, nb = tree.query(pts[i], k + 1)
        c = oracle.quadric_fit(oracle.plane_align(pts[nb[1:]] - pts[i]))
        oracle.quadric_curvatures(c)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    force_dist = os.environ.get("PCT_FORCE_DIST") == "1"       # rehearse the multi-GPU code path with one rank
    dist = torch = None
    if world > 1 or force_dist:
        # torch ships its own libamdhip64.so.7: it must be loaded BEFORE libpct_hip.so so that the process has one
        # HIP runtime (the loader then binds our library to the copy that is already there)
        import torch
        import torch.distributed as dist

    import __graft_entry__ as ge
    ge.build()
    from point_cloud_toolbox_amd import _capi, shapes
    from point_cloud_toolbox_amd.dist import ShardedCurvature, shard_range

    k = args.k
    n_total = args.points_per_gpu * world
    lo, hi = shard_range(n_total, rank, world)
    if world == 1:
        local = shapes.torus_random(n_total, seed=1234)
        order = "random (theta,phi) seed 1234"
    else:
        # multi-GPU clouds arrive in scan order (the reference's own generator is theta-major, utils.py:888-891):
        # rank r's index range is the r-th major-angle wedge, random inside it
        local = shapes.torus_scan_order(n_total, world, rank, seed=1234)
        order = f"random (theta,phi) seed 1234 in scan order ({world} major-angle wedges = the index-range shards)"

    if dist is not None:
        torch.cuda.set_device(local_rank)
        if force_dist and "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    handle = _capi.Handle(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        handle.synchronize()

    if dist is None:
        handle.set_points(local)                              # resident before the timed region

        def step():
            handle.curvature(k, 0.0, _capi.KNN_GRID)
    else:
        dev = torch.device("cuda", local_rank)
        sc = ShardedCurvature(n_total, k, rank, world, handle=handle, device=dev)
        sizes = {shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)}
        assert len(sizes) == 1, "bench shards are equal-sized"
        # the local shard is resident on the device before the timed region; the exchange is part of the step
        local_dev = torch.from_numpy(local).to(dev)
        # two gather buffers: the exchange of step i+1 (RCCL stream) overlaps the kernels of step i (handle's
        # stream), as in a pipeline over a stream of clouds.  Every timed step still contains one full exchange
        # and one full compute pass; K steps = K exchanges + K passes inside the timed region.
        bufs = [torch.empty((n_total, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        state = {"i": 0, "ticket": None}
        state["ticket"] = sc.begin_exchange(local_dev, bufs[0])

        def step():
            i = state["i"]
            cur = sc.end_exchange(state["ticket"], bufs[i % 2])
            state["ticket"] = sc.begin_exchange(local_dev, bufs[(i + 1) % 2])   # RCCL over xGMI: 12 B/point per rank
            sc.run_device(cur)
            state["i"] = i + 1

    for _ in range(args.warmup):
        step()
    knn_ms = fit_ms = grid_ms = fast_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = handle.timings()                                 # hipEvent times recorded on the handle's stream
        knn_ms += tm["knn_ms"]; fit_ms += tm["fit_ms"]; grid_ms += tm["grid_ms"]; fast_ms += tm["knn_fast_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        steps = args.steps
        nq = hi - lo
        knn_avg_s = fast_ms / steps / 1e3                   # the dominant kernel (k_knn_fast) alone
        algo_bytes = nq * (12 + 8 * k)                        # B_knn(k) = 12 + 8k per point
        achieved = algo_bytes / knn_avg_s / 1e9
        out = {
            "metric": "points/sec curvature (1M-pt torus, k=50); HBM GB/s k-NN vs peak",
            "value": n_total * steps / dt,
            "unit": "points/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic torus R=1 r=1/3, {order}, float32, "
                                   f"{args.points_per_gpu} points per GPU ({n_total} total), k={k}, grid k-NN + "
                                   "fused plane-align/quadric-fit/curvature (BASELINE configs[2])",
                       "points_total": n_total, "k": k,
                       "parallelism": f"point-index-range shards x{world}" + (" + RCCL all-gather of coordinates (double-buffered, overlapped with "
                                      "the previous pass), each rank keeps the points near its range" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "k_knn_fast", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": measured_traffic(nq, k),
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": fast_ms / steps},
            "stage_ms": {"grid_build": grid_ms / steps, "knn": knn_ms / steps, "knn_fast_kernel": fast_ms / steps,
                         "fit_curvature": fit_ms / steps},
            "target_points_per_s": 1e7,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(local, k)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    handle.close()


if __name__ == "__main__":
    main()
