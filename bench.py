"""Headline bench: points/s of Gaussian+mean curvature (BASELINE.json metric).  No PyTorch in this process.

Default workload (--config c3 = BASELINE configs[2], the one the metric is quoted on): synthetic torus R=1, r=1/3,
random (theta, phi), seed 1234, float32, 1 000 000 points PER GPU at k=50 (weak scaling: N ranks hold an N-million-
point torus in scan order, rank r owns index range r = the r-th major-angle wedge).  One "step" = one full pass of the
hot path over the resident cloud: [N>1: RCCL all-gather of the coordinate shards, called from libpct_hip.so] ->
cell-list build -> k-NN sweep -> fused plane-align / quadric fit / K,H.  Inputs are in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          (torchrun = process launcher only)

Other configs (strong scaling, the cloud is fixed and sharded by index range):
    --config c4   egg carton 5 000 000 points, k=50                        (BASELINE configs[3])
    --config c5   sample_scans/bunny.txt tiled x557 = 20 022 479 points, k=80, hybrid eps=0.0062   (BASELINE configs[4])
    --verify      every rank checks the sampled reference goldens (tests/golden/g7_*) that fall into its range

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field definitions).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("LIBC_FATAL_STDERR_", "1")      # glibc's fatal diagnostics to stderr, not to the terminal
os.environ.setdefault("PCT_ABORT_TRACE", "1")          # a bare abort() anywhere: native backtrace of the raising thread first
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes: the host driver supports dmabuf IPC only
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def measured_traffic(n_points, k):
    """HBM bytes per sweep launch from the newest committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the file).
    A STATIC figure: counters cannot be read inside a timed run; the file it comes from is named next to it."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_knn_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if n_points == 1_000_000 and k == 50:
                return d["hbm_bytes_per_launch"], "profiles/" + os.path.basename(path) + " (static: PMC passes of this build's kernel, not this run)"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def measured_instructions():
    """Wave-instructions per launch from the newest committed SQ counter passes (STATIC, like `traffic`): what the
    issue-bound view of the kernels is computed from.  {kernel: {"valu": n, "salu": n, "lds": n}}, file name."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_counters.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            out = {}
            for name, c in d.items():
                out[name] = {"valu": c.get("SQ_INSTS_VALU"), "salu": c.get("SQ_INSTS_SALU"), "lds": c.get("SQ_INSTS_LDS")}
            return out, "profiles/" + os.path.basename(path)
        except (OSError, ValueError):
            continue
    return {}, None


def class_surface(local, k):
    """What a drop-in user of the reference's class gets (utils.py:481-501's call sequence without the meshing):
    PointCloud(points=...) -> plant_kdtree(k) -> compute_pointwise_explicit_quadratic_curvature() -> K, H as NumPy.
    A fresh object in the warm process; wall time per call."""
    from pointCloudToolbox import PointCloud
    n = len(local)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        pc = PointCloud(points=local, normals=np.zeros((n, 0)))
        t1 = time.perf_counter()
        pc.plant_kdtree(k, algorithm="grid")
        t2 = time.perf_counter()
        K, H = pc.compute_pointwise_explicit_quadratic_curvature()
        t3 = time.perf_counter()
        pc.close()
        r = {"constructor_ms": 1e3 * (t1 - t0), "plant_kdtree_ms": 1e3 * (t2 - t1), "curvature_ms": 1e3 * (t3 - t2),
             "total_ms": 1e3 * (t3 - t0)}
        if best is None or r["plant_kdtree_ms"] + r["curvature_ms"] < best["plant_kdtree_ms"] + best["curvature_ms"]:
            best = r
    # the fused extension of the class: one call, the neighbour table never leaves the device
    pc = PointCloud(points=local, normals=np.zeros((n, 0)))
    pc.compute_curvature_fused(k, algorithm="grid")
    t0 = time.perf_counter()
    pc.compute_curvature_fused(k, algorithm="grid")
    fused_ms = 1e3 * (time.perf_counter() - t0)
    pc.close()
    best.update({"value": n / (best["total_ms"] - best["constructor_ms"]) * 1e3, "unit": "points/s",
                 "what": "PointCloud(points=...) [the constructor's three matrix norms, pct:45-47, are host work and listed apart] -> "
                         "plant_kdtree(k) [upload 12 B/point, cell list, sweep writing indices AND distances] -> "
                         "compute_pointwise_explicit_quadratic_curvature() [fit + K, H to the host]; fresh object, warm process, best of 3; "
                         "value = points / (plant + curvature)",
                 "compute_curvature_fused_ms": fused_ms})
    return best


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
           "sample": f"{m} random rows of the same {n}-point cloud, k={k}, per-point loop "
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


def local_shard(cfg, shapes, n_total, rank, world, lo, hi):
    """Rows [lo, hi) of the config's cloud, generated by this rank alone."""
    if cfg == "c3":
        if world == 1:
            return shapes.torus_random(n_total, seed=1234), "random (theta,phi) seed 1234"
        # multi-GPU clouds arrive in scan order (the reference's own generator is theta-major, utils.py:888-891):
        # rank r's index range is the r-th major-angle wedge, random inside it
        return (shapes.torus_scan_order(n_total, world, rank, seed=1234),
                f"random (theta,phi) seed 1234 in scan order ({world} major-angle wedges = the index-range shards)")
    if cfg == "c4":
        return shapes.egg_carton_random(n_total, seed=1234, lo=lo, hi=hi), "(x,y) ~ U[-1,1]^2 seed 1234, unsorted"
    bunny = np.load(os.path.join(GOLDEN, "bunny_xyz_f32.npy"))
    per = len(bunny)
    t0, t1 = lo // per, (hi - 1) // per
    tiles = shapes.tile_cloud(bunny, 557, only=range(t0, t1 + 1))
    return np.ascontiguousarray(tiles[lo - t0 * per: hi - t0 * per]), "557 translated copies of the scan, tile after tile"


CONFIGS = {
    "c3": dict(k=50, eps=0.0, golden="g7_torus1m_k50_sample.npz", what="synthetic torus R=1 r=1/3", base="BASELINE configs[2]"),
    "c4": dict(k=50, eps=0.0, n_total=5_000_000, golden="g7_egg5m_k50_sample.npz", what="synthetic egg carton 0.1 sin(pi x) cos(pi y)",
               base="BASELINE configs[3]"),
    "c5": dict(k=80, eps=0.0062, n_total=20_022_479, golden="g7_bunny20m_k80_eps_sample.npz",
               what="sample_scans/bunny.txt tiled to 20 022 479 points, hybrid eps-ball query", base="BASELINE configs[4]"),
}


def verify_against_golden(handle, cfg, lo, hi, oracle, kh=None):
    """Sampled reference rows that fall into [lo, hi): indices and distances bit for bit, K/H at 1e-5.
    kh: (K, H) of rows [lo, hi) as a slab-ownership step returned them (the rank holds no neighbour rows of its index
    range then: K and H only)."""
    g = dict(np.load(os.path.join(GOLDEN, CONFIGS[cfg]["golden"])))
    sel = (g["rows"] >= lo) & (g["rows"] < hi)
    rows = g["rows"][sel]
    if len(rows) == 0:
        return 0
    if kh is None:
        idx, dist, cnt = handle.get_neighbor_rows(rows)
        if "count" in g:
            assert np.array_equal(cnt, g["count"][sel]), "valid-neighbour counts differ from the reference"
        assert np.array_equal(idx, g["idx"][sel]) and np.array_equal(dist, g["dists"][sel]), "neighbour rows differ from the reference"
        _, K, H, _ = handle.get_fit(lo, hi, coefs=False, H2=False)
    else:
        K, H = kh
    fK, fH = 1e-2 * np.nanmax(np.abs(g["K"])), 1e-2 * np.nanmax(np.abs(g["H"]))
    assert oracle.curvature_tolerance_ok(K[rows - lo], g["K"][sel], fK).all(), "K outside 1e-5 of the reference"
    assert oracle.curvature_tolerance_ok(H[rows - lo], g["H"][sel], fH).all(), "H outside 1e-5 of the reference"
    return int(len(rows))


def secondary_lattice(capi, shapes, k):
    """The reference's OWN 1 M-point torus (utils.py:883-896: a 1000 x 1000 (theta, phi) lattice, every point with
    symmetric partners at nearly equal distances) through the same step, on a fresh handle."""
    pts = shapes.torus_grid(1000)
    h = capi.Handle(0)
    h.set_points(pts)
    h.set_stats(True)
    h.curvature(k, 0.0, capi.KNN_GRID)
    redone = h.timings()["redone_queries"]
    h.set_stats(False)
    for _ in range(3):
        h.curvature(k, 0.0, capi.KNN_GRID)
    steps = 10
    h.synchronize()
    t0 = time.perf_counter()
    acc = {"grid_ms": 0.0, "knn_ms": 0.0, "knn_fast_ms": 0.0, "fit_ms": 0.0}
    for _ in range(steps):
        h.curvature(k, 0.0, capi.KNN_GRID)
        t = h.timings()
        for key in acc:
            acc[key] += t[key]
    h.synchronize()
    dt = time.perf_counter() - t0
    h.close()
    return {"workload": "the reference's own torus generator output: 1000 x 1000 (theta,phi) lattice (utils.py:883-896), float32, k=%d" % k,
            "value": len(pts) * steps / dt, "unit": "points/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "redo_fraction": redone / len(pts),
            "stage_ms": {"grid_build": acc["grid_ms"] / steps, "knn": acc["knn_ms"] / steps,
                         "knn_fast_kernel": acc["knn_fast_ms"] / steps, "fit_curvature": acc["fit_ms"] / steps}}


def secondary_uneven(capi, k):
    """A cloud whose density spans four decades -- a plane seen from one terrestrial laser station, density ~ 1/r^2 for
    r = 0.01 ... 1 -- through PCT_KNN_AUTO (default algorithm of the class surface): the census sends it to the
    hierarchical cell list.  A uniform cell list runs this cloud 13x slower (tools/density_probe.py)."""
    rng = np.random.default_rng(5)
    n = 1_000_000
    r, a = 0.01 * 100 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(a), r * np.sin(a)
    pts = np.ascontiguousarray(np.stack([x, y, 0.05 * np.sin(x) * np.cos(y)], 1), dtype=np.float32)
    h = capi.Handle(0)
    h.set_points(pts)
    h.set_stats(True)
    h.curvature(k, 0.0, capi.KNN_AUTO)
    t = h.timings()
    redone, algo = t["redone_queries"], t["algo"]
    h.set_stats(False)
    for _ in range(3):
        h.curvature(k, 0.0, capi.KNN_AUTO)
    steps = 10
    h.synchronize()
    t0 = time.perf_counter()
    acc = {"grid_ms": 0.0, "knn_ms": 0.0, "knn_fast_ms": 0.0, "fit_ms": 0.0}
    for _ in range(steps):
        h.curvature(k, 0.0, capi.KNN_AUTO)
        t = h.timings()
        for key in acc:
            acc[key] += t[key]
    h.synchronize()
    dt = time.perf_counter() - t0
    h.close()
    names = {capi.KNN_GRID: "uniform cell list", capi.KNN_GRID_LEVELS: "chain of cell lists", capi.KNN_TREE: "hierarchical cell list"}
    return {"workload": "plane scanned from one station: 1 M points, density ~ 1/r^2 over r = 0.01..1, float32, k=%d, PCT_KNN_AUTO" % k,
            "algorithm": names.get(algo, str(algo)),
            "value": n * steps / dt, "unit": "points/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "redo_fraction": redone / n,
            "stage_ms": {"build": acc["grid_ms"] / steps, "knn": acc["knn_ms"] / steps,
                         "knn_fast_kernel": acc["knn_fast_ms"] / steps, "fit_curvature": acc["fit_ms"] / steps}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=0, help="neighbours (default: the config's)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--verify", action="store_true", help="check the sampled reference goldens on every rank")
    ap.add_argument("--ownership", choices=("auto", "range", "slab"), default="auto",
                    help="multi-GPU: which rows a rank answers -- its index range, or a slab of equal population along the cloud's "
                         "longest axis (clouds in no spatial order; auto: slab for c4, range otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip first-call / end-to-end / lattice / repeat measurements")
    args = ap.parse_args()

    # ONE JSON line on stdout: whatever the libraries below print on it (RCCL 2.27 announces its version there at
    # communicator creation, hipcc may talk during a rebuild) goes to stderr instead -- at the descriptor level.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (e.g. torch.distributed.run)")
    force_dist = os.environ.get("PCT_FORCE_DIST") == "1"       # rehearse the multi-GPU code path with one rank

    import __graft_entry__ as ge
    ge.build()
    from point_cloud_toolbox_amd import _capi, shapes
    from point_cloud_toolbox_amd.dist import RcclExchange, ShardedCurvature, shard_range
    assert "torch" not in sys.modules, "bench.py runs without PyTorch"

    cfg = CONFIGS[args.config]
    k = args.k or cfg["k"]
    eps = cfg["eps"]
    n_total = cfg.get("n_total") or args.points_per_gpu * world
    lo, hi = shard_range(n_total, rank, world)
    weak = "n_total" not in cfg

    # ---- first call of a fresh process: library load, hipMalloc of every buffer, cell-size search from scratch ------
    first_call_ms = None
    if world == 1 and not args.no_extras and args.config == "c3":
        pts0 = shapes.torus_random(n_total, seed=99)
        t0 = time.perf_counter()
        h0 = _capi.Handle(local_rank)
        h0.set_points(pts0)
        h0.curvature(k, eps, _capi.KNN_GRID)
        h0.get_fit(0, 8, coefs=False, H2=False)
        first_call_ms = 1e3 * (time.perf_counter() - t0)
        h0.close()

    local, order = local_shard(args.config, shapes, n_total, rank, world, lo, hi)
    handle = _capi.Handle(local_rank)
    dist_mode = world > 1 or force_dist
    exchange = RcclExchange(handle, rank, world) if dist_mode else None

    def barrier():
        if exchange is not None:
            exchange.barrier()                                # drains this handle's streams, then RCCL all-reduce
        handle.synchronize()

    sc = None
    ownership = args.ownership if args.ownership != "auto" else ("slab" if args.config == "c4" else "range")
    if not dist_mode:
        handle.set_points(local)                              # resident before the timed region
        # A stream of clouds: the call returns once its kernels are enqueued, the host prepares the next step under the fit
        # of this one (pct_set_async; every getter and handle.synchronize() wait for the pending call).  The timed region
        # ends with a barrier that waits for everything; PCT_BENCH_SYNC_STEPS=1 runs the blocking calls instead.
        async_steps = os.environ.get("PCT_BENCH_SYNC_STEPS") != "1"
        handle.set_async(async_steps)

        def step():
            handle.curvature(k, eps, _capi.KNN_GRID)
    else:
        sc = ShardedCurvature(n_total, k, rank, world, eps=eps, handle=handle, exchange=exchange, ownership=ownership)
        # the local shard is resident on the device before the timed region; the exchange is part of the step.
        # Two gather buffers: the exchange of step i+1 (RCCL, exchange stream) overlaps the kernels of step i (compute
        # stream), as in a pipeline over a stream of clouds.  Every timed step still contains one full exchange and one
        # full compute pass; K steps = K exchanges + K passes inside the timed region.
        sc.upload_shard(local)
        state = {"i": 0, "ticket": sc.begin_exchange(0)}

        def step():
            i = state["i"]
            cur = sc.end_exchange(state["ticket"])
            state["ticket"] = sc.begin_exchange(i + 1)        # RCCL over xGMI: 12 B/point per rank
            sc.run_device(cur)
            state["i"] = i + 1

    def timed_region(steps):
        barrier()
        t0 = time.perf_counter()
        a_knn = a_fit = a_grid = a_fast = 0.0
        for _ in range(steps):
            step()
            tm = handle.stage_times_done()                    # hipEvent times of the last FINISHED step (asynchronous steps: the
            a_knn += tm.knn_ms                                # one before the step just enqueued), one reused struct
            a_fit += tm.fit_ms
            a_grid += tm.grid_ms
            a_fast += tm.knn_fast_ms
        barrier()
        dt = time.perf_counter() - t0
        acc = {"knn_ms": a_knn, "fit_ms": a_fit, "grid_ms": a_grid, "knn_fast_ms": a_fast}
        if exchange is not None:
            dt = float(exchange.allreduce([dt], "max")[0])
        return dt, acc, handle.timings()

    # The secondary measurement (the reference's own lattice torus, its own handle) runs BEFORE the headline region: its
    # order is free, and a GPU that has just worked holds its clocks -- the W warm-up steps alone (3 ms) leave the first
    # region 5-6 % slower than the same region repeated (repeat_ms_per_step).
    secondary = uneven = None
    if world == 1 and not dist_mode and not args.no_extras and args.config == "c3":
        secondary = secondary_lattice(_capi, shapes, k)
        uneven = secondary_uneven(_capi, k)
    # Clock ramp: a fresh process's first 20-step region (17 ms) otherwise measures the GPU's way up to its working clocks
    # (0.86 against 0.83 ms per step, the same regions repeated: repeat_ms_per_step); a production stream runs on a warm
    # device.  60 ms of the very same step, untimed, before the W warm-up steps; reported in the line (clock_ramp),
    # PCT_BENCH_RAMP_MS=0 switches it off.
    ramp_ms = float(os.environ.get("PCT_BENCH_RAMP_MS", "60"))
    ramp_steps = 0
    if dist_mode:
        # (every rank must issue the same collectives: a step count fixed from the request, not from a rank's clock)
        for _ in range(int(ramp_ms / 2.0)):
            step()
            ramp_steps += 1
    else:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < ramp_ms:
            step()
            ramp_steps += 1
    for _ in range(args.warmup):
        step()
    dt, acc, last_tm = timed_region(args.steps)

    verified = None
    if args.verify:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pct_oracle as oracle
        if args.config == "c3" and world > 1:
            mine = 0                                          # the scan-ordered weak-scaling cloud has no reference sample
        else:
            mine = verify_against_golden(handle, args.config, lo, hi, oracle, kh=sc.download() if sc is not None and sc.slab else None)
        tot = exchange.allreduce([mine, 1.0], "sum") if exchange is not None else np.array([mine, 1.0])
        verified = {"rows_checked": int(tot[0]), "ranks_seen": int(tot[1])}

    repeats = [1e3 * dt / args.steps]
    extras = {}
    if not args.no_extras:
        for _ in range(2):                                    # spread of the same region, same process, same box
            d2, _, _ = timed_region(args.steps)
            repeats.append(1e3 * d2 / args.steps)
        if not dist_mode:
            handle.set_stats(True)
            handle.curvature(k, eps, _capi.KNN_GRID)
            extras["redo_fraction"] = handle.timings()["redone_queries"] / n_total
            extras["fit_svd_rows"] = handle.timings()["fit_svd_rows"]
            handle.set_stats(False)
            # end to end: host coordinates in (12 B/point over PCIe), K and H out (8 B/point), every step
            m = max(10, args.steps // 2)
            t0 = 0.0
            for it in range(m + 1):                           # (round 0 warms the host pages of the output block)
                if it == 1:
                    handle.synchronize()
                    t0 = time.perf_counter()
                handle.set_points(local)
                handle.curvature(k, eps, _capi.KNN_GRID)
                handle.get_fit(0, n_total, coefs=False, H2=False)
            e2e = (time.perf_counter() - t0) / m
            extras["end_to_end"] = {"value": n_total / e2e, "unit": "points/s", "ms_per_step": 1e3 * e2e,
                                    "what": "H2D of float32 coordinates (12 B/point, pageable host memory) + step + D2H of K, H (8 B/point, one copy)"}
            # ... and for a STREAM of clouds from host memory: two handles of this GPU in turn, asynchronous calls -- the
            # transfers of one cloud (pageable host memory both ways) run while the device works on the other
            try:
                hb = _capi.Handle(local_rank)
                hb.set_async(True)
                pair = (handle, hb)
                stream_last = []
                t0 = 0.0
                for it in range(2 * m + 2):
                    hh = pair[it & 1]
                    if it >= 2:
                        got = hh.get_fit(0, n_total, coefs=False, H2=False)   # results of the cloud this handle took two turns ago
                        if it >= 2 * m:                                   # (kept for the check after the clock has stopped)
                            stream_last.append((got[1].copy(), got[2].copy()) if it == 2 * m else (got[1], got[2]))
                    if it == 2:
                        t0 = time.perf_counter()
                    if it < 2 * m + 2 - 2:
                        hh.set_points(local)
                        hh.curvature(k, eps, _capi.KNN_GRID)
                e2p = (time.perf_counter() - t0) / (2 * m - 2)
                hb.close()
                if len(stream_last) == 2 and not (np.array_equal(stream_last[0][0], stream_last[1][0], equal_nan=True)
                                                  and np.array_equal(stream_last[0][1], stream_last[1][1], equal_nan=True)):
                    raise RuntimeError("the two handles of the stream disagree")      # (the same cloud went to both: same bits)
                extras["end_to_end_stream"] = {"value": n_total / e2p, "unit": "points/s", "ms_per_cloud": 1e3 * e2p,
                                               "what": "the same transfers and step for a stream of clouds: two handles in turn, "
                                                       "pct_set_async -- one cloud's H2D / D2H overlap the other's kernels"}
            except Exception as ex:                                       # (an extra: never fails the line)
                extras["end_to_end_stream"] = {"error": repr(ex)}
            # the sweep in the mode SURVEY 8(d) prices: plant_kdtree's, which writes indices AND distances (12 + 8k B/point)
            acc8 = 0.0
            for _ in range(10):
                handle.knn(k, eps, _capi.KNN_GRID)
                acc8 += handle.timings()["knn_fast_ms"]
            extras["_knn_fast_ms_with_distances"] = acc8 / 10
            handle.curvature(k, eps, _capi.KNN_GRID)
            extras["class_surface"] = class_surface(local, k)
            # a fresh handle in the warm process: every buffer allocated again, cell size searched from scratch
            t0 = time.perf_counter()
            h1 = _capi.Handle(local_rank)
            h1.set_points(local)
            h1.curvature(k, eps, _capi.KNN_GRID)
            h1.get_fit(0, 8, coefs=False, H2=False)
            extras["fresh_handle_first_call_ms"] = 1e3 * (time.perf_counter() - t0)
            extras["fresh_handle_grid_passes"] = h1.timings()["grid_iters"]
            h1.close()
            if first_call_ms is not None:
                extras["first_call_ms"] = first_call_ms

    if rank == 0:
        steps = args.steps
        nq = hi - lo
        fast_ms = acc["knn_fast_ms"]
        knn_avg_s = fast_ms / steps / 1e3                     # the dominant kernel alone (hipEvents on the handle's stream)
        # The timed step is the FUSED call: its sweep writes the index table only (12 + 4k B/point) -- the fit never
        # reads distances, pct_get_neighbors derives them on demand.  SURVEY 8d's B_knn(k) = 12 + 8k prices the sweep
        # that writes both (plant_kdtree's): measured next to it when the extras run (roofline.with_distances).
        algo_bytes = nq * (12 + 4 * k)
        achieved = algo_bytes / knn_avg_s / 1e9
        traffic, traffic_source = measured_traffic(nq, k)
        par = f"point-index-range shards x{world}"
        if sc is not None and sc.slab:
            par = (f"each rank HOLDS an index range and ANSWERS one of {world} slabs of equal population along the cloud's longest axis "
                   "(pct_set_query_slab: the cloud is in no spatial order); the rows return to their holders as (index, K, H) records, "
                   "12 B/point, in a second exchange per step")
        issued = None
        if dist_mode:
            issued = handle.comm_counters()
            forms = []
            if issued["allgather"]:
                forms.append(f"{issued['allgather']} x ncclAllGather straight into the gather buffer")
            if issued["padded_allgather"]:
                forms.append(f"{issued['padded_allgather']} x ncclAllGather of shards padded to the largest + compaction pass")
            if issued["broadcast_groups"]:
                forms.append(f"{issued['broadcast_groups']} x group of per-rank ncclBroadcast")
            if forms:
                par += (" + RCCL exchange of the float32 coordinates over xGMI, called from libpct_hip.so on the handle's exchange "
                        "stream (" + "; ".join(forms) + "; no PyTorch in the process; double-buffered, overlapped with the previous "
                        "pass), each rank keeps the points near its range")
            else:
                par += (" -- multi-GPU driver with a world of one rank: communicator and buffers set up, no collective issued "
                        "(PCT_COMM_FORCE=allgather|padded|bcast issues one per step)")
        out = {
            "metric": "points/sec curvature (1M-pt torus, k=50); HBM GB/s k-NN vs peak",
            "value": n_total * steps / dt,
            "unit": "points/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if args.config != "c5" else "the reference's sample scan, tiled",
            "config": {"workload": f"{cfg['what']}, {order}, float32, "
                                   + (f"{args.points_per_gpu} points per GPU ({n_total} total)" if weak else f"{n_total} points in all")
                                   + f", k={k}" + (f", eps={eps}" if eps else "") + ", grid k-NN + fused plane-align/quadric-fit/curvature "
                                   f"({cfg['base']})",
                       "points_total": n_total, "k": k, "parallelism": par,
                       **({"ownership": "slab" if sc.slab else "range"} if sc is not None else {}),
                       **({"collectives_issued": issued} if issued is not None else {})},
            "roofline": {"bound": "hbm", "kernel": "k_knn_pair, no distance table (the fused step's sweep)", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_bytes_per_point": "12 + 4k (coordinates in, index table out)",
                         "avg_launch_ms": fast_ms / steps},
            "stage_ms": {"grid_build": acc["grid_ms"] / steps, "knn": acc["knn_ms"] / steps, "knn_fast_kernel": fast_ms / steps,
                         "fit_curvature": acc["fit_ms"] / steps},
            "repeat_ms_per_step": repeats,
            "calls": ("asynchronous (pct_set_async): a step returns once its kernels are enqueued, the timed region ends with a wait for "
                      "everything; PCT_BENCH_SYNC_STEPS=1 for blocking calls") if (not dist_mode and async_steps) else "blocking",
            "clock_ramp": {"ms": ramp_ms, "steps": ramp_steps,
                           "what": "untimed runs of the same step before the W warm-up steps, so that the timed region starts at working clocks"},
            "grid_points_per_rank": last_tm["grid_points"],
            "target_points_per_s": 1e7,
        }
        w8 = extras.pop("_knn_fast_ms_with_distances", None)
        if w8:
            b8 = nq * (12 + 8 * k)
            out["roofline"]["with_distances"] = {
                "what": "the same kernel in plant_kdtree's mode (indices AND float32 distances written): SURVEY 8d's B_knn(k) = 12 + 8k",
                "algorithmic_bytes_per_launch": b8, "avg_launch_ms": w8, "achieved": b8 / w8 / 1e6, "unit": "GB/s",
                "frac": b8 / w8 / 1e6 / HBM_PEAK_GBPS}
        # every kernel of the step against the resource that binds it (none is HBM-bound: DESIGN 4).  Issue view: a
        # vector instruction other than plain float32 arithmetic occupies a SIMD for 4 cycles on this part
        # (tools/ubench/valu_rate.hip: v_med3, DPP moves, v_cmp, integer, every fp64 op; v_fma_f32 2.5), 1024 SIMDs,
        # 2.4 GHz peak clock; instruction counts are STATIC figures from the committed SQ counter passes.
        if args.config == "c3" and nq == 1_000_000 and k == 50 and not dist_mode:
            insts, insts_src = measured_instructions()
            simd_cycles = 1024 * 2.4e9
            rk = {}

            def issue(kernel, ms):
                v = (insts.get(kernel) or {}).get("valu")
                return {} if not v or not ms else {"valu_insts_per_launch": v, "frac_of_valu_issue_peak": v * 4.0 / (ms * 1e-3 * simd_cycles)}
            rk["k_knn_pair"] = {"bound": "valu issue", **issue("k_knn_pair", fast_ms / steps), "avg_launch_ms": fast_ms / steps,
                                "hbm_frac": achieved / HBM_PEAK_GBPS}
            fit_ms_ = acc["fit_ms"] / steps
            fit_bytes = nq * (16 * k + 48)                             # B_fit(k) = 16k + 48
            rk["k_fit"] = {"bound": "valu issue (fp64)", **issue("k_fit", fit_ms_), "avg_launch_ms": fit_ms_,
                           "algorithmic_bytes_per_launch": fit_bytes, "hbm_achieved_GBps": fit_bytes / fit_ms_ / 1e6,
                           "hbm_frac": fit_bytes / fit_ms_ / 1e6 / HBM_PEAK_GBPS}
            grid_ms_ = acc["grid_ms"] / steps
            grid_bytes = nq * 76
            rk["cell_list_build"] = {"bound": "launch chain + one host synchronisation", "avg_ms": grid_ms_,
                                     "algorithmic_bytes": grid_bytes, "hbm_achieved_GBps": grid_bytes / grid_ms_ / 1e6,
                                     "hbm_frac": grid_bytes / grid_ms_ / 1e6 / HBM_PEAK_GBPS}
            rk["instruction_counts_source"] = insts_src
            out["roofline_kernels"] = rk
        out.update(extras)
        if verified is not None:
            out["verified"] = verified
        if secondary is not None:
            out["secondary"] = secondary
        if uneven is not None:
            out["secondary_uneven_density"] = uneven
        if world == 1 and not args.no_cpu_baseline and args.config == "c3":
            out["cpu_baseline"] = cpu_baseline(local, k)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if exchange is not None:
        exchange.barrier()
    if sc is not None:
        sc.close()
    handle.close()


if __name__ == "__main__":
    main()
