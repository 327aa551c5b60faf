"""One handle fed a stream of clouds (developer tool): the warm-start state a handle keeps between calls -- cell-edge
hint, speculative bounding box, reused culling box -- must never change a result.  Same-size clouds that move, shrink,
grow or change density between calls, owned ranges that change or stay, zero-copy device buffers rewritten in place;
every answer is compared with the exhaustive sweep of a fresh handle.  python tools/fuzz_stream.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes


def cloud(rng, n):
    kind = rng.integers(0, 5)
    if kind == 0: p = shapes.torus_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64)
    elif kind == 1: p = shapes.egg_carton_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64)
    elif kind == 2: p = rng.normal(size=(n, 3)) * [1, 1, 0.02]
    elif kind == 3:
        p = shapes.torus_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64); p[: n // 4] *= 0.2
    else: p = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), 0.05 * rng.normal(size=n)], 1)
    return p


N_MAX = int(os.environ.get("FUZZ_N_MAX", "40000"))      # largest cloud of the stream


def run(seed0=0, budget=None, cases=None, verbose=True):
    t_end = time.time() + (budget if budget is not None else 1e9)
    rng = np.random.default_rng([seed0, 4242])
    h = _capi.Handle(0)
    dev = None
    n = int(rng.integers(2000, N_MAX)); it = 0
    lo, hi = 0, n
    while time.time() < t_end and (cases is None or it < cases):
        r = rng.random()
        if r < 0.25: n = int(rng.integers(2000, N_MAX)); lo, hi = 0, n           # a different size
        pts = cloud(rng, n)
        r = rng.random()
        if r < 0.3: pts = pts * 10.0 ** rng.uniform(-2, 2)                           # same size, another scale
        if rng.random() < 0.3: pts = pts + rng.normal(size=3) * 10.0 ** rng.uniform(0, 3)   # ... moved away
        pts = np.ascontiguousarray(pts, dtype=np.float32 if rng.random() < 0.85 else np.float64)
        if rng.random() < 0.4:                                                      # owned range: new, or kept
            lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
        lo, hi = min(lo, n - 1), min(max(hi, lo + 1), n)
        k = int(rng.choice([5, 20, 50, 64, 90])); k = min(k, n - 1)
        eps = float(np.ptp(pts, axis=0).max()) * 10.0 ** rng.uniform(-2.3, -1.0) if rng.random() < 0.25 else 0.0
        if verbose: print(f"case {it}: n={n} [{lo},{hi}) k={k} eps={eps:.3g} {pts.dtype} ...", flush=True)
        if pts.dtype == np.float32 and rng.random() < 0.5:
            # zero-copy hand-over, the multi-GPU step's way: ONE device buffer, rewritten in place for every cloud
            if dev is None: dev = h.device_alloc(N_MAX * 12)
            h.device_upload(dev, pts)
            h.use_points_device(dev, n)
        else:
            h.set_points(pts)
        h.set_query_range(lo, hi)
        # one handle, every structure in turn: uniform list, hierarchical list (a shard asked of it: the chain), the default's choice
        algo = int(rng.choice([_capi.KNN_GRID, _capi.KNN_GRID, _capi.KNN_TREE, _capi.KNN_AUTO]))
        h.curvature(k, eps, algo)
        if verbose: print(f"   sweep done (asked {algo}, ran {h.timings()['algo']})", flush=True)
        got = h.get_neighbors(lo, hi, want_count=True) + h.get_fit(lo, hi)[:3]
        f = _capi.Handle(0); f.set_points(pts); f.set_query_range(lo, hi); f.curvature(k, eps, _capi.KNN_BRUTE)
        want = f.get_neighbors(lo, hi, want_count=True) + f.get_fit(lo, hi)[:3]
        f.close()
        ok = all(np.array_equal(w, g, equal_nan=True) for w, g in zip(want, got))
        if verbose: print(f"   ok={ok}", flush=True)
        if not ok:
            if dev is not None: h.device_free(dev)
            h.close()
            return it, f"seed={seed0} case {it}: n={n} [{lo},{hi}) k={k} eps={eps} dtype={pts.dtype}"
        it += 1
    if dev is not None: h.device_free(dev)
    h.close()
    return it, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, verbose=os.environ.get("FUZZ_VERBOSE") == "1")
    if bad:
        print("MISMATCH", bad, flush=True); sys.exit(1)
    print(f"stream fuzz ok: {n_done} cases")
