"""Large random clouds: the fast sweep (+ redo) and the hierarchical cell list against the all-exact sweep of the uniform cell list, sharded too
(developer tool).  python tools/fuzz_big.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes


def make_case(seed0, it, n_max_log10=6.3):
    """The random case (seed0, it): returns (rng, pts, n, k, kind, eps)."""
    rng = np.random.default_rng([seed0, it, 99])
    n = int(10 ** rng.uniform(5, n_max_log10))
    k = int(rng.choice([8, 20, 30, 50, 63, 64, 80, 100, 127]))
    kind = int(rng.integers(0, 5))
    if kind == 0: pts = shapes.torus_random(n, seed=int(rng.integers(1 << 30)))
    elif kind == 1: pts = shapes.egg_carton_random(n, seed=int(rng.integers(1 << 30)))
    elif kind == 2: pts = shapes.fibonacci_sphere(n) * rng.uniform(0.1, 10)
    elif kind == 3:
        pts = shapes.torus_random(n, seed=int(rng.integers(1 << 30)))
        pts[: n // 3] *= 0.3                                   # a denser copy inside
    else:
        xy = rng.uniform(-1, 1, size=(n, 2)) * rng.uniform(0.2, 1.0, size=(n, 1)) ** 2
        pts = np.stack([xy[:, 0], xy[:, 1], 0.1 * np.sin(4 * xy[:, 0])], 1)      # density falls off outwards
    if rng.random() < 0.12:                                        # magnitudes whose squares leave float32
        pts = pts.astype(np.float64) * 10.0 ** rng.choice([-28, -17, 14, 21, 29])
        kind += 10
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    eps = 0.0
    if rng.random() < 0.25:
        eps = float(np.ptp(pts, axis=0).max()) * 10.0 ** rng.uniform(-2.7, -1.5)
    return rng, pts, n, k, kind, eps


def run(seed0, budget=None, cases=None, verbose=True, n_max_log10=6.3):
    """Runs random cases until `budget` seconds or `cases` cases are done; returns (cases, first mismatch or None)."""
    t_end = time.time() + (budget if budget is not None else 1e9)
    it = 0
    while time.time() < t_end and (cases is None or it < cases):
        rng, pts, n, k, kind, eps = make_case(seed0, it, n_max_log10)
        if verbose: print(f"case {it}: n={n} k={k} kind={kind} eps={eps:.4g}", flush=True)
        t0 = time.time()
        h = _capi.Handle(0)
        h.set_points(pts)
        h.curvature(k, eps, _capi.KNN_GRID_EXACT)
        ie, de, ce = h.get_neighbors(0, n, want_count=True)
        cfe, Ke, He, _ = h.get_fit(0, n)
        t1 = time.time()
        h.curvature(k, eps, _capi.KNN_GRID)
        tg = h.timings()["total_ms"]
        ig, dg, cg = h.get_neighbors(0, n, want_count=True)
        cfg, Kg, Hg, _ = h.get_fit(0, n)
        ok = (np.array_equal(ie, ig) and np.array_equal(de, dg) and np.array_equal(ce, cg) and np.array_equal(cfe, cfg, equal_nan=True)
              and np.array_equal(Ke, Kg, equal_nan=True) and np.array_equal(He, Hg, equal_nan=True))
        # the hierarchical cell list (and, for what it declines, the chain) over the same cloud
        h.curvature(k, eps, _capi.KNN_TREE)
        tt = h.timings()
        it3, dt3, ct3 = h.get_neighbors(0, n, want_count=True)
        cft, Kt, Ht, _ = h.get_fit(0, n)
        ok = ok and (np.array_equal(ie, it3) and np.array_equal(de, dt3) and np.array_equal(ce, ct3) and np.array_equal(cfe, cft, equal_nan=True)
                     and np.array_equal(Ke, Kt, equal_nan=True) and np.array_equal(He, Ht, equal_nan=True))
        if verbose: print(f"   tree step {tt['total_ms']:.2f} ms (ran {tt['algo']})", flush=True)
        lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
        h.set_query_range(lo, hi)
        h.curvature(k, eps, _capi.KNN_GRID)
        i2, d2, c2 = h.get_neighbors(lo, hi, want_count=True)
        _, K2, H2, _ = h.get_fit(lo, hi)
        ok2 = (np.array_equal(i2, ie[lo:hi]) and np.array_equal(d2, de[lo:hi]) and np.array_equal(c2, ce[lo:hi])
               and np.array_equal(K2, Ke[lo:hi], equal_nan=True) and np.array_equal(H2, He[lo:hi], equal_nan=True))
        h.close()
        if verbose: print(f"   exact {t1 - t0:.2f} s, grid step {tg:.2f} ms, ok={ok} shard_ok={ok2}", flush=True)
        if not (ok and ok2):
            return it, f"seed=({seed0},{it}) n={n} k={k} kind={kind} eps={eps} whole={ok} shard {lo}:{hi}={ok2}"
        it += 1
    return it, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
    if bad:
        print("MISMATCH", bad, flush=True)
        sys.exit(1)
    print(f"done: {n_done} cases, no mismatch")
