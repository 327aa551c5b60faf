"""Tiny clouds (2..300 points): grid, chained and all-exact sweeps against the exhaustive one (developer tool).
python tools/fuzz_tiny.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi


def run(seed0=9, budget=None, cases=None):
    """Runs random cases until `budget` seconds or `cases` cases are done; returns (cases, first mismatch or None)."""
    t_end = time.time() + (budget if budget is not None else 1e9)
    it = 0
    while time.time() < t_end and (cases is None or it < cases):
        rng = np.random.default_rng([seed0, it])
        n = int(rng.integers(2, 300)); k = int(rng.integers(1, min(127, n - 1) + 1))
        kind = rng.integers(0, 4)
        pts = (rng.normal(size=(n, 3)) if kind == 0 else rng.uniform(0, 1, (n, 3)) * [1, 1, 0] if kind == 1
               else np.round(rng.uniform(0, 3, (n, 3))) if kind == 2 else np.repeat(rng.normal(size=(1, 3)), n, 0) + rng.normal(size=(n, 3)) * 1e-7)
        pts = np.ascontiguousarray(pts, dtype=np.float32 if rng.random() < 0.8 else np.float64)
        eps = float(rng.uniform(0.05, 2)) if rng.random() < 0.3 else 0.0
        if rng.random() < 0.2:                                         # magnitudes whose squares leave float32
            mag = 10.0 ** rng.uniform(-30, 30); pts = (pts.astype(np.float64) * mag).astype(pts.dtype); eps *= mag
        h = _capi.Handle(0); h.set_points(pts)
        h.curvature(k, eps, _capi.KNN_BRUTE)
        ib, db, cb = h.get_neighbors(0, n, want_count=True); cfb, Kb, Hb, _ = h.get_fit(0, n)
        for algo in (_capi.KNN_GRID, _capi.KNN_GRID_LEVELS, _capi.KNN_GRID_EXACT, _capi.KNN_TREE):
            if os.environ.get("FUZZ_TINY_VERBOSE"):
                print(f"tiny case ({seed0},{it}): n={n} k={k} kind={kind} eps={eps} dtype={pts.dtype} algo={algo}", file=sys.stderr, flush=True)
            h.curvature(k, eps, algo)
            ig, dg, cg = h.get_neighbors(0, n, want_count=True); cfg, Kg, Hg, _ = h.get_fit(0, n)
            if not (np.array_equal(ib, ig) and np.array_equal(db, dg) and np.array_equal(cb, cg) and np.array_equal(cfb, cfg, equal_nan=True)
                    and np.array_equal(Kb, Kg, equal_nan=True)):
                h.close()
                return it, f"seed=({seed0},{it}) n={n} k={k} kind={kind} eps={eps} dtype={pts.dtype} algo={algo}"
        h.close(); it += 1
    return it, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 9, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
    if bad:
        print("MISMATCH", bad, flush=True)
        sys.exit(1)
    print("tiny fuzz ok:", n_done)
