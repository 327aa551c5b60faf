"""Randomised cross-check of the grid sweep (plain, sharded, chained) against the exhaustive sweep (developer tool).
python tools/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes

F64_SHARE = float(os.environ.get("FUZZ_F64_SHARE", "0.15"))       # share of float64 clouds


def make_case(seed0, it):
    """The random case (seed0, it): returns (rng, pts, n, k, kind, eps)."""
    rng = np.random.default_rng([seed0, it])
    n = int(rng.integers(200, 60_000))
    k = int(rng.integers(1, min(127, n - 1) + 1))
    if rng.random() < 0.2:
        k = min(int(rng.choice([1, 5, 31, 32, 47, 52, 63, 64, 65, 126, 127])), n - 1)
    kind = rng.integers(0, 11)
    if kind == 0:
        pts = shapes.torus_random(n, seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        pts = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3)
    elif kind == 2:
        pts = np.round(rng.uniform(-1, 1, size=(n, 3)) * rng.integers(3, 200)) / 16.0        # lattice: ties
    elif kind == 3:
        c = rng.uniform(-1, 1, size=(8, 3)); w = rng.integers(0, 8, size=n)
        pts = c[w] + rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-4, -0.5, size=8))[w, None]
    elif kind == 4:
        pts = shapes.egg_carton_random(n, seed=int(rng.integers(1 << 30))) + rng.uniform(-500, 500, size=3)
    elif kind == 5:
        pts = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1e-3, n), np.zeros(n)], 1)     # nearly a line
        pts[rng.choice(n, max(1, n // 500), replace=False)] += rng.normal(size=3) * 50        # outliers
    elif kind == 6:                                                                         # exact plane, jittered lattice
        m = int(np.sqrt(n)) + 1
        gx, gy = np.meshgrid(np.arange(m), np.arange(m))
        pts = np.stack([gx.ravel()[:n], gy.ravel()[:n], np.zeros(n)], 1) / m + rng.normal(scale=rng.choice([0, 1e-4, 1e-2]) / m, size=(n, 3)) * [1, 1, 0]
    elif kind == 7:                                                                         # far from the origin: float32 quantisation
        pts = shapes.torus_random(n, seed=int(rng.integers(1 << 30))) * 10.0 ** rng.uniform(-2, 1) + 10.0 ** rng.uniform(2, 5)
    elif kind == 8:                                                                         # two clusters far apart
        half = n // 2
        pts = np.vstack([rng.normal(size=(half, 3)) * [1, 1, 0.05], rng.normal(size=(n - half, 3)) * [0.3, 0.3, 0.01] + rng.uniform(20, 2000)])
    elif kind == 9:                                                                         # very anisotropic box
        pts = rng.uniform(0, 1, size=(n, 3)) * [1, 1e-3, 1e-6]
    else:                                                                                   # extreme magnitudes (float32 squares overflow / underflow)
        pts = shapes.torus_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64) * 10.0 ** rng.choice([-30, -22, -15, 12, 18, 25, 30])
    pts = np.ascontiguousarray(pts, dtype=np.float64 if rng.random() < F64_SHARE else np.float32)
    eps = 0.0
    if rng.random() < 0.3:
        ext = float(np.ptp(pts, axis=0).max())
        eps = ext * 10.0 ** rng.uniform(-2.5, -0.5)
    return rng, pts, n, k, kind, eps


def run(seed0, budget=None, cases=None, verbose=True):
  """Runs random cases until `budget` seconds or `cases` cases are done; returns (cases, description of the first mismatch or None)."""
  t_end = time.time() + (budget if budget is not None else 1e9)
  it = 0
  while time.time() < t_end and (cases is None or it < cases):
      rng, pts, n, k, kind, eps = make_case(seed0, it)
      if verbose: print(f"case {it}: n={n} k={k} kind={kind} eps={eps:.4g} dtype={pts.dtype}", flush=True)
      t_case = time.time()
      h = _capi.Handle(0)
      h.set_points(pts)
      h.curvature(k, eps, _capi.KNN_BRUTE)
      t_brute = time.time() - t_case
      ib, db, cb = h.get_neighbors(0, n, want_count=True)
      cfb, Kb, Hb, _ = h.get_fit(0, n)
      for algo, name in ((_capi.KNN_GRID, "grid"), (_capi.KNN_GRID_LEVELS, "levels"), (_capi.KNN_TREE, "tree")):
          if name == "levels" and rng.random() < 0.5:
              continue
          h.curvature(k, eps, algo)
          ig, dg, cg = h.get_neighbors(0, n, want_count=True)
          cfg, Kg, Hg, _ = h.get_fit(0, n)
          ok = (np.array_equal(ib, ig) and np.array_equal(db, dg) and np.array_equal(cb, cg) and np.array_equal(cfb, cfg, equal_nan=True)
                and np.array_equal(Kb, Kg, equal_nan=True) and np.array_equal(Hb, Hg, equal_nan=True))
          if not ok:
              return it, f"algo={name} seed=({seed0},{it}) n={n} k={k} kind={kind} eps={eps} dtype={pts.dtype}"
      lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
      h.set_query_range(lo, hi)
      h.knn(k, eps=eps, algo=_capi.KNN_GRID)
      i2, d2, c2 = h.get_neighbors(lo, hi, want_count=True)
      if not (np.array_equal(i2, ib[lo:hi]) and np.array_equal(d2, db[lo:hi]) and np.array_equal(c2, cb[lo:hi])):
          return it, f"shard {lo}:{hi} seed=({seed0},{it}) n={n} k={k} kind={kind} eps={eps} dtype={pts.dtype}"
      h.close()
      if verbose: print(f"   {time.time() - t_case:.2f} s (brute {t_brute:.2f} s)", flush=True)
      it += 1
  return it, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
    if bad:
        print("MISMATCH", bad, flush=True)
        sys.exit(1)
    print(f"done: {n_done} cases, no mismatch")
