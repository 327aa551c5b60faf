"""Replays tools/fuzz_tiny.py case by case, printing each case before it runs (developer tool: find the case that aborts)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 13
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 300
for it in range(first, last):
    rng = np.random.default_rng([seed0, it])
    n = int(rng.integers(2, 300)); k = int(rng.integers(1, min(127, n - 1) + 1))
    kind = rng.integers(0, 4)
    pts = (rng.normal(size=(n, 3)) if kind == 0 else rng.uniform(0, 1, (n, 3)) * [1, 1, 0] if kind == 1
           else np.round(rng.uniform(0, 3, (n, 3))) if kind == 2 else np.repeat(rng.normal(size=(1, 3)), n, 0) + rng.normal(size=(n, 3)) * 1e-7)
    pts = np.ascontiguousarray(pts, dtype=np.float32 if rng.random() < 0.8 else np.float64)
    eps = float(rng.uniform(0.05, 2)) if rng.random() < 0.3 else 0.0
    if rng.random() < 0.2:
        mag = 10.0 ** rng.uniform(-30, 30); pts = (pts.astype(np.float64) * mag).astype(pts.dtype); eps *= mag
    h = _capi.Handle(0); h.set_points(pts)
    for algo in (_capi.KNN_BRUTE, _capi.KNN_GRID, _capi.KNN_GRID_LEVELS, _capi.KNN_GRID_EXACT, _capi.KNN_TREE):
        print(f"case {it} n={n} k={k} kind={kind} eps={eps} dtype={pts.dtype} algo={algo}", flush=True)
        h.curvature(k, eps, algo)
        h.get_fit(0, n)
    h.close()
print("done")
