import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
pts = shapes.torus_random(1_000_000, seed=1234).astype(np.float64) + 1e-9 * np.random.default_rng(1).normal(size=(1_000_000, 3))
h = _capi.Handle(0)
h.set_points(pts)
for k in (50, 64, 80, 100):
    best = None
    for _ in range(4):
        h.curvature(k, 0.0, _capi.KNN_GRID)
        t = h.timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    print("float64 cloud k", k, {a: round(best[a], 3) for a in ("grid_ms", "knn_ms", "knn_fast_ms", "fit_ms", "total_ms")}, flush=True)
