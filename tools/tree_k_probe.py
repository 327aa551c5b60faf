import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
rng = np.random.default_rng(5)
n = 1_000_000
r = 10.0 ** rng.uniform(-2.0, 0.0, size=n); th = rng.uniform(0, 2 * np.pi, size=n)
scan = np.stack([r * np.cos(th), r * np.sin(th), 0.02 * np.sin(8 * r)], 1).astype(np.float32)
h = _capi.Handle(0)
h.set_points(scan)
for k in (50, 80, 100):
    best = None
    for _ in range(4):
        h.curvature(k, 0.0, _capi.KNN_TREE)
        t = h.timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    print("1/r^2 scan, tree, k", k, {a: round(best[a], 3) for a in ("grid_ms", "knn_ms", "knn_fast_ms", "fit_ms", "total_ms")}, "redone", best["redone_queries"], flush=True)
