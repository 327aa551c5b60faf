import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
seed0, it = 1, 2
rng = np.random.default_rng([seed0, it])
n = int(rng.integers(200, 60_000)); k = int(rng.integers(1, min(127, n - 1) + 1)); kind = rng.integers(0, 6)
if len(sys.argv) > 2: n = int(sys.argv[2])
pts = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1e-3, n), np.zeros(n)], 1)
pts[rng.choice(n, max(1, n // 500), replace=False)] += rng.normal(size=3) * 50
pts = np.ascontiguousarray(pts, dtype=np.float64 if rng.random() < 0.15 else np.float32)
eps = 0.0
if rng.random() < 0.3:
    ext = float(np.ptp(pts, axis=0).max()); eps = ext * 10.0 ** rng.uniform(-2.5, -0.5)
print(n, k, kind, eps, pts.dtype, np.ptp(pts, axis=0), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "grid"
h = _capi.Handle(0); h.set_points(pts); h.set_stats(os.environ.get('STATS','1')=='1')
t0 = time.time()
if mode == "grid": h.curvature(k, eps, _capi.KNN_GRID)
elif mode == "levels": h.curvature(k, eps, _capi.KNN_GRID_LEVELS)
elif mode == "grid_noeps": h.curvature(k, 0.0, _capi.KNN_GRID)
elif mode == "exact": h.curvature(k, 0.0, _capi.KNN_GRID_EXACT)
elif mode == "knn_only": h.knn(k, 0.0, _capi.KNN_GRID)
print(mode, f"{time.time()-t0:.2f} s", h.timings(), flush=True)
