"""The hierarchical cell list on a 1/r^2 scan, n calls of pct_curvature (for the profiler: no statistics pass).
python tools/tree_probe.py [points] [calls] [k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(5)
r, a = 0.01 * 100 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n)
x, y = r * np.cos(a), r * np.sin(a)
p = np.ascontiguousarray(np.stack([x, y, 0.05 * np.sin(x) * np.cos(y)], 1), dtype=np.float32)
h = _capi.Handle(0)
h.set_points(p)
tot = 0.0
for i in range(calls):
    h.curvature(k, 0.0, _capi.KNN_TREE)
    tot += h.timings()["total_ms"]
t = h.timings()
print(f"1/r^2 scan, {n} points, k={k}: {tot / calls:.3f} ms per call (last: build {t['grid_ms']:.3f} knn {t['knn_ms']:.3f} fast {t['knn_fast_ms']:.3f} fit {t['fit_ms']:.3f})")
h.close()
