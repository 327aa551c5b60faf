import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
pts = shapes.torus_random(1_000_000, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
for k in (30, 50, 80, 100):
    best = None
    for _ in range(6):
        h.curvature(k, 0.0, _capi.KNN_GRID)
        t = h.timings()
        if best is None or t["knn_ms"] < best["knn_ms"]:
            best = t
    print(os.environ.get("PCT_ITEMS_Q"), k, round(best["knn_fast_ms"], 4), round(best["knn_ms"], 4), round(best["total_ms"], 4), flush=True)
