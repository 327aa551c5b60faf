"""rank 7 of an 8-way shard of an 8M-point torus on one GPU (developer tool, run under rocprofv3 --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
G, per = 8, 1_000_000
n = per * G
pts = shapes.torus_random(n, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
h.set_query_range(n - per, n)
for _ in range(5):
    h.curvature(50, 0.0, _capi.KNN_GRID)
t = h.timings()
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
