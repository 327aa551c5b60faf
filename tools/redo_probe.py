"""Developer tool: who goes to the exact sweep on the bench cloud (python tools/redo_probe.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [50]
pts = shapes.torus_random(n, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
h.set_stats(True)
for kk in ks:
    for _ in range(3):
        h.curvature(kk, 0.0, _capi.KNN_GRID)
    t = h.timings()
    print(kk, {k: t[k] for k in ("redone_queries", "lds_overflows", "ring_fallbacks", "occupied_cells", "occupancy", "knn_ms", "knn_fast_ms")}, flush=True)
