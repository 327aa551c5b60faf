"""Per-step wall time of the bench step right after the warm-up (developer tool): is the first timed region slower, and why?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
pts = shapes.torus_random(1_000_000, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
for _ in range(3):
    h.curvature(50, 0.0, _capi.KNN_GRID)
h.synchronize()
rows = []
for i in range(80):
    t0 = time.perf_counter()
    h.curvature(50, 0.0, _capi.KNN_GRID)
    dt = time.perf_counter() - t0
    tm = h.timings()
    rows.append((1e3 * dt, tm["grid_ms"], tm["knn_fast_ms"], tm["knn_ms"], tm["fit_ms"], tm["grid_iters"]))
for i in range(0, 80, 4):
    print(i, " ".join(f"{r[0]:.3f}" for r in rows[i:i + 4]), "| device grid/fast/knn/fit of first:", " ".join(f"{v:.3f}" for v in rows[i][1:5]), "iters", rows[i][5])
