"""Developer tool: where an end-to-end step (upload, fused call, K/H download) spends its wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts = shapes.torus_random(n, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
for _ in range(5): h.curvature(50, 0.0, _capi.KNN_GRID)
acc = np.zeros(4)
reps = 20
for _ in range(reps):
    t0 = time.perf_counter(); h.set_points(pts)
    t1 = time.perf_counter(); h.curvature(50, 0.0, _capi.KNN_GRID)
    t2 = time.perf_counter(); h.get_fit(0, n, coefs=False, H2=False)
    t3 = time.perf_counter()
    tm = h.timings()
    acc += [t1 - t0, t2 - t1, t3 - t2, tm["total_ms"] * 1e-3]
acc *= 1e3 / reps
print(f"upload {acc[0]:.3f} ms | fused call {acc[1]:.3f} ms (device events {acc[3]:.3f}, grid passes {tm['grid_iters']}) | download {acc[2]:.3f} ms | sum {acc[:3].sum():.3f}")
h.synchronize(); t0 = time.perf_counter()
for _ in range(reps): h.curvature(50, 0.0, _capi.KNN_GRID)
h.synchronize(); print(f"resident step {1e3*(time.perf_counter()-t0)/reps:.3f} ms")
