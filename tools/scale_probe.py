"""Scale check beyond 2^31 table entries: N-million-point torus, k=50; sampled rows against a host brute force
(developer tool).  python tools/scale_probe.py [millions]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 60_000_000
k = 50
t0 = time.time(); pts = shapes.torus_random(n, seed=77); print(f"generated {n} points in {time.time()-t0:.1f} s", flush=True)
h = _capi.Handle(0)
h.set_points(pts)
h.curvature(k, 0.0, _capi.KNN_GRID)
print("cold call:", {kk: round(v, 1) for kk, v in h.timings().items() if kk in ("grid_ms", "knn_ms", "fit_ms", "total_ms")}, flush=True)
h.curvature(k, 0.0, _capi.KNN_GRID)
t = h.timings()
print({kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in t.items() if kk in ("grid_ms", "knn_ms", "fit_ms", "total_ms", "cells", "occupied_cells", "occupancy")}, flush=True)
rng = np.random.default_rng(1)
rows = np.sort(rng.choice(n, 64, replace=False)).astype(np.int64)
rows[0], rows[-1] = 0, n - 1
idx, dist, _ = h.get_neighbor_rows(rows)
_, K, H, _ = h.get_fit(n - 1000, n)
assert np.isfinite(K).all() and np.isfinite(H).all()
p64 = pts.astype(np.float64)
bad = 0
for j, r in enumerate(rows):
    d2 = ((p64 - p64[r]) ** 2)
    d2 = (d2[:, 0] + d2[:, 1]) + d2[:, 2]
    order = np.argpartition(d2, k + 1)[: k + 1]
    order = order[np.lexsort((order, d2[order]))][1:]
    if not (np.array_equal(order.astype(np.int32), idx[j]) and np.array_equal(np.sqrt(d2[order]).astype(np.float32), dist[j])):
        bad += 1
        print("row", r, "differs", idx[j][:5], order[:5])
print(f"{len(rows)} sampled rows checked, {bad} differ", flush=True)
h.close()
