"""Handle lifetime check (developer tool): create / use / destroy many handles; device memory must come back.
A leak of the per-handle buffers (about 1.4 GB at 2 M points, k=50) would exhaust the card long before the loop ends."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def free_bytes():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value, t.value
pts = shapes.torus_random(2_000_000, seed=3)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 250
h = _capi.Handle(0); h.set_points(pts); h.curvature(50, 0.0, _capi.KNN_GRID); h.close()
f0, total = free_bytes()
t0 = time.time()
for i in range(iters):
    h = _capi.Handle(0)
    h.set_points(pts if i % 3 else pts[: 700_000 + 1000 * i])
    h.set_query_range(0, 500_000 if i % 2 else h.n)
    h.curvature(50, 0.004 if i % 5 == 0 else 0.0, _capi.KNN_GRID if i % 4 else _capi.KNN_GRID_LEVELS)
    h.get_fit(0, 1000); h.get_neighbors(0, 1000)
    if i % 7 == 0: h.query_points(pts[:5].astype(np.float64), 10)
    if i % 11 == 0: h.voxel_downsample(pts[:100_000], 0.05)
    h.close()
    if i % 50 == 0:
        f, _ = free_bytes(); print(f"iter {i}: free {f / 2**30:.2f} GiB (start {f0 / 2**30:.2f}), {time.time() - t0:.0f} s", flush=True)
f1, _ = free_bytes()
print(f"free before {f0 / 2**30:.3f} GiB, after {f1 / 2**30:.3f} GiB of {total / 2**30:.0f}; drift {(f0 - f1) / 2**20:.1f} MiB over {iters} handles")
assert f0 - f1 < 512 * 2**20, "device memory did not come back"
print("leak probe ok")
