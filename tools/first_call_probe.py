"""Where the first call on a fresh handle goes (developer tool): wall time of each step against the device time inside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts = shapes.torus_random(n, seed=3)
warm = _capi.Handle(0); warm.set_points(pts[:10000]); warm.curvature(50); warm.close()      # runtime and code objects loaded
for rep in range(3):
    t0 = time.perf_counter(); h = _capi.Handle(0); t1 = time.perf_counter()
    h.set_points(pts); t2 = time.perf_counter()
    h.curvature(50, 0.0, _capi.KNN_GRID); t3 = time.perf_counter()
    tm = h.timings()
    h.curvature(50, 0.0, _capi.KNN_GRID); t4 = time.perf_counter()
    K = h.get_fit(0, n, coefs=False, H2=False); t5 = time.perf_counter()
    h.close(); t6 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.2f} ms | set_points {1e3*(t2-t1):.2f} (H2D {tm['upload_ms']:.2f}) | first curvature {1e3*(t3-t2):.2f} (device {tm['total_ms']:.2f}, grid passes {tm['grid_iters']}) | "
          f"second {1e3*(t4-t3):.2f} | get K,H {1e3*(t5-t4):.2f} | close {1e3*(t6-t5):.2f}", flush=True)
