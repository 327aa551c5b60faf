import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge; ge.build()
from point_cloud_toolbox_amd import _capi, shapes
for scale in (1e-30, 1e-22, 1e-15, 1e12, 1e18, 1e25, 1e30):
    for dt in (np.float32, np.float64):
        pts = (shapes.torus_random(20000, seed=5).astype(np.float64) * scale).astype(dt)
        h = _capi.Handle(0); h.set_points(pts)
        res = {}
        for name, algo in (("brute", _capi.KNN_BRUTE), ("grid", _capi.KNN_GRID)):
            try:
                h.curvature(30, 0.0, algo); res[name] = h.get_neighbors(0, 20000, want_count=True) + (h.get_fit(0, 20000)[1],)
            except Exception as e:
                res[name] = repr(e)[:80]
        if isinstance(res["brute"], str) or isinstance(res["grid"], str):
            print(scale, dt.__name__, res["brute"] if isinstance(res["brute"], str) else "brute ok", "|", res["grid"] if isinstance(res["grid"], str) else "grid ok")
        else:
            b, g = res["brute"], res["grid"]
            print(scale, dt.__name__, "idx equal", np.array_equal(b[0], g[0]), "dist equal", np.array_equal(b[1], g[1]), "K equal", np.array_equal(b[3], g[3], equal_nan=True), "redo", h.timings()["redone_queries"], "K finite", np.isfinite(g[3]).mean())
        h.close()
