"""Sweep cost on clouds of uneven density (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(5)
def plane(n, f):
    xy = f(n)
    return np.stack([xy[:, 0], xy[:, 1], 0.05 * np.sin(xy[:, 0]) * np.cos(xy[:, 1])], 1)
clouds = {
    "uniform plane": plane(n, lambda n: rng.uniform(-1, 1, (n, 2))),
    "two densities 10:1": plane(n, lambda n: np.vstack([rng.uniform(-1, 0, (n * 10 // 11, 2)), rng.uniform(0, 1, (n - n * 10 // 11, 2))])),
    "two densities 100:1": plane(n, lambda n: np.vstack([rng.uniform(-1, 0, (n * 100 // 101, 2)), rng.uniform(0, 1, (n - n * 100 // 101, 2))])),
    "lidar 1/r": plane(n, lambda n: (lambda r, a: np.stack([r * np.cos(a), r * np.sin(a)], 1))(rng.uniform(0.01, 1, n), rng.uniform(0, 2 * np.pi, n))),
    "lidar 1/r^2": plane(n, lambda n: (lambda r, a: np.stack([r * np.cos(a), r * np.sin(a)], 1))(0.01 * 100 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n))),
    "gaussian blob 3d": rng.normal(size=(n, 3)),
}
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, p in clouds.items():
    if only and only not in name:
        continue
    p = np.ascontiguousarray(p, dtype=np.float32)
    h = _capi.Handle(0)
    h.set_points(p)
    h.set_stats(True)
    h.curvature(50, 0.0, _capi.KNN_GRID)
    s = h.timings()
    h.set_stats(False)
    best = None
    for _ in range(3):
        h.curvature(50, 0.0, _capi.KNN_GRID)
        t = h.timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    lv = None
    for _ in range(3):
        h.curvature(50, 0.0, _capi.KNN_GRID_LEVELS)
        t = h.timings()
        if lv is None or t["total_ms"] < lv["total_ms"]:
            lv = t
    h.set_stats(True)
    h.curvature(50, 0.0, _capi.KNN_TREE)
    ts = h.timings()
    h.set_stats(False)
    tr = None
    for _ in range(3):
        h.curvature(50, 0.0, _capi.KNN_TREE)
        t = h.timings()
        if tr is None or t["total_ms"] < tr["total_ms"]:
            tr = t
    if os.environ.get("PCT_PROBE_CHECK"):              # same bits as the plain sweep? (both are exact)
        _, Kt, Ht, _ = h.get_fit(0, len(p), coefs=False, H2=False)
        it, dt, _ = h.get_neighbors(0, min(len(p), 200000))
        h.curvature(50, 0.0, _capi.KNN_GRID)
        _, Kg, Hg, _ = h.get_fit(0, len(p), coefs=False, H2=False)
        ig, dg, _ = h.get_neighbors(0, min(len(p), 200000))
        print(f"{name:22s} TREE == GRID: idx {np.array_equal(it, ig)} dist {np.array_equal(dt, dg)} K {np.array_equal(Kt, Kg, equal_nan=True)} H {np.array_equal(Ht, Hg, equal_nan=True)}")
    print(f"{name:22s} TREE   total {tr['total_ms']:8.3f} ms knn {tr['knn_ms']:.3f} (fast {tr['knn_fast_ms']:.3f}) fit {tr['fit_ms']:.3f} build {tr['grid_ms']:.3f} | items {tr['occupied_cells']} segments {tr['cells']} redo {ts['redone_queries']} ovf-items {ts['lds_overflows']} up-level {ts['ring_fallbacks']}")
    au = None
    for _ in range(3):
        h.curvature(50, 0.0, _capi.KNN_AUTO)
        t = h.timings()
        if au is None or t["total_ms"] < au["total_ms"]:
            au = t
    print(f"{name:22s} AUTO   total {au['total_ms']:8.3f} ms (levels {au['levels']})")
    print(f"{name:22s} LEVELS total {lv['total_ms']:8.3f} ms knn {lv['knn_ms']:.3f} fit {lv['fit_ms']:.3f} levels {lv['levels']}")
    print(f"{name:22s} total {best['total_ms']:8.3f} ms  grid {best['grid_ms']:.3f} knn {best['knn_ms']:.3f} (fast {best['knn_fast_ms']:.3f}) fit {best['fit_ms']:.3f} | "
          f"iters {best['grid_iters']} m {best['occupancy']:.1f} cells {best['cells']} redo {s['redone_queries']} ovf-items {s['lds_overflows']} ring>1 {s['ring_fallbacks']}", flush=True)
    h.close()
