"""The hierarchical cell list on the C5-shaped cloud (bunny x 557 = 20 M points, k=80, eps hybrid) and on an 8 M-point 1/r^2
scan: sampled rows and curvatures against the uniform cell list, bit for bit (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes

def check(name, pts, k, eps):
    n = len(pts)
    rows = np.sort(np.random.default_rng(3).choice(n, 20000, replace=False))
    h = _capi.Handle(0)
    h.set_points(pts)
    out = {}
    for algo, tag in ((_capi.KNN_TREE, "tree"), (_capi.KNN_GRID, "grid")):
        best = None
        for _ in range(2):
            h.curvature(k, eps, algo)
            t = h.timings()
            if best is None or t["total_ms"] < best["total_ms"]:
                best = t
        i, d, c = h.get_neighbor_rows(rows)
        _, K, H, _ = h.get_fit(0, n, coefs=False, H2=False)
        out[tag] = (i, d, c, K, H, best)
    a, b = out["tree"], out["grid"]
    same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[:5], b[:5]))
    print(f"{name}: n={n} k={k} eps={eps}: tree {a[5]['total_ms']:.2f} ms (build {a[5]['grid_ms']:.2f} knn {a[5]['knn_ms']:.2f} fit {a[5]['fit_ms']:.2f}, algo {a[5]['algo']}) | "
          f"grid {b[5]['total_ms']:.2f} ms | identical {same}", flush=True)
    h.close()
    return same

ok = True
bunny = np.load(os.path.join(ROOT, "tests", "golden", "bunny_xyz_f32.npy"))
ok &= check("C5 tiled bunny", shapes.tile_cloud(bunny, 557), 80, 0.0062)
rng = np.random.default_rng(9)
n = 8_000_000
r, a = 0.01 * 100 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n)
x, y = r * np.cos(a), r * np.sin(a)
ok &= check("1/r^2 scan", np.ascontiguousarray(np.stack([x, y, 0.05 * np.sin(x) * np.cos(y)], 1), dtype=np.float32), 50, 0.0)
sys.exit(0 if ok else 1)
