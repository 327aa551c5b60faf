"""Re-run one case of tools/fuzz_gpu.py and describe the mismatch: python tools/repro_fuzz.py seed0 it"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
seed0, it = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed0, it])
n = int(rng.integers(200, 60_000)); k = int(rng.integers(1, min(127, n - 1) + 1)); kind = rng.integers(0, 6)
if kind == 0: pts = shapes.torus_random(n, seed=int(rng.integers(1 << 30)))
elif kind == 1: pts = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3)
elif kind == 2: pts = np.round(rng.uniform(-1, 1, size=(n, 3)) * rng.integers(3, 200)) / 16.0
elif kind == 3:
    c = rng.uniform(-1, 1, size=(8, 3)); w = rng.integers(0, 8, size=n)
    pts = c[w] + rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-4, -0.5, size=8))[w, None]
elif kind == 4: pts = shapes.egg_carton_random(n, seed=int(rng.integers(1 << 30))) + rng.uniform(-500, 500, size=3)
else:
    pts = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1e-3, n), np.zeros(n)], 1)
    pts[rng.choice(n, max(1, n // 500), replace=False)] += rng.normal(size=3) * 50
pts = np.ascontiguousarray(pts, dtype=np.float64 if rng.random() < 0.15 else np.float32)
eps = 0.0
if rng.random() < 0.3:
    ext = float(np.ptp(pts, axis=0).max()); eps = ext * 10.0 ** rng.uniform(-2.5, -0.5)
print("case", n, k, kind, eps, pts.dtype)
h = _capi.Handle(0); h.set_points(pts)
h.curvature(k, eps, _capi.KNN_BRUTE)
ib, db, cb = h.get_neighbors(0, n, want_count=True); cfb, Kb, Hb, _ = h.get_fit(0, n)
for algo, name in ((_capi.KNN_GRID, "grid"), (_capi.KNN_GRID_LEVELS, "levels")):
    h.curvature(k, eps, algo)
    ig, dg, cg = h.get_neighbors(0, n, want_count=True); cfg, Kg, Hg, _ = h.get_fit(0, n)
    bad_i = np.where((ib != ig).any(1))[0]; bad_d = np.where((db != dg).any(1))[0]; bad_c = np.where(cb != cg)[0]
    bad_f = np.where(~((cfb == cfg) | (np.isnan(cfb) & np.isnan(cfg))).all(1))[0]
    print(name, "levels", h.timings()["levels"], "rows with idx diff", len(bad_i), "dist diff", len(bad_d), "count diff", len(bad_c), "coef diff", len(bad_f))
    for r in list(bad_i[:3]) + list(bad_c[:2]):
        cols = np.where((ib[r] != ig[r]) | (db[r] != dg[r]))[0]
        print("  row", r, "count", cb[r], cg[r], "differing columns", cols)
        for c in cols[:6]: print("     col", c, "brute", ib[r][c], repr(db[r][c]), name, ig[r][c], repr(dg[r][c]), "| exact d to levels idx:", repr(float(np.sqrt(((pts[ig[r][c]].astype(np.float64)-pts[r].astype(np.float64))**2).sum()))) if ig[r][c] < n else None)
