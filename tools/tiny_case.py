"""One fuzz_tiny case replayed: which algorithm differs from the exhaustive sweep, and where (developer tool).
python tools/tiny_case.py seed case [algo ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
seed0, it = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed0, it])
n = int(rng.integers(2, 300)); k = int(rng.integers(1, min(127, n - 1) + 1))
kind = rng.integers(0, 4)
pts = (rng.normal(size=(n, 3)) if kind == 0 else rng.uniform(0, 1, (n, 3)) * [1, 1, 0] if kind == 1
       else np.round(rng.uniform(0, 3, (n, 3))) if kind == 2 else np.repeat(rng.normal(size=(1, 3)), n, 0) + rng.normal(size=(n, 3)) * 1e-7)
pts = np.ascontiguousarray(pts, dtype=np.float32 if rng.random() < 0.8 else np.float64)
eps = float(rng.uniform(0.05, 2)) if rng.random() < 0.3 else 0.0
if rng.random() < 0.2:
    mag = 10.0 ** rng.uniform(-30, 30); pts = (pts.astype(np.float64) * mag).astype(pts.dtype); eps *= mag
print(f"n={n} k={k} kind={kind} eps={eps} dtype={pts.dtype} distinct points {len(np.unique(pts, axis=0))}", flush=True)
h = _capi.Handle(0); h.set_points(pts)
h.curvature(k, eps, _capi.KNN_BRUTE)
ib, db, cb = h.get_neighbors(0, n, want_count=True); cfb, Kb, Hb, _ = h.get_fit(0, n)
for algo in [int(a) for a in sys.argv[3:]] or [2, 3, 5, 4]:
    print("algo", algo, flush=True)
    h.curvature(k, eps, algo)
    t = h.timings()
    ig, dg, cg = h.get_neighbors(0, n, want_count=True); cfg, Kg, Hg, _ = h.get_fit(0, n)
    bad_i = np.nonzero((ib != ig).any(1))[0]; bad_d = np.nonzero((db != dg).any(1))[0]
    bad_c = np.nonzero(~((cfb == cfg) | (np.isnan(cfb) & np.isnan(cfg))).all(1))[0]
    print(f"  rows with differing idx {len(bad_i)} dist {len(bad_d)} coefs {len(bad_c)} count {int((cb != cg).sum())} | cells {t['cells']} items {t['occupied_cells']} cell {t['cell_size']:.3g}", flush=True)
    for q in list(bad_i[:3]) + list(bad_d[:2]):
        c = np.nonzero((ib[q] != ig[q]) | (db[q] != dg[q]))[0]
        print(f"   row {q}: cols {c[:6]} brute idx {ib[q, c[:4]]} d {db[q, c[:4]]} | got idx {ig[q, c[:4]]} d {dg[q, c[:4]]}", flush=True)
h.close()
