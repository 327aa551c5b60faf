"""Re-run one case of tools/fuzz_gpu.py and describe the mismatch: python tools/repro_fuzz.py seed0 it"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
from fuzz_gpu import make_case
seed0, it = int(sys.argv[1]), int(sys.argv[2])
rng, pts, n, k, kind, eps = make_case(seed0, it)
print("case", n, k, kind, eps, pts.dtype)
h = _capi.Handle(0); h.set_points(pts)
h.curvature(k, eps, _capi.KNN_BRUTE)
ib, db, cb = h.get_neighbors(0, n, want_count=True); cfb, Kb, Hb, _ = h.get_fit(0, n)
for algo, name in ((_capi.KNN_GRID, "grid"), (_capi.KNN_GRID_LEVELS, "levels"), (_capi.KNN_TREE, "tree")):
    h.curvature(k, eps, algo)
    ig, dg, cg = h.get_neighbors(0, n, want_count=True); cfg, Kg, Hg, _ = h.get_fit(0, n)
    bad_i = np.where((ib != ig).any(1))[0]; bad_d = np.where((db != dg).any(1))[0]; bad_c = np.where(cb != cg)[0]
    bad_f = np.where(~((cfb == cfg) | (np.isnan(cfb) & np.isnan(cfg))).all(1))[0]
    print(name, "levels", h.timings()["levels"], "rows with idx diff", len(bad_i), "dist diff", len(bad_d), "count diff", len(bad_c), "coef diff", len(bad_f))
    for r in list(bad_i[:3]) + list(bad_c[:2]):
        cols = np.where((ib[r] != ig[r]) | (db[r] != dg[r]))[0]
        print("  row", r, "count", cb[r], cg[r], "differing columns", cols)
        for c in cols[:6]: print("     col", c, "brute", ib[r][c], repr(db[r][c]), name, ig[r][c], repr(dg[r][c]), "| exact d to levels idx:", repr(float(np.sqrt(((pts[ig[r][c]].astype(np.float64)-pts[r].astype(np.float64))**2).sum()))) if ig[r][c] < n else None)
