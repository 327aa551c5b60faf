"""Rows on which the tree sweep and the plain grid sweep disagree (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
which = sys.argv[2] if len(sys.argv) > 2 else "1/r"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(5)
def plane(xy):
    return np.stack([xy[:, 0], xy[:, 1], 0.05 * np.sin(xy[:, 0]) * np.cos(xy[:, 1])], 1)
# same draws as density_probe.py up to the cloud asked for
u = rng.uniform(-1, 1, (n, 2))
a = np.vstack([rng.uniform(-1, 0, (n * 10 // 11, 2)), rng.uniform(0, 1, (n - n * 10 // 11, 2))])
b = np.vstack([rng.uniform(-1, 0, (n * 100 // 101, 2)), rng.uniform(0, 1, (n - n * 100 // 101, 2))])
r1, a1 = rng.uniform(0.01, 1, n), rng.uniform(0, 2 * np.pi, n)
r2, a2 = 0.01 * 100 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n)
clouds = {"uniform": plane(u), "10:1": plane(a), "100:1": plane(b), "1/r": plane(np.stack([r1 * np.cos(a1), r1 * np.sin(a1)], 1)),
          "1/r^2": plane(np.stack([r2 * np.cos(a2), r2 * np.sin(a2)], 1))}
p = np.ascontiguousarray(clouds[which], dtype=np.float32)
h = _capi.Handle(0)
h.set_points(p)
h.set_stats(True)
h.knn(k, 0.0, _capi.KNN_TREE)
t = h.timings()
it, dt, _ = h.get_neighbors(0, n)
h.knn(k, 0.0, _capi.KNN_GRID)
ig, dg, _ = h.get_neighbors(0, n)
if os.environ.get("PCT_DEBUG_NO_BRUTE"):
    ib, db = ig, dg
else:
    h.knn(k, 0.0, _capi.KNN_BRUTE)
    ib, db, _ = h.get_neighbors(0, n)
print("tree == brute:", np.array_equal(it, ib) and np.array_equal(dt, db), "| grid == brute:", np.array_equal(ig, ib) and np.array_equal(dg, db))
bad = np.nonzero((it != ig).any(1) | (dt != dg).any(1))[0]
print(f"{which} n {n} k {k}: tree redone {t['redone_queries']} overflow-items {t['lds_overflows']} up-level {t['ring_fallbacks']} | {len(bad)} rows differ")
for q in bad[:8]:
    cols = np.nonzero((it[q] != ig[q]) | (dt[q] != dg[q]))[0]
    print(f"  row {q} r={np.hypot(p[q,0], p[q,1]):.4f}: first differing column {cols[0]} of {len(cols)}; tree idx {it[q, cols[0]]} d {dt[q, cols[0]]:.6g} | grid idx {ig[q, cols[0]]} d {dg[q, cols[0]]:.6g}; "
          f"true d(tree idx) {np.sqrt(((p[q].astype(np.float64) - p[it[q, cols[0]]].astype(np.float64)) ** 2).sum()):.6g}; dup of query coords {int((p == p[q]).all(1).sum())}; "
          f"same set {set(it[q]) == set(ig[q])}; tree row has {len(set(it[q]))} distinct; tree==brute {np.array_equal(it[q], ib[q])} grid==brute {np.array_equal(ig[q], ib[q])}; tree last d {dt[q,-1]:.6g} grid last d {dg[q,-1]:.6g}")
if len(bad):
    rr = np.hypot(p[bad, 0], p[bad, 1])
    print("  radii of differing rows: min %.4g median %.4g max %.4g" % (rr.min(), np.median(rr), rr.max()))
