"""Sweep the cell-occupancy factor of the grid sweep on the bench cloud (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pts = shapes.torus_random(n, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
stats = os.environ.get('PCT_STATS', '0') == '1'
h.set_stats(stats)
for f in [float(x) for x in (sys.argv[3:] or "0.2 0.25 0.3 0.35 0.4 0.45 0.55".split())]:
    h.set_grid_param(f)
    best = None
    for _ in range(4):
        h.curvature(k, 0.0, _capi.KNN_GRID)
        t = h.timings()
        if best is None or t["knn_ms"] < best["knn_ms"]:
            best = t
    print(f"factor {f:.2f}: knn {best['knn_ms']:.3f} ms grid {best['grid_ms']:.3f} fit {best['fit_ms']:.3f} total {best['total_ms']:.3f} | cell {best['cell_size']:.4f} occ {best['occupied_cells']} "
          f"fallback {best['ring_fallbacks']} ovf {best['lds_overflows']} flush/q {best['flushes']/n:.2f} steps/q {best['candidate_steps']/n:.2f} iters {best['grid_iters']} m {best['occupancy']:.1f} redo {best['redone_queries']}")
