"""C5-like probe (bunny tiles, k=80, eps hybrid): sweep time with the environment's kernel choice (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bunny = np.load(os.path.join(ROOT, "tests", "golden", "bunny_xyz_f32.npy"))
pts = shapes.tile_cloud(bunny, tiles)
h = _capi.Handle(0)
h.set_points(pts)
for eps in (0.0062, 0.0):
    for stats in (True, False):
        h.set_stats(stats)
        for _ in range(2):
            h.curvature(80, eps, _capi.KNN_GRID)
        t = h.timings()
        print(f"n={len(pts)} eps={eps} stats={stats}: grid {t['grid_ms']:.3f} knn {t['knn_ms']:.3f} fast {t['knn_fast_ms']:.3f} fit {t['fit_ms']:.3f} redo {t['redone_queries']} ovf {t['lds_overflows']} fallback {t['ring_fallbacks']} steps {t['candidate_steps']} flush {t['flushes']} m {t['occupancy']:.1f} cell {t['cell_size']:.5f}", flush=True)
h.close()
