import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
n = 1_000_000
rng = np.random.default_rng(5)
ratio = int(sys.argv[1]) if len(sys.argv) > 1 else 10
xy = np.vstack([rng.uniform(-1, 0, (n * ratio // (ratio + 1), 2)), rng.uniform(0, 1, (n - n * ratio // (ratio + 1), 2))])
if len(sys.argv) > 2 and sys.argv[2] == "shuffle": xy = xy[rng.permutation(n)]
p = np.ascontiguousarray(np.stack([xy[:, 0], xy[:, 1], 0.05 * np.sin(xy[:, 0]) * np.cos(xy[:, 1])], 1), dtype=np.float32)
h = _capi.Handle(0); h.set_points(p)
for stats in (True, False):
    h.set_stats(stats)
    for _ in range(3): h.curvature(50, 0.0, _capi.KNN_GRID)
    t = h.timings()
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items() if k in ("grid_ms", "knn_ms", "knn_fast_ms", "fit_ms", "occupied_cells", "redone_queries", "candidate_steps", "flushes", "ring_fallbacks", "lds_overflows", "cells")}, flush=True)
