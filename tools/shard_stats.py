"""Sweep statistics of one rank of a G-way sharded scan-ordered torus (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
from point_cloud_toolbox_amd.dist import shard_range
per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for G in (2, 8):
    n = per * G
    pts = np.concatenate([shapes.torus_scan_order(n, G, r, seed=1234) for r in range(G)])
    for rank in (0, G - 1):
        h = _capi.Handle(0)
        h.set_points(pts)
        h.set_stats(True)
        lo, hi = shard_range(n, rank, G)
        h.set_query_range(lo, hi)
        for _ in range(2):
            h.curvature(50, 0.0, _capi.KNN_GRID)
        print(G, rank, h.timings(), flush=True)
        h.close()
