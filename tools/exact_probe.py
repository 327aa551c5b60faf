import os, sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes
pts = shapes.torus_random(1_000_000, seed=1234)
h = _capi.Handle(0)
h.set_points(pts)
h.set_stats(True)
for algo, name in ((3, "exact-all"), (2, "fast+redo")):
    for _ in range(2):
        h.knn(50, 0.0, algo)
    t = h.timings()
    print(name, {k: t[k] for k in ("knn_ms", "ring_fallbacks", "lds_overflows", "redone_queries", "flushes", "candidate_steps")})
h.set_query_range(0, 2000)
for _ in range(2):
    h.knn(50, 0.0, 3)
print("exact 2000 queries only:", h.timings()["knn_ms"])
h.set_query_range(0, 20)
for _ in range(2):
    h.knn(50, 0.0, 3)
print("exact 20 queries only:", h.timings()["knn_ms"])
