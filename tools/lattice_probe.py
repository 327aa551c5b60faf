"""Stage times and redo fraction on the reference's own lattice inputs next to the random torus (developer tool).

    python tools/lattice_probe.py [n_side] [k] [substring of the cloud's name]
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
clouds = {
    "torus random": shapes.torus_random(n_side * n_side, seed=1234),
    "torus grid (utils.py:883)": shapes.torus_grid(n_side),
    "egg carton grid (utils.py:906)": shapes.egg_carton_grid(n_side),
}
eggf = os.path.join(ROOT, "tests", "golden", "g9_eggcarton_file_k30_sample.npz")
if os.path.exists(eggf):
    clouds["egg_carton.txt (file ctor)"] = np.load(eggf)["points"]
only = sys.argv[3] if len(sys.argv) > 3 else None
out = {}
for name, p in clouds.items():
    if only and only not in name:
        continue
    kk = 30 if "txt" in name else k
    p = np.ascontiguousarray(p, dtype=np.float32)
    h = _capi.Handle(0)
    h.set_points(p)
    if os.environ.get("PCT_PROBE_NO_STATS"):                 # under the profiler: no statistics pass (its atomics would
        h.curvature(kk, 0.0, _capi.KNN_GRID)                 # sit in the kernel averages)
        s = h.timings()
    else:
        h.set_stats(True)
        h.curvature(kk, 0.0, _capi.KNN_GRID)
        s = h.timings()
        h.set_stats(False)
    best = None
    for _ in range(5):
        h.curvature(kk, 0.0, _capi.KNN_GRID)
        t = h.timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    n = len(p)
    out[name] = dict(n=n, k=kk, total_ms=best["total_ms"], grid_ms=best["grid_ms"], knn_ms=best["knn_ms"],
                     knn_fast_ms=best["knn_fast_ms"], fit_ms=best["fit_ms"], redone=s["redone_queries"],
                     redo_fraction=s["redone_queries"] / n, lds_overflows=s["lds_overflows"],
                     ring_fallbacks=s["ring_fallbacks"], occupancy=best["occupancy"], mpts_per_s=n / best["total_ms"] / 1e3)
    print(f"{name:32s} n {n:8d} k {kk:3d} total {best['total_ms']:8.3f} ms ({n / best['total_ms'] / 1e3:7.1f} M pts/s)  grid {best['grid_ms']:.3f} "
          f"knn {best['knn_ms']:.3f} (fast {best['knn_fast_ms']:.3f}) fit {best['fit_ms']:.3f} | redo {s['redone_queries']} "
          f"({100.0 * s['redone_queries'] / n:.2f} %) ovf-items {s['lds_overflows']} ring>1 {s['ring_fallbacks']} occ {best['occupancy']:.1f}", flush=True)
    h.close()
print(json.dumps(out))
