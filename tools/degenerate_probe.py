"""GPU fit against the reference goldens of ill-conditioned neighbourhoods, row classes by singular values (developer tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
import pct_oracle as oracle

G = os.path.join(ROOT, "tests", "golden")


def design_sigmas(P, i, nb):
    rot = oracle.plane_align(P[nb] - P[i])
    p = np.array(rot, dtype=np.float32)
    a, b = p[:, 0], p[:, 1]
    X = np.column_stack((a ** 2, b ** 2, a * b, a, b, np.ones_like(a))).astype(np.float32).astype(np.float64)
    return np.linalg.svd(X, compute_uv=False)


h = _capi.Handle(0)
for name in sorted(f for f in os.listdir(G) if f.startswith("g10_")):
    g = dict(np.load(os.path.join(G, name)))
    P, k = g["points"], int(g["k"])
    h.set_points(P)
    h.knn(k, 0.0, _capi.KNN_GRID)
    idx, dist, _ = h.get_neighbors(0, len(P))
    same = (idx == g["idx"]).all(1)
    h.fit_indices(g["idx"])
    t = h.timings()
    co, K, H, _ = h.get_fit(0, len(P))
    sK, sH = np.nanmax(np.abs(g["K"])), np.nanmax(np.abs(g["H"]))
    rng = np.random.default_rng(0)
    rows = rng.choice(len(P), 1500, replace=False)
    sig = np.array([design_sigmas(P, i, g["idx"][i]) for i in rows])
    rc = np.finfo(np.float64).eps * k
    rel = sig / sig[:, :1]
    band = ((rel > rc / 100) & (rel < rc * 100)).any(1)                  # a singular value near the cut-off
    kept = np.where(rel > rc, rel, np.inf).min(1)                         # smallest kept singular value (relative)
    errK = np.abs(K[rows] - g["K"][rows]) / np.maximum(np.abs(g["K"][rows]), 1e-300)
    errH = np.abs(H[rows] - g["H"][rows]) / np.maximum(np.abs(g["H"][rows]), 1e-300)
    absK, absH = np.abs(K[rows] - g["K"][rows]), np.abs(H[rows] - g["H"][rows])
    nanmis = (np.isnan(K[rows]) != np.isnan(g["K"][rows])).sum()
    print(f"{name}: dists equal {np.array_equal(dist, g['dists'])}, idx rows equal {same.mean():.3f}, svd rows {t['fit_svd_rows']} / {len(P)}, "
          f"|K| max {sK:.3g} |H| max {sH:.3g}, nan mismatch {nanmis}, band rows {band.sum()}")
    for lo, hi in [(0, 1e-12), (1e-12, 1e-10), (1e-10, 1e-8), (1e-8, 1e-6), (1e-6, 1e-4), (1e-4, 1)]:
        m = (kept >= lo) & (kept < hi) & ~band
        if m.sum():
            print(f"   kept sigma_min/sigma_1 in [{lo:.0e},{hi:.0e}) n={m.sum():5d}: rel err K med {np.median(errK[m]):.1e} max {errK[m].max():.1e} | H med {np.median(errH[m]):.1e} max {errH[m].max():.1e}"
                  f" | abs K max {absK[m].max():.2e} abs H max {absH[m].max():.2e}")
    if band.sum():
        print(f"   band rows: rel err K med {np.median(errK[band]):.1e} max {errK[band].max():.1e}; H med {np.median(errH[band]):.1e} max {errH[band].max():.1e}")

g = dict(np.load(os.path.join(G, "g6b_degenerate_unit_cases.npz")))
for n in sorted(k[:-3] for k in g if k.endswith("_in")):
    nb = g[n + "_in"]
    cloud = np.vstack([np.zeros((1, 3), nb.dtype), nb])
    h.set_points(cloud)
    h.fit_indices(np.arange(1, len(cloud), dtype=np.int32)[None, :], query=np.array([0]))
    co, K, H, H2 = h.get_fit(0, 1)
    print(f"{n:22s} svd {h.timings()['fit_svd_rows']} coefs gpu {co[0]} ref {g[n + '_coefs']}  K {K[0]:.6g}/{g[n + '_curv'][0]:.6g} H {H[0]:.6g}/{g[n + '_curv'][1]:.6g}")
h.close()
