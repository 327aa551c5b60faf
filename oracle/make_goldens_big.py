"""Sampled reference goldens for the big BASELINE configs (build container only; TEST INFRASTRUCTURE).

C4: egg carton 5 M points, k=50.   C5: sample_scans/bunny.txt tiled x557 (20 022 479 points), k=80, hybrid
eps=0.0062.  The reference's own staticmethods are evaluated on 2000 fixed sample rows each
(cKDTree built exactly as pointCloudToolbox.py:74).  Also stores the bunny scan itself as a float32 fixture
(a data file of the reference, needed to rebuild the C5 cloud on the GPU box).
Run from the repo root:  MPLBACKEND=Agg python oracle/make_goldens_big.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, REF, load_reference, load_shapes  # noqa: E402


def sampled(ref, points, k, rows, eps=None):
    import scipy as sp
    tree = sp.spatial.cKDTree(np.array(points, dtype=np.float32))      # as pct:74
    n = len(points)
    idx = np.full((len(rows), k), n, np.int32)
    dists = np.full((len(rows), k), np.inf, np.float32)
    count = np.zeros(len(rows), np.int32)
    coefs = np.full((len(rows), 6), np.nan, np.float32)
    K = np.full(len(rows), np.nan, np.float32)
    H = np.full(len(rows), np.nan, np.float32)
    kw = {} if eps is None else {"distance_upper_bound": eps}
    for o, i in enumerate(rows):
        d, nb = tree.query(points[i], k + 1, **kw)                     # as pct:83 (+ SciPy's native eps bound)
        d, nb = d[1:], nb[1:]
        m = int(np.sum(nb < n))
        idx[o, :m], dists[o, :m], count[o] = nb[:m], d[:m], m
        if m >= 6:
            rot = ref.PointCloud.get_best_fit_plane_and_rotate(points[nb[:m]] - points[i])
            coefs[o] = ref.PointCloud.fit_quadratic_surface(rot)
            K[o], H[o] = ref.PointCloud.calculate_explicit_quadratic_curvatures(coefs[o])[:2]
    return dict(rows=np.asarray(rows, np.int64), k=np.int32(k), idx=idx, dists=dists, count=count,
                coefs=coefs, K=K, H=H, eps=np.float64(-1 if eps is None else eps))


def main():
    ref = load_reference()
    sh = load_shapes()
    rng = np.random.default_rng(4242)
    bunny = np.loadtxt(os.path.join(REF, "sample_scans", "bunny.txt"))[:, :3].astype(np.float32)
    np.save(os.path.join(OUT, "bunny_xyz_f32.npy"), bunny)
    P = sh.egg_carton_random(5_000_000, seed=1234)
    rows = np.sort(rng.choice(len(P), 2000, replace=False))
    np.savez_compressed(os.path.join(OUT, "g7_egg5m_k50_sample.npz"), **sampled(ref, P, 50, rows))
    print("C4 done", flush=True)
    P = sh.tile_cloud(bunny, 557)
    rows = np.sort(rng.choice(len(P), 2000, replace=False))
    np.savez_compressed(os.path.join(OUT, "g7_bunny20m_k80_eps_sample.npz"), **sampled(ref, P, 80, rows, eps=0.0062))
    print("C5 done", flush=True)


if __name__ == "__main__":
    main()
