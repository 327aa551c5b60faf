"""Which Cholesky pivot ratio still lets normal equations follow numpy.linalg.lstsq (CPU study; TEST INFRASTRUCTURE).

k_fit (csrc/pct_fit.hip) solves the quadric through normal equations and hands a row to the SVD kernel when its
smallest pivot ratio d_j / g_jj falls below kPivotRatioMin.  This script measures, on anisotropic scan-line lattices
(aspect 1 : 1 ... 1 : 30, wavy surface and jittered cylinder), the error of a float64 Cholesky solve of the reference's
design matrix against lstsq(rcond=None), binned by that ratio:

    ratio >= 1e-9            <= 2.4e-7 relative in K, H   (float32 rounding noise)
    [1e-10, 1e-9)            up to 1.3e-5
    below                    up to 1e-2 and worse

kPivotRatioMin = 1e-6 leaves three decades.  Run:  python oracle/calibrate_pivot_ratio.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pct_oracle as o                                   # noqa: E402
from make_goldens_degenerate import scanline_cloud       # noqa: E402  (no reference import happens there at module level)


def design(points, i, nb):
    p = np.array(o.plane_align(points[nb] - points[i]), dtype=np.float32)
    a, b, c = p[:, 0], p[:, 1], p[:, 2]
    return np.column_stack((a ** 2, b ** 2, a * b, a, b, np.ones_like(a))).astype(np.float32), c


def chol_solve(X, z):
    X, z = X.astype(np.float64), z.astype(np.float64)
    G, b = X.T @ X, X.T @ z
    L, rmin = np.zeros((6, 6)), 1.0
    for j in range(6):
        d = G[j, j] - (L[j, :j] ** 2).sum()
        rmin = min(rmin, d / G[j, j]) if G[j, j] > 0 else 0.0
        if d <= 0:
            return np.full(6, np.nan), rmin
        L[j, j] = np.sqrt(d)
        for i in range(j + 1, 6):
            L[i, j] = (G[i, j] - (L[i, :j] * L[j, :j]).sum()) / L[j, j]
    return np.linalg.solve(L.T, np.linalg.solve(L, b)), rmin


def main():
    rng = np.random.default_rng(0)
    res = []
    for aspect in (1, 2, 4, 8, 12, 16, 20, 30):
        for surf, jit in (("wavy", 0.0), ("cylinder", 2e-4)):
            P = scanline_cloud(60, 400, 0.005, 0.005 * aspect, surf, jitter=jit)
            rows = rng.choice(len(P), 150, replace=False)
            idx, _ = o.knn(P, 30, query_rows=rows)
            for r, i in enumerate(rows):
                X, z = design(P, i, idx[r])
                ref = np.linalg.lstsq(X, z, rcond=None)[0]
                c, rmin = chol_solve(X, z)
                Kr, Hr = o.quadric_curvatures(ref)[:2]
                Kn, Hn = o.quadric_curvatures(c.astype(np.float32))[:2]
                e = max(abs(Kn - Kr) / max(abs(Kr), 1e-3), abs(Hn - Hr) / max(abs(Hr), 1e-3)) if np.isfinite(Kn) else np.inf
                res.append((rmin, e))
    res = np.array(res)
    for lo in (1e-14, 1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1):
        m = (res[:, 0] >= lo) & (res[:, 0] < lo * 10)
        if m.sum():
            print(f"pivot ratio in [{lo:.0e},{lo * 10:.0e}): n={m.sum():4d}  error median {np.median(res[m, 1]):.2e}  max {np.max(res[m, 1]):.2e}")


if __name__ == "__main__":
    main()
