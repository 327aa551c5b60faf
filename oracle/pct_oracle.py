"""CPU oracle for the per-point curvature path  --  TEST INFRASTRUCTURE ONLY.

This module is a CPU restatement (NumPy / SciPy) of the hot path of
``/root/reference/pointCloudToolbox.py`` (class ``PointCloud``).  It exists so
that the HIP path can be checked; nothing in the shipped package imports it.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file.

Parity status: PINNED.  ``oracle/make_goldens.py`` imports the unmodified
reference in the build container and stores its outputs (neighbour indices,
distances, quadric coefficients, K, H) under ``tests/golden/``;
``tests/test_oracle_goldens.py`` checks every function below against them.
The third-party arithmetic the reference delegates to (``scipy.spatial.cKDTree``,
``numpy.cov``, ``numpy.linalg.svd``, ``numpy.linalg.lstsq``) is not vendored by
the reference and carries no version pin there; goldens were captured under
NumPy 2.2.6 / SciPy 1.15.3.

Each function cites the reference lines (``pct:N`` = pointCloudToolbox.py:N).
Two flavours are provided:

* ``*_loop``     one Python iteration per point, same call pattern as the
                 reference (this is what ``bench.py`` times as the 1-core
                 "port" baseline);
* ``*_batched``  vectorised float64 restatement (fast checker for big inputs).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

__all__ = [
    "knn", "knn_loop", "plane_align", "quadric_fit", "quadric_curvatures",
    "curvature_loop", "curvature_batched", "pipeline_loop", "pipeline_batched",
    "neighbor_study", "curvature_tolerance_ok",
]


# --------------------------------------------------------------------------
# A3  plant_kdtree  (pct:69-89)
# --------------------------------------------------------------------------
def _tree(points):
    # pct:74 -- the tree is built on a float32-rounded copy of the cloud.
    return cKDTree(np.array(points, dtype=np.float32))


def knn(points, k, eps=None, workers=-1, query_rows=None, tree=None):
    """k nearest neighbours of every point, self dropped (pct:81-85).

    The tree holds float32-rounded coordinates (pct:74) while the query row is
    passed in the cloud's native dtype (pct:83).  ``k+1`` results are requested
    and result 0 is discarded on the assumption that it is the point itself
    (pct:84-85).  Output dtypes follow pct:78-79: float32 distances, int32
    indices, each row ascending by distance.

    ``eps`` (not in the reference code): hybrid query -- the at
    most ``k`` nearest neighbours with distance < eps, expressed through
    SciPy's ``distance_upper_bound``; missing slots carry index N / dist inf
    and ``count`` gives the number of valid entries per row.
    """
    points = np.asarray(points)
    tree = tree if tree is not None else _tree(points)
    q = points if query_rows is None else points[np.asarray(query_rows)]
    kw = {}
    if eps is not None:
        kw["distance_upper_bound"] = float(eps)
    d, i = tree.query(q, k + 1, workers=workers, **kw)
    dists = d[:, 1:].astype(np.float32)
    idx = i[:, 1:].astype(np.int32)
    if eps is None:
        return idx, dists
    count = np.sum(idx < tree.n, axis=1).astype(np.int32)
    return idx, dists, count


def knn_loop(points, k, rows=None):
    """Same as :func:`knn` but one ``query`` call per point like pct:81-85."""
    points = np.asarray(points)
    tree = _tree(points)
    rows = range(len(points)) if rows is None else rows
    dists = np.empty((len(rows), k), dtype=np.float32)
    idx = np.empty((len(rows), k), dtype=np.int32)
    for o, i in enumerate(rows):
        d, n = tree.query(points[i], k + 1)
        dists[o] = d[1:]
        idx[o] = n[1:]
    return idx, dists


# --------------------------------------------------------------------------
# A5  get_best_fit_plane_and_rotate  (pct:270-321)
# --------------------------------------------------------------------------
def plane_align(nbrs):
    """Rotate a centred neighbourhood so that its PCA normal becomes +z."""
    nbrs = np.asarray(nbrs)
    if not np.all(np.isfinite(nbrs)):                      # pct:273-274
        raise ValueError("Non-finite values in input points")
    cov = np.cov(nbrs, rowvar=False)                       # pct:277
    _, _, vt = np.linalg.svd(cov, full_matrices=True)      # pct:280
    n = vt[-1]                                             # pct:283
    ref = nbrs[-1] - nbrs[0]                               # pct:286
    n_hat = n / np.linalg.norm(n)                          # pct:289
    ref_hat = ref / np.linalg.norm(ref)                    # pct:290
    if np.dot(n_hat, ref_hat) < 0:                         # pct:293-297
        n = -n
    a = n / np.linalg.norm(n)                              # pct:301
    z = np.array([0, 0, 1])
    v = np.cross(a, z)                                     # pct:303
    c = np.dot(a, z)                                       # pct:304
    s = np.linalg.norm(v)                                  # pct:305
    if s == 0:                                             # pct:308-309
        rot = np.eye(3)
    else:                                                  # pct:311-312
        kx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        rot = np.eye(3) + kx + kx.dot(kx) * ((1 - c) / (s ** 2))
    out = np.dot(rot, nbrs.T).T                            # pct:315
    if not np.all(np.isfinite(out)):                       # pct:318-319
        raise ValueError("Non-finite values after rotation")
    return out


# --------------------------------------------------------------------------
# A6  fit_quadratic_surface  (pct:331-360)
# --------------------------------------------------------------------------
def quadric_fit(rotated):
    """Least-squares z = A a^2 + B b^2 + C ab + D a + E b + F, float32 in/out."""
    p = np.array(rotated, dtype=np.float32)                # pct:350
    if p.ndim != 2 or p.shape[1] != 3:                     # pct:351-352
        raise ValueError("Input points must have shape (N, 3)")
    a, b, c = p[:, 0], p[:, 1], p[:, 2]
    if not np.all(np.isfinite(p)):                         # pct:356-357
        raise ValueError("Input contains non-finite values.")
    X = np.column_stack((a ** 2, b ** 2, a * b, a, b, np.ones_like(a))).astype(np.float32)  # pct:358
    coefs = np.linalg.lstsq(X, c, rcond=None)[0]           # pct:359
    return coefs


# --------------------------------------------------------------------------
# A7  calculate_explicit_quadratic_curvatures  (pct:398-431)
# --------------------------------------------------------------------------
def quadric_curvatures(coefs):
    """(K, H, k1, k2, H^2) of the Monge patch at the origin, float32 scalars."""
    A, B, C, D, E, _F = coefs                              # pct:400
    Fx, Fy = D, E                                          # pct:403-404
    Fxx, Fyy, Fxy = 2 * A, 2 * B, C                        # pct:407-409
    den_k = (1 + Fx ** 2 + Fy ** 2) ** 2                   # pct:412
    den_h = (1 + Fx ** 2 + Fy ** 2) ** 1.5                 # pct:413
    K = (Fxx * Fyy - Fxy ** 2) / den_k                     # pct:416
    H = ((1 + Fx ** 2) * Fyy - 2 * Fx * Fy * Fxy + (1 + Fy ** 2) * Fxx) / (2 * den_h)  # pct:419
    H2 = H ** 2                                            # pct:422
    disc = max(H ** 2 - K, 0)                              # pct:425
    root = np.sqrt(disc)
    return K, H, H + root, H - root, H2                    # pct:428-431


# --------------------------------------------------------------------------
# A4 + A8 + A9  all-points loops (pct:635-647, 657-674, 505-509)
# --------------------------------------------------------------------------
def curvature_loop(points, idx, rows=None):
    """Reference-faithful per-point loop.  ``idx`` rows align with ``rows``.

    Returns (coefs (M,6) f32, K (M,) f32, H (M,) f32, H2 (M,) f32).
    """
    points = np.asarray(points)
    rows = range(len(idx)) if rows is None else rows
    M = len(idx)
    coefs = np.empty((M, 6), dtype=np.float32)
    K = np.empty(M, dtype=np.float32)
    H = np.empty(M, dtype=np.float32)
    H2 = np.empty(M, dtype=np.float32)
    for o, i in enumerate(rows):
        nb = points[idx[o]]                                # pct:640
        centred = nb - points[i]                           # pct:641 (native dtype)
        c = quadric_fit(plane_align(centred))              # pct:644-647
        coefs[o] = c
        k_g, k_h, _, _, k_h2 = quadric_curvatures(c)       # pct:668
        K[o], H[o], H2[o] = k_g, k_h, k_h2
    return coefs, K, H, H2


def pipeline_loop(points, k, rows=None):
    """plant_kdtree(k) -> compute_pointwise_explicit_quadratic_curvature()."""
    points = np.asarray(points)
    rows = range(len(points)) if rows is None else list(rows)
    idx, dists = knn_loop(points, k, rows)
    coefs, K, H, H2 = curvature_loop(points, idx, rows)
    return dict(idx=idx, dists=dists, coefs=coefs, K=K, H=H, H2=H2)


# --------------------------------------------------------------------------
# Vectorised float64 restatement of A4-A8 (same arithmetic choreography:
# native-dtype centring, float64 covariance/eigen/rotation, float32 design
# matrix, float64 solve, float32 coefficients, float32 curvature formulas).
# --------------------------------------------------------------------------
def curvature_batched(points, idx, rows=None, count=None, chunk=65536):
    points = np.asarray(points)
    idx = np.asarray(idx)
    M, k = idx.shape
    rows = np.arange(M) if rows is None else np.asarray(rows)
    coefs = np.empty((M, 6), dtype=np.float32)
    for s in range(0, M, chunk):
        e = min(M, s + chunk)
        cnt = None if count is None else np.asarray(count[s:e])
        coefs[s:e] = _fit_chunk(points, idx[s:e], rows[s:e], cnt)
    K, H, H2 = _curv_f32(coefs)
    return coefs, K, H, H2


def _fit_chunk(points, idx, rows, count):
    m, k = idx.shape
    if count is not None:
        valid = np.arange(k)[None, :] < count[:, None]
        safe = np.where(valid, idx, 0)
    else:
        valid = np.ones((m, k), dtype=bool)
        safe = idx
    q = points[safe] - points[rows][:, None, :]            # native dtype (pct:641)
    q64 = q.astype(np.float64)
    w = valid.astype(np.float64)[..., None]
    n = valid.sum(1).astype(np.float64)
    mean = (q64 * w).sum(1) / n[:, None]
    d = (q64 - mean[:, None, :]) * w
    cov = np.einsum("mki,mkj->mij", d, d) / (n - 1)[:, None, None]   # ddof=1 (pct:277)
    _, vec = np.linalg.eigh(cov)
    nrm = vec[:, :, 0]                                     # smallest eigenvalue == Vt[-1]
    last = np.take_along_axis(q, (valid.sum(1) - 1)[:, None, None].repeat(3, 2), 1)[:, 0]
    ref = (last - q[:, 0]).astype(np.float64)              # pct:286 (native dtype subtraction)
    n_hat = nrm / np.sqrt((nrm * nrm).sum(1))[:, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        ref_hat = ref / np.sqrt((ref * ref).sum(1))[:, None]
    flip = (n_hat * ref_hat).sum(1) < 0                    # NaN -> no flip (pct:296)
    nrm = np.where(flip[:, None], -nrm, nrm)
    a = nrm / np.sqrt((nrm * nrm).sum(1))[:, None]
    v0, v1 = a[:, 1], -a[:, 0]                             # a x z
    c = a[:, 2]
    s2 = v0 * v0 + v1 * v1
    s = np.sqrt(s2)
    with np.errstate(invalid="ignore", divide="ignore"):
        f = (1 - c) / (s ** 2)
    rot = np.zeros((m, 3, 3))
    rot[:, 0, 0] = 1 - v1 * v1 * f
    rot[:, 0, 1] = v0 * v1 * f
    rot[:, 0, 2] = v1
    rot[:, 1, 0] = v0 * v1 * f
    rot[:, 1, 1] = 1 - v0 * v0 * f
    rot[:, 1, 2] = -v0
    rot[:, 2, 0] = -v1
    rot[:, 2, 1] = v0
    rot[:, 2, 2] = 1 - (v0 * v0 + v1 * v1) * f
    ident = s == 0                                         # pct:308-309
    rot[ident] = np.eye(3)
    p = np.einsum("mij,mkj->mki", rot, q64).astype(np.float32)     # pct:315, pct:350
    aa, bb, zz = p[..., 0], p[..., 1], p[..., 2]
    X = np.stack([aa * aa, bb * bb, aa * bb, aa, bb, np.ones_like(aa)], -1)   # float32 (pct:358)
    X64 = X.astype(np.float64) * w
    z64 = zz.astype(np.float64) * w[..., 0]
    # scale columns by a power of two of the neighbourhood radius (exact) so the
    # normal equations stay well conditioned at any point density.
    h = np.sqrt(np.max((q64 * q64).sum(2) * w[..., 0], axis=1))
    e = np.where(h > 0, -np.floor(np.log2(np.where(h > 0, h, 1.0))), 0.0)
    sc = np.exp2(e)
    D = np.stack([sc * sc, sc * sc, sc * sc, sc, sc, np.ones_like(sc)], -1)
    Xs = X64 * D[:, None, :]
    G = np.einsum("mki,mkj->mij", Xs, Xs)
    r = np.einsum("mki,mk->mi", Xs, z64)
    out = np.empty((m, 6), dtype=np.float32)
    try:
        sol = np.linalg.solve(G, r[..., None])[..., 0]
        out[:] = (sol * D).astype(np.float32)
    except np.linalg.LinAlgError:
        for i in range(m):
            out[i] = np.linalg.lstsq(X64[i][valid[i]], z64[i][valid[i]], rcond=None)[0]
    return out


def _curv_f32(coefs):
    """pct:398-431 vectorised, float32 arithmetic in the same operation order."""
    c = np.asarray(coefs, dtype=np.float32)
    A, B, C, D, E = (c[:, i] for i in range(5))
    one, two = np.float32(1), np.float32(2)
    Fx, Fy, Fxx, Fyy, Fxy = D, E, two * A, two * B, C
    w = one + Fx * Fx + Fy * Fy
    den_k = w * w
    den_h = np.power(w, np.float32(1.5))
    K = (Fxx * Fyy - Fxy * Fxy) / den_k
    H = ((one + Fx * Fx) * Fyy - two * Fx * Fy * Fxy + (one + Fy * Fy) * Fxx) / (two * den_h)
    return K.astype(np.float32), H.astype(np.float32), (H * H).astype(np.float32)


def pipeline_batched(points, k, eps=None, rows=None, workers=-1, tree=None):
    points = np.asarray(points)
    rows_a = np.arange(len(points)) if rows is None else np.asarray(rows)
    if eps is None:
        idx, dists = knn(points, k, workers=workers, query_rows=rows_a, tree=tree)
        count = None
    else:
        idx, dists, count = knn(points, k, eps=eps, workers=workers, query_rows=rows_a, tree=tree)
    coefs, K, H, H2 = curvature_batched(points, idx, rows_a, count)
    out = dict(idx=idx, dists=dists, coefs=coefs, K=K, H=H, H2=H2)
    if count is not None:
        out["count"] = count
    return out


# --------------------------------------------------------------------------
# A10  explicit_quadratic_neighbor_study  (pct:732-800)
# --------------------------------------------------------------------------
def neighbor_study(points, sample_rows, tol=1e-7, lower_bound=3, upper_bound=99, tree=None):
    """Bisection on the neighbour count per sampled point (pct:772-800).

    ``sample_rows`` replaces the reference's unseeded ``np.random.randint``
    draw (pct:753) so that the result is reproducible.  The neighbourhood
    here *includes* the point itself (pct:759-761), unlike A4.
    Returns (result_int, per-point converged counts).
    """
    points = np.asarray(points)
    tree = tree if tree is not None else _tree(points)

    def k_gauss(p, n):
        nb = points[tree.query(p, n + 1)[1]]               # pct:759-760
        rot = plane_align(nb - p)                          # pct:761-763
        try:
            cf = quadric_fit(rot)                          # pct:765
        except Exception:
            cf = (0, 0, 0, 0, 0, 0)                        # pct:766-767
        return quadric_curvatures(cf)[0]

    conv = []
    for i in sample_rows:
        p = points[i]
        lo, hi, best = lower_bound, upper_bound, None
        while lo <= hi:                                    # pct:778-786
            mid = (lo + hi) // 2
            if abs(k_gauss(p, mid + 1) - k_gauss(p, mid)) < tol:
                best, hi = mid, mid - 1
            else:
                lo = mid + 1
        conv.append(hi if best is None else best)          # pct:787-788
    if not conv:
        return 0, conv
    return int(np.mean(conv)) + 1, conv                    # pct:800


# --------------------------------------------------------------------------
# Parity contract: |x - ref| <= 1e-5 * max(|ref|, floor)
# --------------------------------------------------------------------------
def curvature_tolerance_ok(x, ref, floor, rtol=1e-5):
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(x - ref) <= rtol * np.maximum(np.abs(ref), floor)


# --------------------------------------------------------------------------
# N3  load_mesh_compute_energies  (/root/reference/utils.py:702-765)
# utils.py cannot be imported here (needs open3d / pyvista), so this restatement is pinned by the closed-form
# energies the reference itself quotes (main_shape_validation.py:33-45: sphere 4*pi / 4*pi) -- "parity unpinned"
# by a reference run.
# --------------------------------------------------------------------------
def mesh_energies(vertices, triangles, gaussian_curvature, mean_curvature):
    vertices = np.asarray(vertices)
    triangles = np.asarray(triangles)
    areas = np.zeros(len(triangles))
    for i, tri in enumerate(triangles):                               # utils.py:723-728
        v0, v1, v2 = vertices[tri[0]], vertices[tri[1]], vertices[tri[2]]
        areas[i] = 0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0))
    g = np.asarray(gaussian_curvature)
    m = np.asarray(mean_curvature)
    m2 = m ** 2                                                       # utils.py:745
    fk = np.zeros(len(triangles)); fm2 = np.zeros(len(triangles))
    for i, tri in enumerate(triangles):                               # utils.py:753-758
        verts = np.array(tri)
        fk[i] = np.mean(g[verts])
        fm2[i] = np.mean(m2[verts])
    return np.nansum(fm2 * areas), np.nansum(fk * areas), np.sum(areas)


# --------------------------------------------------------------------------
# N4  scan preparation.  Neither MODULE is importable here (convert_asc_to_ply.py runs its conversion at import time,
# utils.py needs open3d), but the two function definitions compile on their own: oracle/make_goldens_prep.py runs the
# reference's own bodies on seeded inputs (tests/golden/g11_prep.npz) and tests/test_prep.py holds these restatements
# to them -- pinned.
# --------------------------------------------------------------------------
def voxel_downsample(coordinates, voxel_size=0.1):
    """/root/reference/convert_asc_to_ply.py:20-51."""
    coordinates = np.array(coordinates)
    voxel_indices = np.floor(coordinates / voxel_size).astype(np.int32)            # :34
    voxel_dict = {}
    for i, voxel in enumerate(voxel_indices):                                       # :39-46
        key = tuple(voxel)
        if key not in voxel_dict:
            voxel_dict[key] = coordinates[i]
    return np.array(list(voxel_dict.values()))                                      # :49


def surface_variation(points, k_fraction=0.025, max_neighbors=100, as_written=False):
    """/root/reference/utils.py:778-829 (estimate_curvature).

    QUIRK: the reference's subscripts 'nik,njk->nij' (utils.py:822) contract the COORDINATE axis, so its "covariance"
    is the k x k Gram matrix of the neighbourhood (rank <= 3), not the 3 x 3 covariance its comments describe.  The
    smallest eigenvalue of that matrix is zero up to LAPACK round-off for k > 3, i.e. the function returns ~1e-9
    noise.  ``as_written=True`` reproduces that; the default restates what the docstring and comments specify
    (smallest / sum of the eigenvalues of the 3 x 3 covariance; both matrices have the same trace).
    """
    from sklearn.neighbors import NearestNeighbors
    num_points = len(points)
    k = min(max(5, int(k_fraction * num_points)), max_neighbors)                    # :807
    nbrs = NearestNeighbors(n_neighbors=k).fit(points)                              # :810
    _, indices = nbrs.kneighbors(points)                                            # :812
    neighbors = points[indices]
    means = neighbors.mean(axis=1, keepdims=True)                                   # :818
    centered = neighbors - means
    if as_written:
        cov = np.einsum('nik,njk->nij', centered, centered) / (k - 1)               # :822 verbatim subscripts
    else:
        centered = centered.astype(np.float64)
        cov = np.einsum('nki,nkj->nij', centered, centered) / (k - 1)               # the documented (dim, dim) covariance
    eigenvalues, _ = np.linalg.eigh(cov)                                            # :825
    sums = np.sum(eigenvalues, axis=1)
    return eigenvalues[:, 0] / (sums + 1e-10)                                       # :828
