"""Parity of the HIP path (through the C ABI) with the oracle and the reference goldens.

Bars:  neighbour indices / float32 distances bit-exact;
K, H:  |x - ref| <= 1e-5 * max(|ref|, 1e-2 * scale)  with scale = max|ref| of the input.
"""
import os

import numpy as np
import pytest

import pct_oracle as oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-5          # north-star tolerance (float64 reference, relative)
FLOOR = 1e-2         # absolute floor as a fraction of the input's curvature scale


def assert_curvature(K, H, refK, refH, mask=None):
    fK, fH = FLOOR * np.nanmax(np.abs(refK)), FLOOR * np.nanmax(np.abs(refH))
    okK = oracle.curvature_tolerance_ok(K, refK, fK, RTOL)
    okH = oracle.curvature_tolerance_ok(H, refH, fH, RTOL)
    if mask is not None:
        okK, okH = okK | ~mask, okH | ~mask
    assert okK.all(), f"K: {np.count_nonzero(~okK)} rows outside 1e-5 (worst {np.nanmax(np.abs(K - refK)):.3e})"
    assert okH.all(), f"H: {np.count_nonzero(~okH)} rows outside 1e-5 (worst {np.nanmax(np.abs(H - refH)):.3e})"


def run_cloud(gpu, pts, k, algorithm="auto", eps=None):
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.collect_stats = True
    pc.plant_kdtree(k, eps=eps, algorithm=algorithm)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    return pc, K, H


def test_cross_lane_primitives(gpu):
    h = gpu["capi"].Handle(0)
    assert h.selftest() == 0
    h.close()


# ------------------------------------------------------------------ goldens
@pytest.mark.parametrize("name", ["g1_sphere2k_k30.npz", "g2_torus4k_k50.npz", "g3_egg4k_k50.npz"])
@pytest.mark.parametrize("algorithm", ["brute", "grid"])
def test_reference_goldens(gpu, golden, name, algorithm):
    g = golden(name)
    pc, K, H = run_cloud(gpu, g["points"], int(g["k"]), algorithm)
    assert pc.neighbor_indices.dtype == np.int32 and pc.dists.dtype == np.float32
    assert np.array_equal(pc.neighbor_indices, g["idx"])
    assert np.array_equal(pc.dists, g["dists"])
    assert K.dtype == np.float32 and H.dtype == np.float32 and K.shape == (len(g["points"]),)
    assert_curvature(K, H, g["K"], g["H"])
    assert_curvature(pc.K_H_sq_quadratic, H, g["H2"], g["H"])
    co = np.asarray(pc.quadratic_coefficients)
    assert co.shape == (len(K), 6) and co.dtype == np.float32
    assert (co == g["coefs"]).all(1).mean() > 0.99            # float32 rounding of an fp64 solve
    assert len(pc.K_quadratic) == len(K) and pc.K_quadratic[3] == K[3]


def test_file_constructor_golden(gpu, golden, tmp_path):
    g = golden("g4_bunny4k_file_k30.npz")
    f = tmp_path / "scan.txt"
    np.savetxt(f, g["raw"])
    pc = gpu["PointCloud"](str(f))
    pc.plant_kdtree(30)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    assert np.array_equal(pc.neighbor_indices, g["idx"]) and np.array_equal(pc.dists, g["dists"])
    assert_curvature(K, H, g["K"], g["H"])


def test_whole_bunny_scan_through_the_file_constructor(gpu, golden, tmp_path):
    """BASELINE.md's first plumbing line in full: all 35 947 rows of sample_scans/bunny.txt, file constructor (parser,
    float32 cast, max-shift of pct:56-57), k = 30, against the unmodified reference run on the whole file
    (oracle/make_goldens_bunny_full.py: 3 000 sampled rows of every output, the shifted cloud in full)."""
    g = golden("g4_bunny_full_file_k30_sample.npz")
    raw = np.load(os.path.join(os.path.dirname(__file__), "golden", "bunny_xyz_f32.npy"))
    assert len(raw) == int(g["n"]) == 35947
    f = tmp_path / "bunny.txt"
    np.savetxt(f, raw.astype(np.float64))                  # (the float32 values of the scan's text, as text again)
    pc = gpu["PointCloud"](str(f))
    assert pc.points.dtype == np.float32 and np.array_equal(pc.points, g["points"])     # the shifted cloud, bit for bit
    pc.plant_kdtree(30)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    rows = g["rows"]
    assert np.array_equal(pc.neighbor_indices[rows], g["idx"]) and np.array_equal(pc.dists[rows], g["dists"])
    assert_curvature(K[rows], H[rows], g["K"], g["H"])
    coefs = np.asarray(pc.quadratic_coefficients)[rows]
    assert (coefs == g["coefs"]).all(1).mean() > 0.98


def test_regular_grid_ties(gpu, golden):
    """Exact lattice (sample_scans/egg_carton.txt corner): k-th/(k+1)-th distances tie."""
    g = golden("g5_egggrid64_k30.npz")
    pc, K, H = run_cloud(gpu, g["points"], 30, "grid")
    assert np.array_equal(pc.dists, g["dists"])               # sorted distance rows are tie-independent
    same = (pc.neighbor_indices == g["idx"]).all(1)
    same_set = np.array([set(a) == set(b) for a, b in zip(pc.neighbor_indices, g["idx"])])
    # rows may differ from cKDTree's arbitrary tie order only where equal distances occur
    for i in np.where(~same)[0]:
        diff = pc.neighbor_indices[i] != g["idx"][i]
        d = g["dists"][i]
        for j in np.where(diff)[0]:
            assert (d == d[j]).sum() >= 2 or j == 29
    assert_curvature(K, H, g["K"], g["H"], mask=same)
    # identical indices in -> contract out, for every row
    pc.neighbor_indices = g["idx"]
    K2, H2 = pc.compute_pointwise_explicit_quadratic_curvature()
    assert_curvature(K2, H2, g["K"], g["H"])
    assert same_set.mean() > 0.9


def test_unit_neighbourhood_branches(gpu, golden):
    """G6: plane (s == 0 branch, pct:308), paraboloids (orientation), saddle, tilted patch."""
    g = golden("g6_unit_cases.npz")
    capi = gpu["capi"]
    h = capi.Handle(0)
    for n in sorted(k[:-3] for k in g if k.endswith("_in")):
        nb = g[n + "_in"]
        cloud = np.vstack([np.zeros((1, 3), nb.dtype), nb])   # query at the origin, neighbours as given
        h.set_points(cloud)
        h.fit_indices(np.arange(1, len(cloud), dtype=np.int32)[None, :], query=np.array([0]))
        co, K, H, H2 = h.get_fit(0, 1)
        ref = g[n + "_curv"]
        scale = max(1.0, float(np.abs(g[n + "_coefs"][:3]).max()))
        assert np.allclose(co[0], g[n + "_coefs"], rtol=1e-5, atol=2e-6 * scale), n
        assert abs(K[0] - ref[0]) <= 1e-5 * max(abs(ref[0]), 1e-2 * scale * scale), n
        assert abs(H[0] - ref[1]) <= 1e-5 * max(abs(ref[1]), 1e-2 * scale), n
    h.close()


# ---------------------------------------------- ill-conditioned / rank-deficient neighbourhoods (lstsq = gelsd, pct:359)
def _design_diagnostics(P, i, nb):
    """Singular values of the reference's design matrix of one neighbourhood (relative to the largest) and the
    orientation test's dot product (pct:293): what decides whether the reference's own answer is well defined."""
    q = P[nb] - P[i]
    cov = np.cov(q, rowvar=False)
    n = np.linalg.svd(cov)[2][-1]
    ref = q[-1] - q[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        dot = np.dot(n / np.linalg.norm(n), ref / np.linalg.norm(ref))
    p = np.array(oracle.plane_align(q), dtype=np.float32)
    a, b = p[:, 0], p[:, 1]
    X = np.column_stack((a ** 2, b ** 2, a * b, a, b, np.ones_like(a))).astype(np.float32).astype(np.float64)
    s = np.linalg.svd(X, compute_uv=False)
    return s / s[0], dot


def test_degenerate_unit_neighbourhoods(gpu, golden):
    """G6b: what the unmodified reference returns for collinear points, a planar curve, repeated points and
    under-determined neighbourhoods -- gelsd cuts the singular values below eps max(m,6) sigma_1 and returns the
    minimum-norm solution (pct:359); the kernel's SVD path (k_fit_svd) does the same."""
    g = golden("g6b_degenerate_unit_cases.npz")
    h = gpu["capi"].Handle(0)
    for n in sorted(k[:-3] for k in g if k.endswith("_in")):
        nb = g[n + "_in"]
        cloud = np.vstack([np.zeros((1, 3), nb.dtype), nb])
        h.set_points(cloud)
        h.fit_indices(np.arange(1, len(cloud), dtype=np.int32)[None, :], query=np.array([0]))
        co, K, H, _ = h.get_fit(0, 1)
        ref_c, ref = g[n + "_coefs"], g[n + "_curv"]
        assert np.isfinite(co).all() and np.isfinite(K[0]) and np.isfinite(H[0]), n
        if n == "line_noise1e6":
            # a singular value sits AT the cut-off (sigma_6 / sigma_1 ~ 1e-15 vs 6.7e-15): whether LAPACK keeps it is
            # decided by its rounding -- the reference's answer is not a function of the input here.  Finite, no more.
            continue
        if n == "line_oblique":
            # collinear up to float64 rounding: the right-hand side is rounding noise (1e-18), the answer noise / sigma
            assert np.abs(co[0]).max() < 1e-5 and abs(K[0]) < 1e-8 and abs(H[0]) < 1e-5, n
            continue
        scale = max(1.0, float(np.abs(ref_c[:3]).max()))
        assert np.allclose(co[0], ref_c, rtol=1e-5, atol=2e-6 * scale), (n, co[0], ref_c)
        assert abs(K[0] - ref[0]) <= 1e-5 * max(abs(ref[0]), 1e-2 * scale * scale), n
        assert abs(H[0] - ref[1]) <= 1e-5 * max(abs(ref[1]), 1e-2 * scale), n
        if n != "six_points_x5":
            assert h.timings()["fit_svd_rows"] == 1, n          # these are the rows the normal equations cannot do
    h.close()


def test_staticmethods_of_the_class(gpu, golden):
    """PointCloud.get_best_fit_plane_and_rotate / fit_quadratic_surface (pct:270-321, 331-360) as callable
    staticmethods, against the blocks the reference's own staticmethods returned (G6 / G6b `_rot`, `_coefs`)."""
    PC = gpu["PointCloud"]
    free_normal = {"line_axis", "line_axis_f32", "line_oblique", "line_noise1e6", "planar_curve", "aniso_lattice_1to20"}
    for name in ("g6_unit_cases.npz", "g6b_degenerate_unit_cases.npz"):
        g = golden(name)
        for n in sorted(k[:-3] for k in g if k.endswith("_in")):
            nb, ref_rot, ref_c = g[n + "_in"], g[n + "_rot"], g[n + "_coefs"]
            rot = PC.get_best_fit_plane_and_rotate(nb)
            assert rot.shape == nb.shape and rot.dtype == np.float64, n
            if n not in free_normal:       # (collinear / planar input: the normal is LAPACK's pick in a null space)
                assert np.allclose(rot, ref_rot, rtol=0, atol=1e-13 * np.abs(nb).max()), n
            # lengths and the best-fit normal's alignment survive whatever the pick
            assert np.allclose(np.linalg.norm(rot, axis=1), np.linalg.norm(nb.astype(np.float64), axis=1), rtol=1e-12, atol=1e-18), n
            cf = PC.fit_quadratic_surface(ref_rot)                        # the reference's own rotated block in
            assert cf.shape == (6,) and cf.dtype == np.float32, n
            if n not in ("line_noise1e6", "line_oblique"):
                scale = max(1.0, float(np.abs(ref_c[:3]).max()))
                assert np.allclose(cf, ref_c, rtol=1e-5, atol=2e-6 * scale), (n, cf, ref_c)
            out = PC.calculate_explicit_quadratic_curvatures(ref_c)
            assert np.allclose(np.array(out, np.float32), g[n + "_curv"], rtol=1e-6, atol=1e-30), n
    with pytest.raises(ValueError, match="Non-finite values in input points"):
        PC.get_best_fit_plane_and_rotate(np.array([[0, 0, 0], [1, np.nan, 0], [0, 1, 0.0]]))
    with pytest.raises(ValueError, match="Input points must have shape"):
        PC.fit_quadratic_surface(np.zeros((5, 2)))
    with pytest.raises(ValueError, match="Input contains non-finite values"):
        PC.fit_quadratic_surface(np.array([[0, 0, 0], [1, np.inf, 0], [0, 1, 0.0]]))
    # the loop the reference runs (pct:638-647) reproduces the fused result row by row
    g2 = golden("g2_torus4k_k50.npz")
    P = g2["points"]
    for i in (0, 17, 3999):
        rot = PC.get_best_fit_plane_and_rotate(P[g2["idx"][i]] - P[i])
        cf = PC.fit_quadratic_surface(rot)
        assert np.allclose(cf, g2["coefs"][i], rtol=1e-5, atol=1e-6 * np.abs(g2["coefs"][i]).max())


def test_batched_plane_rotate_and_quadric_entries(gpu, golden):
    """pct_plane_rotate / pct_fit_quadric with a whole block of neighbourhoods per call (what a caller with many
    neighbourhoods would do instead of the reference's per-point loop), float32 and float64 input."""
    g = golden("g2_torus4k_k50.npz")
    P, idx = g["points"], g["idx"]
    rows = np.arange(0, 4000, 13)
    h = gpu["capi"].Handle(0)
    for dtype in (np.float32, np.float64):
        block = (P[idx[rows]] - P[rows][:, None, :]).astype(dtype)              # (batch, 50, 3) centred neighbourhoods
        rot = h.plane_rotate(block)
        assert rot.shape == block.shape and rot.dtype == np.float64
        for b in (0, 7, len(rows) - 1):
            assert np.allclose(rot[b], oracle.plane_align(block[b]), rtol=0, atol=1e-13)
        co = h.fit_quadric(rot.astype(np.float32))
        assert co.shape == (len(rows), 6) and co.dtype == np.float32
        if dtype == np.float32:
            assert (co == g["coefs"][rows]).all(1).mean() > 0.95
        assert np.allclose(co, g["coefs"][rows], rtol=1e-5, atol=1e-6 * np.abs(g["coefs"][rows]).max())
    with pytest.raises(ValueError, match="Non-finite values in input points"):
        bad = block.copy()
        bad[3, 4, 1] = np.nan
        h.plane_rotate(bad)
    with pytest.raises(ValueError):
        h.plane_rotate(block[:, :1])                                           # one point has no covariance
    h.close()


@pytest.mark.parametrize("tag", ["plane_1to20", "cyl_1to20", "wavy_1to20", "wavy_1to4", "wavy_1to4_jitter"])
def test_scan_line_clouds_against_the_reference(gpu, golden, tag):
    """G10: clouds sampled densely along scan lines and sparsely across (0.005 x 0.1 / 0.02, k = 30: every
    neighbourhood of the 1 : 20 clouds lies on ONE line), run through the whole reference class.  Row classes:
      stable   no singular value of the design matrix within 100x of gelsd's cut-off and the kept ones above 1e-6
               sigma_1: the usual 1e-5 contract;
      noisy    kept singular values below 1e-6 sigma_1: the reference's own answer moves by more than 1e-5 when its
               input moves by an ulp (condition^2 x eps x residual): compared at 1e-3;
      band     a singular value within 100x of the cut-off: whether LAPACK keeps it is decided by rounding: finite only;
      and where the orientation test's dot product (pct:293) is zero or rounding noise, the sign of the normal is
      LAPACK's arbitrary sign of Vt[-1]: H is compared up to sign."""
    g = golden(f"g10_scanline_{tag}_k30.npz")
    P, k = g["points"], int(g["k"])
    h = gpu["capi"].Handle(0)
    h.set_points(P)
    h.knn(k, 0.0, gpu["capi"].KNN_GRID)
    idx, dist, _ = h.get_neighbors(0, len(P))
    assert np.array_equal(dist, g["dists"])                       # (index rows differ inside the lattice's exact ties)
    h.fit_indices(g["idx"])                                       # the reference's own neighbourhoods, every row
    co, K, H, _ = h.get_fit(0, len(P))
    svd_rows = h.timings()["fit_svd_rows"]
    h.close()
    assert np.isfinite(K).all() and np.isfinite(H).all() and np.isfinite(g["K"]).all()
    assert (svd_rows > 0.9 * len(P)) == ("1to20" in tag)
    rows = np.random.default_rng(3).choice(len(P), 1200, replace=False)
    diag = [_design_diagnostics(P, i, g["idx"][i]) for i in rows]
    rel = np.array([d[0] for d in diag])
    dot = np.array([d[1] for d in diag])
    rc = np.finfo(np.float64).eps * k
    band = ((rel > rc / 100) & (rel < rc * 100)).any(1)
    kept_min = np.where(rel > rc, rel, np.inf).min(1)
    noisy = ~band & (kept_min < 1e-6)
    stable = ~band & ~noisy
    unsigned = ~(np.abs(dot) > 1e-9)                              # zero, noise or NaN
    Kg, Hg, Kr, Hr = K[rows], H[rows], g["K"][rows], g["H"][rows]
    Hg = np.where(unsigned, np.abs(Hg), Hg)
    Hr = np.where(unsigned, np.abs(Hr), Hr)
    fK, fH = 1e-3, 1e-3                                           # absolute floors: the surfaces' curvatures are O(0.1 .. 1)
    for cls, tol in ((stable, 1e-5), (noisy, 1e-3)):
        okK = oracle.curvature_tolerance_ok(Kg[cls], Kr[cls], fK, tol)
        okH = oracle.curvature_tolerance_ok(Hg[cls], Hr[cls], fH, tol)
        assert okK.all() and okH.all(), (tag, tol, int((~okK).sum()), int((~okH).sum()))
    print(f"{tag}: svd rows {svd_rows}, sampled stable {stable.sum()} noisy {noisy.sum()} band {band.sum()} unsigned {unsigned.sum()}")


# ------------------------------------------------------- seeded vs the oracle
@pytest.mark.parametrize("k", [6, 30, 50, 63, 64, 80, 100, 127, 128, 200, 300, 511])
def test_knn_k_sweep(gpu, k):
    """cKDTree.query takes any k (pct:83): one or two list registers in the fast sweeps up to 127, beyond that the
    wave-per-query exact sweep with a 256- or 512-wide list (pct_knn_wide.hip)."""
    pts = gpu["shapes"].torus_random(20_000 if k <= 127 else 6_000, seed=100 + k)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.plant_kdtree(k, algorithm="grid")
    idx, d = oracle.knn(pts, k)
    assert np.array_equal(pc.neighbor_indices, idx) and np.array_equal(pc.dists, d)
    assert (np.diff(pc.dists, axis=1) >= 0).all()
    assert (pc.neighbor_indices != np.arange(len(pts))[:, None]).all()      # self dropped (pct:84-85)


def test_long_rows_fit_and_small_clouds(gpu):
    """k > 127 through the whole class (fit of 200-neighbour rows), through the exhaustive sweep of a small cloud, and
    the limits: k = 512 is refused with a message, k + 1 > N as the reference's IndexError path (PCT_ERR_K_TOO_LARGE)."""
    pts = gpu["shapes"].torus_random(5_000, seed=77)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.plant_kdtree(200)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    ref = oracle.pipeline_batched(pts, 200)
    assert np.array_equal(pc.neighbor_indices, ref["idx"]) and np.array_equal(pc.dists, ref["dists"])
    assert oracle.curvature_tolerance_ok(K, ref["K"], 1e-2 * np.abs(ref["K"]).max()).all()
    assert oracle.curvature_tolerance_ok(H, ref["H"], 1e-2 * np.abs(ref["H"]).max()).all()
    small = gpu["shapes"].torus_random(700, seed=78)
    ps = gpu["PointCloud"](points=small, normals=np.zeros((700, 0)))
    ps.plant_kdtree(300, algorithm="brute")
    idx, d = oracle.knn(small, 300)
    assert np.array_equal(ps.neighbor_indices, idx) and np.array_equal(ps.dists, d)
    with pytest.raises(ValueError, match="outside"):
        ps.plant_kdtree(512)


def test_points_assigned_or_edited_after_a_planting(gpu):
    """pct:74 builds a new tree from self.points AS THEY ARE at every planting and pct:640 gathers the coordinates of
    the moment: a cloud that was assigned, or rewritten in place, after an upload must not be answered from the old
    upload (round 2 did: the device copy was made once)."""
    shapes = gpu["shapes"]
    a = shapes.torus_random(6_000, seed=41)
    b = shapes.egg_carton_random(6_000, seed=42)
    pc = gpu["PointCloud"](points=a, normals=np.zeros((len(a), 0)))
    pc.plant_kdtree(30)
    assert np.array_equal(pc.neighbor_indices, oracle.knn(a, 30)[0])
    pc.points = b                                         # assigned: the next planting sees the new cloud
    pc.plant_kdtree(30)
    rb = oracle.pipeline_batched(b, 30)
    assert np.array_equal(pc.neighbor_indices, rb["idx"]) and np.array_equal(pc.dists, rb["dists"])
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    assert oracle.curvature_tolerance_ok(K, rb["K"], 1e-2 * np.abs(rb["K"]).max()).all()
    b *= np.float32(1.5)                                  # rewritten in place (same object, same buffer)
    pc.plant_kdtree(30)
    rc = oracle.pipeline_batched(b, 30)
    assert np.array_equal(pc.neighbor_indices, rc["idx"]) and np.array_equal(pc.dists, rc["dists"])
    K, H = pc.compute_curvature_fused(30)
    assert oracle.curvature_tolerance_ok(K, rc["K"], 1e-2 * np.abs(rc["K"]).max()).all()
    # edited between planting and fit: the reference fits the NEW coordinates with the table of the last planting
    pc.plant_kdtree(30)
    old_idx = pc.neighbor_indices.copy()
    b += np.float32(0.25) * np.sin(7 * b[:, ::-1])        # not a similarity: the table of the old cloud is not the new one's
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    coefs, Kr, Hr, _ = oracle.curvature_batched(b, old_idx, np.arange(len(b)), None)
    assert np.array_equal(pc.neighbor_indices, old_idx)
    assert oracle.curvature_tolerance_ok(K, Kr, 1e-2 * np.abs(Kr).max()).all() and oracle.curvature_tolerance_ok(H, Hr, 1e-2 * np.abs(Hr).max()).all()


def test_brute_and_grid_agree_bitwise(gpu):
    pts = gpu["shapes"].egg_carton_random(30_000, seed=8)
    a, Ka, Ha = run_cloud(gpu, pts, 50, "brute")
    b, Kb, Hb = run_cloud(gpu, pts, 50, "grid")
    assert np.array_equal(a.neighbor_indices, b.neighbor_indices) and np.array_equal(a.dists, b.dists)
    assert np.array_equal(Ka, Kb) and np.array_equal(Ha, Hb)


def _stress_cloud(kind, rng, n):
    if kind == "blobs":            # clusters of very different density
        c = rng.uniform(-1, 1, size=(6, 3))
        sc = 10.0 ** rng.uniform(-3.5, -0.5, size=6)
        w = rng.integers(0, 6, size=n)
        p = c[w] + rng.normal(size=(n, 3)) * sc[w, None]
    elif kind == "outliers":       # a surface plus far points in every direction
        p = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), np.zeros(n)], 1)
        p[:, 2] = 0.2 * np.sin(3 * p[:, 0]) * np.cos(2 * p[:, 1])
        m = max(3, n // 2000)
        p[rng.choice(n, m, replace=False)] = rng.normal(size=(m, 3)) * 10.0 ** rng.uniform(1, 2.5, size=(m, 1))
    elif kind == "noisy_line":     # nearly one-dimensional
        t = rng.uniform(0, 10, n)
        p = np.stack([t, 0.3 * t, -0.1 * t], 1) + rng.normal(scale=1e-3, size=(n, 3))
    elif kind == "shell_and_core": # a sphere with a dense core: many queries need more than the stencil
        u = rng.normal(size=(n, 3))
        u /= np.linalg.norm(u, axis=1)[:, None]
        r = np.where(rng.uniform(size=n) < 0.3, rng.uniform(0, 0.01, n), 1.0)
        p = u * r[:, None]
    else:                          # "quantised": coordinates on a coarse lattice -> masses of exact ties
        p = np.round(rng.uniform(-1, 1, size=(n, 3)) * 40) / 40
        p[:, 2] = np.round(0.3 * np.sin(p[:, 0] * 3) * 40) / 40
    return np.ascontiguousarray(p, dtype=np.float32)


@pytest.mark.parametrize("kind", ["blobs", "outliers", "noisy_line", "shell_and_core", "quantised"])
def test_grid_sweep_equals_exhaustive_sweep_on_hostile_clouds(gpu, kind):
    """Cell list (trimmed box, clamped outliers, fast + exact sweeps, owned-range culling) against the
    exhaustive sweep of the same library, bit for bit, on clouds built to break a uniform grid.  The
    exhaustive sweep itself is pinned to the oracle by the tests above."""
    capi = gpu["capi"]
    for n, k, eps in ((20_000, 30, 0.0), (12_000, 70, 0.0), (15_000, 25, 0.05)):
        pts = _stress_cloud(kind, np.random.default_rng([len(kind), n]), n)
        h = capi.Handle(0)
        h.set_points(pts)
        h.knn(k, eps=eps, algo=capi.KNN_BRUTE)
        ib, db, cb = h.get_neighbors(0, n, want_count=True)
        h.knn(k, eps=eps, algo=capi.KNN_GRID)
        ig, dg, cg = h.get_neighbors(0, n, want_count=True)
        assert np.array_equal(ib, ig) and np.array_equal(db, dg) and np.array_equal(cb, cg), (kind, n, k, eps)
        lo, hi = n // 3, n // 3 + n // 4           # a shard in the middle
        h.set_query_range(lo, hi)
        h.knn(k, eps=eps, algo=capi.KNN_GRID)
        i2, d2, c2 = h.get_neighbors(lo, hi, want_count=True)
        assert np.array_equal(i2, ib[lo:hi]) and np.array_equal(d2, db[lo:hi]) and np.array_equal(c2, cb[lo:hi])
        h.close()


@pytest.mark.parametrize("k", [20, 50, 100])
def test_exact_sweep_matches_fast_sweep(gpu, k):
    """The float-key fast sweep (+ redo of flagged queries) and the all-exact sweep agree bit for bit."""
    pts = gpu["shapes"].torus_random(30_000, seed=55)
    a, Ka, Ha = run_cloud(gpu, pts, k, "grid")
    b, Kb, Hb = run_cloud(gpu, pts, k, "grid_exact")
    assert np.array_equal(a.neighbor_indices, b.neighbor_indices) and np.array_equal(a.dists, b.dists)
    assert np.array_equal(Ka, Kb) and np.array_equal(Ha, Hb)
    assert a.last_timings["redone_queries"] < 0.1 * len(pts)        # key collisions, ring fallbacks, LDS overflows are rare


def test_lattice_block_equals_the_exhaustive_sweep(gpu, golden):
    g = golden("g5_egggrid64_k30.npz")
    pc = gpu["PointCloud"](points=g["points"], normals=np.zeros((len(g["points"]), 0)))
    pc.collect_stats = True
    pc.plant_kdtree(30, algorithm="grid")
    print("egg lattice block: redone", pc.last_timings["redone_queries"], "of", len(g["points"]))
    ex = gpu["PointCloud"](points=g["points"], normals=np.zeros((len(g["points"]), 0)))
    ex.plant_kdtree(30, algorithm="brute")
    assert np.array_equal(pc.neighbor_indices, ex.neighbor_indices) and np.array_equal(pc.dists, ex.dists)


@pytest.mark.parametrize("shape,k", [("sphere", 30), ("torus", 50), ("egg", 50)])
def test_pipeline_vs_oracle_100k(gpu, shape, k):
    sh = gpu["shapes"]
    pts = {"sphere": lambda: sh.fibonacci_sphere(100_000), "torus": lambda: sh.torus_random(100_000, seed=31),
           "egg": lambda: sh.egg_carton_random(100_000, seed=32)}[shape]()
    pc, K, H = run_cloud(gpu, pts, k)
    ref = oracle.pipeline_batched(pts, k)
    assert np.array_equal(pc.neighbor_indices, ref["idx"]) and np.array_equal(pc.dists, ref["dists"])
    assert_curvature(K, H, ref["K"], ref["H"])
    # the vectorised flavour solves normal equations like the kernel's main path: a shared blind spot.  Rows of the
    # reference-faithful loop (np.cov / svd / lstsq per point, bit-identical to the reference on the goldens) as well.
    rows = np.random.default_rng(9).choice(len(pts), 1500, replace=False)
    loop = oracle.pipeline_loop(pts, k, rows)
    assert np.array_equal(pc.neighbor_indices[rows], loop["idx"])
    assert_curvature(K[rows], H[rows], loop["K"], loop["H"])
    assert pc.last_timings["fit_svd_rows"] == 0                 # nothing ill-conditioned on these shapes
    if shape == "sphere":     # closed form K = H = 1 up to the estimator's own O(h^2) bias
        assert np.abs(K - 1).max() < 2e-3 and np.abs(H - 1).max() < 2e-3


def test_float64_cloud_native_dtype(gpu):
    """float64 points: float32-rounded tree, float64 queries and centring (pct:74, 83, 641)."""
    pts = gpu["shapes"].torus_random(20_000, seed=77, dtype=np.float64)
    pc, K, H = run_cloud(gpu, pts, 40)
    ref = oracle.pipeline_batched(pts, 40)
    assert np.array_equal(pc.neighbor_indices, ref["idx"]) and np.array_equal(pc.dists, ref["dists"])
    assert_curvature(K, H, ref["K"], ref["H"])


def test_hybrid_eps_query(gpu):
    pts = gpu["shapes"].egg_carton_random(20_000, seed=4)
    pc, K, H = run_cloud(gpu, pts, 50, eps=0.06)
    ref = oracle.pipeline_batched(pts, 50, eps=0.06)
    assert np.array_equal(pc.neighbor_counts, ref["count"])
    assert 6 <= ref["count"].min() < 50 == ref["count"].max()           # both branches exercised
    assert np.array_equal(pc.neighbor_indices, ref["idx"]) and np.array_equal(pc.dists, ref["dists"])
    assert_curvature(K, H, ref["K"], ref["H"])


def test_density_contrast_forces_ring_fallback_and_lds_overflow(gpu):
    rng = np.random.default_rng(5)
    dense = rng.normal(scale=0.002, size=(6000, 3))
    sparse = rng.uniform(-1, 1, size=(3000, 3))
    pts = np.vstack([dense, sparse]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.collect_stats = True
    pc.plant_kdtree(50, algorithm="grid")
    t = pc.last_timings
    idx, d = oracle.knn(pts, 50)
    assert np.array_equal(pc.neighbor_indices, idx) and np.array_equal(pc.dists, d)
    assert t["ring_fallbacks"] > 0 and t["lds_overflows"] > 0


def test_duplicate_points(gpu):
    pts = gpu["shapes"].torus_random(5000, seed=2)
    pts[100:200] = pts[0:100]                                  # exact twins
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.plant_kdtree(20, algorithm="grid")
    _, d = oracle.knn(pts, 20)
    assert np.array_equal(pc.dists, d)                         # tie order is arbitrary, distances are not
    assert (pc.dists[0:200, 0] == 0).all()


def test_degenerate_clouds_do_not_break_the_sweep(gpu):
    """Identical points, a line, a planar lattice: every distance ties or every stencil overflows."""
    PC = gpu["PointCloud"]
    same = np.tile(np.array([[0.25, -1.0, 3.0]], np.float32), (300, 1))
    pc = PC(points=same, normals=np.zeros((300, 0)))
    pc.plant_kdtree(10, algorithm="grid")
    # all k+1 candidates tie at distance 0: 'drop result 0' drops the lowest index, not necessarily self
    assert (pc.dists == 0).all() and (pc.neighbor_indices == np.arange(1, 11)[None, :]).all()
    line = np.zeros((5000, 3), np.float32)
    line[:, 0] = np.random.default_rng(0).uniform(0, 1, 5000)
    pc = PC(points=line, normals=np.zeros((5000, 0)))
    pc.plant_kdtree(20, algorithm="grid")
    i, d = oracle.knn(line, 20)
    assert np.array_equal(pc.dists, d)
    untied = (np.diff(d, axis=1) > 0).all(1)                     # |x-a| == |x-b| happens on a line: tie order is free
    assert untied.mean() > 0.5 and np.array_equal(pc.neighbor_indices[untied], i[untied])
    gx, gy = np.meshgrid(np.arange(70, dtype=np.float32), np.arange(70, dtype=np.float32))
    lattice = np.column_stack([gx.ravel(), gy.ravel(), np.zeros(4900, np.float32)])
    pc = PC(points=lattice, normals=np.zeros((4900, 0)))
    pc.collect_stats = True
    pc.plant_kdtree(12, algorithm="grid")
    _, d = oracle.knn(lattice, 12)
    assert np.array_equal(pc.dists, d)                           # indices: ties ordered by public index
    ex = PC(points=lattice, normals=np.zeros((4900, 0)))
    ex.plant_kdtree(12, algorithm="brute")
    assert np.array_equal(pc.neighbor_indices, ex.neighbor_indices)
    # exact ties everywhere: the fast sweep orders them in place by (d2, index); the exact sweep is the exception
    print("integer lattice: redone", pc.last_timings["redone_queries"], "of 4900")
    assert pc.last_timings["redone_queries"] < 2450


# ---------------------------------------------------------- full bench size
def test_torus_1m_sampled_oracle_and_properties(gpu, golden):
    """BASELINE config C3 at full size against the sampled reference golden G7."""
    g = golden("g7_torus1m_k50_sample.npz")
    sh = gpu["shapes"]
    pts, Ktrue, Htrue = sh.torus_random(1_000_000, seed=1234, with_truth=True)
    pc, K, H = run_cloud(gpu, pts, 50)
    rows = g["rows"]
    idx, d = pc.neighbor_indices, pc.dists
    assert np.array_equal(idx[rows], g["idx"]) and np.array_equal(d[rows], g["dists"])
    assert_curvature(K[rows], H[rows], g["K"], g["H"])
    # size-independent properties on all 1M rows
    assert (np.diff(d, axis=1) >= 0).all() and np.isfinite(d).all()
    assert idx.min() >= 0 and idx.max() < len(pts)
    assert (idx != np.arange(len(pts), dtype=np.int32)[:, None]).all()
    rec = np.linalg.norm(pts[idx[::997, -1]].astype(np.float64) - pts[::997].astype(np.float64), axis=1)
    assert np.array_equal(rec.astype(np.float32), d[::997, -1])          # distances re-derive from the indices
    # closed form (plot_shape_validation_results.py:28-45 generalised over phi): estimator bias only
    assert np.median(np.abs(K - Ktrue)) < 2e-2 and np.median(np.abs(H - Htrue)) < 1e-2


def test_sphere_100k_sampled_golden(gpu, golden):
    g = golden("g7_sphere100k_k30_sample.npz")
    pts = gpu["shapes"].fibonacci_sphere(100_000)
    pc, K, H = run_cloud(gpu, pts, 30)
    rows = g["rows"]
    assert np.array_equal(pc.neighbor_indices[rows], g["idx"]) and np.array_equal(pc.dists[rows], g["dists"])
    assert_curvature(K[rows], H[rows], g["K"], g["H"])


# ------------------------------------------------------------- boundary API
def test_query_range_shards_are_bit_identical(gpu):
    """Index-range sharding (multi-GPU ownership) does not change any value."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(40_000, seed=6)
    h = capi.Handle(0)
    h.set_points(pts)
    h.curvature(50)
    i0, d0, _ = h.get_neighbors(0, len(pts))
    _, K0, H0, _ = h.get_fit(0, len(pts))
    parts = []
    for lo, hi in [(0, 13_333), (13_333, 40_000)]:
        h.set_query_range(lo, hi)
        h.curvature(50)
        i, d, _ = h.get_neighbors(lo, hi)
        _, K, H, _ = h.get_fit(lo, hi)
        parts.append((i, d, K, H))
        with pytest.raises(ValueError):
            h.get_fit(0, len(pts))
    h.close()
    for j, ref in enumerate((i0, d0, K0, H0)):
        assert np.array_equal(np.concatenate([p[j] for p in parts]), ref)


def _slab_pass(h, pts, k, parts, eps=0.0):
    """Every slab of `parts` on one handle, its records appended to one device buffer (what the ranks' all-gather
    would assemble), then scattered back into public order.  Returns K, H, per-part timings."""
    n = len(pts)
    rec = h.device_alloc(n * 12)
    Kd, Hd = h.device_alloc(n * 4), h.device_alloc(n * 4)
    off, tms = 0, []
    try:
        for part in range(parts):
            h.set_query_slab(part, parts)
            h.curvature(k, eps)
            counts = h.slab_counts(parts)
            assert sum(counts) == n and len(counts) == parts
            rows = h.slab_records(rec + off * 12, n - off)
            assert rows == counts[part]
            t = h.timings()
            t["rows"] = rows
            tms.append(t)
            off += rows
        assert off == n
        h.scatter_records(rec, n, 0, n, Kd, Hd)
        K, H = np.empty(n, np.float32), np.empty(n, np.float32)
        h.device_download(Kd, K)
        h.device_download(Hd, H)
        # a rank scatters its own index range out of everybody's records
        lo, hi = n // 3, n // 3 + 1000
        h.scatter_records(rec, n, lo, hi, Kd, Hd)
        Ks = np.empty(hi - lo, np.float32)
        h.device_download(Kd, Ks)
        assert np.array_equal(Ks.view(np.uint32), K[lo:hi].view(np.uint32))
        # records that do not cover the range are refused (here: the last slab's rows are missing)
        if parts > 1 and tms[-1]["rows"] > 0:
            with pytest.raises(ValueError, match="partition"):
                h.scatter_records(rec, n - tms[-1]["rows"], 0, n, Kd, Hd)
    finally:
        for p_ in (rec, Kd, Hd):
            h.device_free(p_)
    return K, H, tms


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_slab_shards_of_an_unsorted_cloud_are_bit_identical(gpu, parts):
    """Ownership by slab (pct_set_query_slab): a cloud in NO spatial order is cut into slabs of equal population; each
    part's cell list holds its slab and a margin instead of the whole cloud, and K, H of every row are the bits of the
    unsharded run."""
    capi = gpu["capi"]
    pts = gpu["shapes"].egg_carton_random(300_000, seed=31)
    n, k = len(pts), 50
    h = capi.Handle(0)
    h.set_points(pts)
    h.curvature(k)
    _, K0, H0, _ = h.get_fit(0, n, coefs=False, H2=False)
    K, H, tms = _slab_pass(h, pts, k, parts)
    assert np.array_equal(K.view(np.uint32), K0.view(np.uint32)) and np.array_equal(H.view(np.uint32), H0.view(np.uint32))
    for t in tms:
        assert abs(t["rows"] - n / parts) < 0.01 * n, tms                        # equal populations (4096 bins)
        assert t["rows"] <= t["grid_points"] <= n / parts + 0.25 * n, t          # the slab and its margin, not the cloud
        assert t["limit_retries"] == 0
    # by-index getters have no meaning for a slab; an index range (or a new cloud) brings them back
    with pytest.raises(ValueError, match="slab"):
        h.get_fit(0, 10)
    with pytest.raises(ValueError, match="slab"):
        h.knn(k)
    h.set_query_range(0, n)
    h.curvature(k)
    _, K1, _, _ = h.get_fit(0, n, coefs=False, H2=False)
    assert np.array_equal(K1.view(np.uint32), K0.view(np.uint32))
    with pytest.raises(ValueError):
        h.slab_counts(parts)
    h.close()


def test_slab_margin_too_thin_is_noticed_and_redone(gpu, monkeypatch):
    """The margin either side of a slab is a guess; the sweep checks it per query and the pass is repeated with every
    point when a neighbourhood reaches past a face -- forced here by a margin of a hundredth of a cell.  Also: eps > 0,
    k = 80 (two list registers), k = 200 (exact sweep), a cloud whose longest axis holds ties."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(120_000, seed=8)
    n = len(pts)
    h = capi.Handle(0)
    for k, eps in ((30, 0.0), (80, 0.0), (40, 0.02), (200, 0.0)):
        h.set_points(pts)
        h.curvature(k, eps)
        _, K0, H0, _ = h.get_fit(0, n, coefs=False, H2=False)
        monkeypatch.setenv("PCT_SLAB_MARGIN", "0.01")
        K, H, tms = _slab_pass(h, pts, k, 4, eps)
        assert sum(t["limit_retries"] for t in tms) >= 1, tms
        assert any(t["grid_points"] == n for t in tms)
        monkeypatch.delenv("PCT_SLAB_MARGIN")
        assert np.array_equal(K.view(np.uint32), K0.view(np.uint32), ) and np.array_equal(H.view(np.uint32), H0.view(np.uint32))
        K, H, tms = _slab_pass(h, pts, k, 4, eps)
        assert sum(t["limit_retries"] for t in tms) == 0
        assert np.array_equal(K.view(np.uint32), K0.view(np.uint32)) and np.array_equal(H.view(np.uint32), H0.view(np.uint32))
    # a lattice: thousands of points share each coordinate of the cut axis, whole planes fall into one bin
    g = gpu["shapes"].egg_carton_grid(300)
    h.set_points(g)
    h.curvature(20)
    _, K0, H0, _ = h.get_fit(0, len(g), coefs=False, H2=False)
    K, H, tms = _slab_pass(h, g, 20, 7)
    assert np.array_equal(K.view(np.uint32), K0.view(np.uint32)) and np.array_equal(H.view(np.uint32), H0.view(np.uint32))
    # every point in one plane across the cut axis but a few: most slabs are empty
    flat = gpu["shapes"].egg_carton_random(20_000, seed=5).copy()
    flat[:, 0] *= 1e-3; flat[:, 1] *= 1e-3; flat[:, 2] = 0.0
    flat[:5, 2] = np.linspace(1.0, 5.0, 5)
    h.set_points(flat)
    h.curvature(20, algo=capi.KNN_GRID)
    _, K0, H0, _ = h.get_fit(0, len(flat), coefs=False, H2=False)
    K, H, tms = _slab_pass(h, flat, 20, 4)
    assert sorted(t["rows"] for t in tms)[:2] == [0, 0] or min(t["rows"] for t in tms) == 0, tms
    assert np.array_equal(K.view(np.uint32), K0.view(np.uint32), ) and np.array_equal(H.view(np.uint32), H0.view(np.uint32))
    # refused: small clouds, float64 clouds, bad parts
    h.set_points(pts[:1000])
    with pytest.raises(ValueError, match="4096"):
        h.set_query_slab(0, 2)
    h.set_points(pts.astype(np.float64))
    with pytest.raises(ValueError):
        h.set_query_slab(0, 2)
    h.set_points(pts)
    for bad in ((2, 2), (-1, 2), (0, 65)):
        with pytest.raises(ValueError):
            h.set_query_slab(*bad)
    h.set_query_slab(0, 1)                    # one part: the same path, the whole cloud as its one slab
    h.curvature(30)
    assert h.slab_counts(1) == [n]
    with pytest.raises(ValueError, match="slab"):
        h.get_fit(0, 10)
    h.set_query_slab(0, 0)                    # off: index ranges again
    h.curvature(30)
    h.get_fit(0, 10)
    h.close()


def test_scan_ordered_shards_keep_only_nearby_points(gpu):
    """A handle that owns a spatially coherent index range (scan order) packs only the points near it;
    every value stays bit-identical to the unsharded run."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(200_000, seed=16)
    theta = np.arctan2(pts[:, 1], pts[:, 0])
    pts = np.ascontiguousarray(pts[np.argsort(theta, kind="stable")])
    n = len(pts)
    h = capi.Handle(0)
    h.set_points(pts)
    h.curvature(50)
    i0, d0, _ = h.get_neighbors(0, n)
    c0, K0, H0, _ = h.get_fit(0, n)
    assert h.timings()["grid_points"] == n
    world = 4
    for r in range(world):
        lo, hi = n * r // world, n * (r + 1) // world
        h.set_query_range(lo, hi)
        h.curvature(50)
        t = h.timings()
        assert hi - lo < t["grid_points"] < 0.6 * n, t            # own quarter + halo, not the cloud
        assert t["limit_retries"] == 0
        i, d, _ = h.get_neighbors(lo, hi)
        c, K, H, _ = h.get_fit(lo, hi)
        assert np.array_equal(i, i0[lo:hi]) and np.array_equal(d, d0[lo:hi])
        assert np.array_equal(c, c0[lo:hi]) and np.array_equal(K, K0[lo:hi]) and np.array_equal(H, H0[lo:hi])
        rows = np.array([lo, (lo + hi) // 2, hi - 1], dtype=np.int64)
        ir, dr, _ = h.get_neighbor_rows(rows)
        assert np.array_equal(ir, i0[rows]) and np.array_equal(dr, d0[rows])
    h.close()


def test_shard_whose_neighbours_lie_past_the_kept_part_is_redone(gpu):
    """Owned rows = a tight cluster smaller than k+1: the neighbours are far-away points the first pack left
    out; the sweep notices and repeats with the whole cloud."""
    capi = gpu["capi"]
    rng = np.random.default_rng(23)
    cluster = rng.normal(scale=1e-3, size=(20, 3))
    rest = rng.uniform(2.0, 3.0, size=(30_000, 3))
    pts = np.vstack([cluster, rest]).astype(np.float32)
    idx, d = oracle.knn(pts, 30)
    h = capi.Handle(0)
    h.set_points(pts)
    h.set_query_range(0, 20)
    h.knn(30, algo=capi.KNN_GRID)
    t = h.timings()
    assert t["grid_points"] == len(pts)                  # 20 kept points cannot fill a row of 30: every point is taken at once
    i, dd, _ = h.get_neighbors(0, 20)
    assert np.array_equal(i, idx[:20]) and np.array_equal(dd, d[:20])
    # enough kept points to fill the rows, but the 30th neighbour of a row lies farther than the kept box reaches:
    # 40 owned points along a line, everything else beside it
    line = np.stack([np.linspace(0, 1, 40), np.zeros(40), np.zeros(40)], 1) + rng.normal(scale=1e-4, size=(40, 3))
    pts = np.vstack([line, rng.uniform(-0.2, 1.2, size=(30_000, 3)) * [1, 1, 0] + [0, 0, 0.35]]).astype(np.float32)
    idx, d = oracle.knn(pts, 30)
    h.set_points(pts)
    h.set_query_range(0, 40)
    h.knn(30, algo=capi.KNN_GRID)
    t = h.timings()
    assert t["limit_retries"] == 1 and t["grid_points"] == len(pts)
    i, dd, _ = h.get_neighbors(0, 40)
    assert np.array_equal(i, idx[:40]) and np.array_equal(dd, d[:40])
    h.close()


def test_far_outliers_do_not_coarsen_the_cell_list(gpu):
    """A few points far from the cloud stretch the bounding box; the grid box is trimmed to mean +- 6 sigma and
    the outliers share the boundary cells.  Results stay exact, the cell edge stays that of the clean cloud."""
    pts = gpu["shapes"].torus_random(60_000, seed=4)
    pc0, _, _ = run_cloud(gpu, pts, 30, algorithm="grid")
    cell0 = pc0.last_timings["cell_size"]
    far = np.array([[90.0, -40.0, 3.0], [-150.0, 2.0, 80.0], [5.0, 70.0, -2.0]], dtype=np.float32)
    both = np.vstack([pts, far]).astype(np.float32)
    pc, K, H = run_cloud(gpu, both, 30, algorithm="grid")
    t = pc.last_timings
    assert t["cell_size"] < 1.5 * cell0, (t["cell_size"], cell0)
    ref = oracle.pipeline_batched(np.asarray(pc.points), 30)
    assert np.array_equal(pc.neighbor_indices, ref["idx"]) and np.array_equal(pc.dists, ref["dists"])
    clean = np.arange(len(both)) < len(pts)          # the outliers' own quadrics are ill-conditioned by construction
    assert_curvature(K, H, ref["K"], ref["H"], mask=clean)


def test_unresolvable_cloud_is_refused_not_run(gpu):
    """Two tight clusters very far apart cannot be separated by 2^27 cells: the all-pairs sweep that would
    follow takes hours, so the call fails with a message instead."""
    capi = gpu["capi"]
    rng = np.random.default_rng(3)
    a = rng.normal(scale=1e-4, size=(300_000, 3))
    b = rng.normal(scale=1e-4, size=(300_000, 3)) + 1000.0
    pts = np.vstack([a, b]).astype(np.float32)
    h = capi.Handle(0)
    h.set_points(pts)
    with pytest.raises(ValueError, match="cannot resolve"):
        h.knn(30, algo=capi.KNN_GRID)
    h.close()


def test_fused_entry_matches_stepwise(gpu):
    pts = gpu["shapes"].egg_carton_random(25_000, seed=9)
    a, Ka, Ha = run_cloud(gpu, pts, 50)
    b = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    Kb, Hb = b.compute_curvature_fused(50)
    assert np.array_equal(Ka, Kb) and np.array_equal(Ha, Hb)
    assert b.last_timings["knn_ms"] > 0 and b.last_timings["fit_ms"] > 0
    # The fused call writes no distance table (its fit never reads one): the distances it hands out on request are
    # derived from the neighbours' records with the sweep's own fp64 expression -- the very bits plant_kdtree stores.
    assert np.array_equal(b.neighbor_indices, a.neighbor_indices)
    assert np.array_equal(b.dists, a.dists)
    # ... with an eps ball (missing entries: index N, distance inf, per-row counts) and through a sampled-row read too
    c = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    c.plant_kdtree(50, eps=0.02)
    d = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    d.compute_curvature_fused(50, eps=0.02)
    assert np.array_equal(c.neighbor_indices, d.neighbor_indices) and np.array_equal(c.dists, d.dists)
    assert np.array_equal(c.neighbor_counts, d.neighbor_counts) and (c.neighbor_counts < 50).any()
    rows = np.array([0, 17, 24_999])
    ih, dh, _ = d._handle.get_neighbor_rows(rows)
    assert np.array_equal(ih, c.neighbor_indices[rows]) and np.array_equal(dh, c.dists[rows])
    # a float64 cloud: the derived distance is measured from the NATIVE float64 query (pct:83), as the sweep's is
    p64 = pts.astype(np.float64) * 1.000000123 + 1e-9
    e = gpu["PointCloud"](points=p64, normals=np.zeros((len(pts), 0)))
    e.plant_kdtree(50)
    f = gpu["PointCloud"](points=p64, normals=np.zeros((len(pts), 0)))
    f.compute_curvature_fused(50)
    ref = oracle.knn(p64, 50)
    assert np.array_equal(e.neighbor_indices, ref[0]) and np.array_equal(e.dists, ref[1])
    assert np.array_equal(f.neighbor_indices, ref[0]) and np.array_equal(f.dists, ref[1])


def test_host_supplied_indices_and_validation(gpu, golden):
    g = golden("g2_torus4k_k50.npz")
    capi = gpu["capi"]
    h = capi.Handle(0)
    h.set_points(g["points"])
    h.fit_indices(g["idx"])
    co, K, H, H2 = h.get_fit(0, len(g["idx"]))
    assert_curvature(K, H, g["K"], g["H"])
    rows = np.array([5, 17, 3999])
    h.fit_indices(g["idx"][rows], query=rows)
    _, K2, H2_, _ = h.get_fit(0, 3)
    assert np.array_equal(K2, K[rows]) and np.array_equal(H2_, H[rows])
    bad = g["idx"].copy()
    bad[7, 3] = len(g["points"])                             # IndexError in the reference (pct:640)
    with pytest.raises(ValueError, match="out of range"):
        h.fit_indices(bad)
    h.close()


def test_long_rows_and_the_kernels_own_index_guard(gpu, monkeypatch):
    """pct_fit_indices takes any k the reference's fit would (rows longer than the LDS staging area are walked in
    global memory), and a table entry outside the cloud -- host validation switched off -- is never dereferenced."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(3000, seed=8)
    idx, _ = oracle.knn(pts, 300)
    rows = np.arange(0, 3000, 7)
    h = capi.Handle(0)
    h.set_points(pts)
    h.fit_indices(idx[rows], query=rows)
    co, K, H, _ = h.get_fit(0, len(rows))
    _, rK, rH, _ = oracle.curvature_batched(pts, idx[rows], rows)
    assert_curvature(K, H, rK, rH)
    short = idx[rows][:, :40].copy()
    h.fit_indices(short, query=rows)
    good = h.get_fit(0, len(rows))
    short[5, 7] = 2_000_000_000
    short[9, 0] = -3
    with pytest.raises(ValueError):
        h.fit_indices(short, query=rows)                       # the reference: IndexError at pct:640
    monkeypatch.setenv("PCT_TRUST_ROWS", "1")
    h.fit_indices(short, query=rows)
    got = h.get_fit(0, len(rows))
    monkeypatch.delenv("PCT_TRUST_ROWS")
    bad = np.zeros(len(rows), bool)
    bad[[5, 9]] = True
    for a, b in zip(got, good):
        assert np.isnan(a[bad]).all() and np.array_equal(a[~bad], b[~bad])
    h.close()


def test_curvatures_from_coefficients(gpu, golden):
    g = golden("g3_egg4k_k50.npz")
    h = gpu["capi"].Handle(0)
    K, H, H2 = h.curvatures_from_coefficients(g["coefs"])
    h.close()
    assert_curvature(K, H, g["K"], g["H"])
    assert (K == g["K"]).mean() > 0.99 and (H2 == g["H2"]).mean() > 0.9
    out = gpu["PointCloud"].calculate_explicit_quadratic_curvatures(g["coefs"][0])
    ref = oracle.quadric_curvatures(g["coefs"][0])
    assert np.allclose(out, ref, rtol=1e-6)


def test_error_conventions(gpu):
    PC = gpu["PointCloud"]
    pts = gpu["shapes"].torus_random(500, seed=1)
    nan_pc = PC(points=pts.copy(), normals=np.zeros((500, 0)))
    nan_pc.points[17, 2] = np.nan       # (a NaN at construction already dies in the ctor's norm, pct:46)
    with pytest.raises(ValueError, match="Non-finite values in input points"):      # pct:274
        nan_pc.plant_kdtree(10)
    with pytest.raises(IndexError):                                                 # k+1 > N (pct:640)
        PC(points=pts[:20], normals=np.zeros((20, 0))).plant_kdtree(20)
    pc = PC(points=pts, normals=np.zeros((500, 0)))
    with pytest.raises(AttributeError):                                             # no neighbor_indices yet
        pc.fit_explicit_quadratic_surfaces_to_neighborhoods()
    pc.plant_kdtree(19)                                                             # k+1 == 20 <= N is fine
    small = PC(points=pts[:20], normals=np.zeros((20, 0)))
    small.plant_kdtree(19)                                                          # k+1 == N exactly
    i, d = oracle.knn(pts[:20], 19)
    assert np.array_equal(small.neighbor_indices, i) and np.array_equal(small.dists, d)


def test_k_neighbors_overwritten_by_plant(gpu):
    pts = gpu["shapes"].torus_random(3000, seed=3)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((3000, 0)), k_neighbors=20)
    pc.plant_kdtree(12)
    assert pc.k_neighbors == 12 and pc.neighbor_indices.shape == (3000, 12)          # pct:71


# ------------------------------------- BASELINE configs C4 / C5 at full size (sampled reference)
def _sampled_check(h, g, n):
    rows = g["rows"]
    idx, dist, cnt = h.get_neighbor_rows(rows)
    assert np.array_equal(cnt, g["count"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dists"])
    _, K, H, _ = h.get_fit(0, n, coefs=False, H2=False)
    ok = g["count"] >= 6
    assert ok.all()
    assert_curvature(K[rows], H[rows], g["K"], g["H"])
    return K, H


# ------------------------------------- the reference's OWN generator output: lattices (SURVEY 8d, C3 secondary input)
def _lattice_check(h, g, pts, k, max_redo_fraction):
    """Tie-aware contract (as G5): sorted distance rows bit-equal; index rows equal to the reference's except inside
    exact-distance ties (cKDTree's order among equal distances is arbitrary, ours is by index) or at the last slot (a
    tie with the (k+1)-th); K/H at 1e-5 on the rows whose indices are identical, and on EVERY sampled row when the
    reference's own index rows are fitted."""
    n = len(pts)
    rows = g["rows"]
    t = h.timings()
    assert t["redone_queries"] <= max_redo_fraction * n, f"{t['redone_queries']} of {n} queries left the fast sweep"
    idx, dist, _ = h.get_neighbor_rows(rows)
    assert np.array_equal(dist, g["dists"])
    same = (idx == g["idx"]).all(1)
    for i in np.where(~same)[0]:
        d = g["dists"][i]
        for j in np.where(idx[i] != g["idx"][i])[0]:
            assert (d == d[j]).sum() >= 2 or j == k - 1, (i, j)
        # whatever the order, the neighbours are at the listed distances
        rec = np.linalg.norm(pts[idx[i]].astype(np.float64) - pts[rows[i]].astype(np.float64), axis=1).astype(np.float32)
        assert np.array_equal(rec, dist[i])
    assert same.mean() > 0.9
    _, K, H, _ = h.get_fit(0, n, coefs=False, H2=False)
    assert_curvature(K[rows], H[rows], g["K"], g["H"], mask=same)
    h.fit_indices(g["idx"], query=rows)                      # identical indices in -> contract out, every sampled row
    co, K2, H2, _ = h.get_fit(0, len(rows))
    assert_curvature(K2, H2, g["K"], g["H"])
    return same.mean()


@pytest.mark.parametrize("shape", ["torus", "egg"])
def test_reference_generator_lattice_1m(gpu, golden, shape):
    """utils.py:883-914: the 1000 x 1000 lattices the reference itself generates for a 1 M-point torus / egg carton.
    Every point has symmetric partners at (nearly) equal distances; the sweep orders equal keys in place
    (order_equal_keys) instead of sending the query to the exact kernel."""
    sh = gpu["shapes"]
    g = golden("g9_torusgrid1m_k50_sample.npz" if shape == "torus" else "g9_egggrid1m_k50_sample.npz")
    pts = sh.torus_grid(1000) if shape == "torus" else sh.egg_carton_grid(1000)
    h = gpu["capi"].Handle(0)
    h.set_points(pts)
    h.set_stats(True)
    h.curvature(50, 0.0, gpu["capi"].KNN_GRID)
    frac = _lattice_check(h, g, pts, 50, 0.05)
    print(f"{shape} lattice: {100 * frac:.1f} % of the sampled index rows equal cKDTree's, redo", h.timings()["redone_queries"])
    h.close()


@pytest.mark.parametrize("variant", ["k100_two_registers", "float64_cloud", "eps_ball", "k30_single_query_path"])
def test_lattice_variants_stay_on_the_fast_sweep_and_equal_the_exhaustive_one(gpu, variant, monkeypatch):
    """order_equal_keys in every instantiation of the sweep (two list registers, float64 queries, eps bound, the
    unpaired loop) on the reference's lattice torus, 300 x 300: bit-identical to the exhaustive sweep, curvatures
    included, and almost nothing left to the exact kernel."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_grid(300)
    k, eps = 50, 0.0
    if variant == "k100_two_registers":
        k = 100
    elif variant == "float64_cloud":
        pts = gpu["shapes"].torus_grid(300, dtype=np.float64)
    elif variant == "eps_ball":
        eps = 0.045
    else:
        k = 30
        monkeypatch.setenv("PCT_NO_PAIR", "1")       # the one-query-per-trip loop (a tuning aid the library reads per call)
    h = capi.Handle(0)
    h.set_points(pts)
    h.set_stats(True)
    h.curvature(k, eps, capi.KNN_GRID)
    redone = h.timings()["redone_queries"]
    ig, dg, cg = h.get_neighbors(0, len(pts), want_count=True)
    cf, K, H, _ = h.get_fit(0, len(pts))
    h.curvature(k, eps, capi.KNN_BRUTE)
    ib, db, cb = h.get_neighbors(0, len(pts), want_count=True)
    cfb, Kb, Hb, _ = h.get_fit(0, len(pts))
    h.close()
    assert np.array_equal(ig, ib) and np.array_equal(dg, db) and np.array_equal(cg, cb)
    assert np.array_equal(cf, cfb, equal_nan=True) and np.array_equal(K, Kb, equal_nan=True) and np.array_equal(H, Hb, equal_nan=True)
    assert redone < 0.05 * len(pts), f"{redone} of {len(pts)} queries left the fast sweep"
    if eps:
        assert cg.min() < k <= cg.max()              # the eps bound binds for some rows and not for others


def test_sample_scan_egg_carton_full_lattice(gpu, golden):
    """sample_scans/egg_carton.txt, all 99 856 points (316 x 316 lattice) as the file constructor leaves them."""
    g = golden("g9_eggcarton_file_k30_sample.npz")
    pts = g["points"]
    assert pts.shape == (99856, 3) and pts.dtype == np.float32
    h = gpu["capi"].Handle(0)
    h.set_points(pts)
    h.set_stats(True)
    h.curvature(30, 0.0, gpu["capi"].KNN_GRID)
    _lattice_check(h, g, pts, 30, 0.05)
    # and bit-identical to the exhaustive sweep, all rows
    ia, da, _ = h.get_neighbors(0, len(pts))
    b = gpu["capi"].Handle(0)
    b.set_points(pts)
    b.knn(30, 0.0, gpu["capi"].KNN_BRUTE)
    ib, db, _ = b.get_neighbors(0, len(pts))
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    b.close()
    h.close()


def test_config_c4_egg_carton_5m(gpu, golden):
    """BASELINE configs[3]: egg carton 5 M points k=50 (here on one GPU; the sharding changes no value)."""
    g = golden("g7_egg5m_k50_sample.npz")
    pts = gpu["shapes"].egg_carton_random(5_000_000, seed=1234)
    h = gpu["capi"].Handle(0)
    h.set_points(pts)
    h.curvature(50)
    _sampled_check(h, g, len(pts))
    h.close()


def test_config_c5_bunny_tiled_20m_eps(gpu, golden):
    """BASELINE configs[4]: bunny.txt tiled to 20 022 479 points, k=80, hybrid eps=0.0062 query."""
    import os
    g = golden("g7_bunny20m_k80_eps_sample.npz")
    bunny = np.load(os.path.join(os.path.dirname(__file__), "golden", "bunny_xyz_f32.npy"))
    pts = gpu["shapes"].tile_cloud(bunny, 557)
    assert len(pts) == 20_022_479
    h = gpu["capi"].Handle(0)
    h.set_points(pts)
    h.curvature(80, float(g["eps"]))
    K, H = _sampled_check(h, g, len(pts))
    assert 30 <= g["count"].min() < 80 == g["count"].max()          # both branches of the hybrid query
    t = h.timings()
    print("C5 timings", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
    h.close()


def test_validate_shape_call_sequence(gpu, tmp_path):
    """The one production caller (utils.py:481-501): file ctor -> plant(100) -> study -> fit -> re-plant -> curvatures.
    The second planting does NOT influence the curvatures: they come from the k=100 fit."""
    pts = gpu["shapes"].torus_random(6000, seed=17)
    f = tmp_path / "verts.txt"
    np.savetxt(f, pts.astype(np.float64))                       # utils.py:372-374 writes the vertices with np.savetxt
    pcl = gpu["PointCloud"](str(f))                              # utils.py:481
    pcl.plant_kdtree(k_neighbors=100)                            # :484
    np.random.seed(3)
    converged = pcl.explicit_quadratic_neighbor_study()          # :487
    pcl.fit_explicit_quadratic_surfaces_to_neighborhoods()       # :495
    pcl.plant_kdtree(k_neighbors=converged)                      # :498
    K, H = pcl.calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points()   # :501
    shifted = pts.copy()
    shifted[:, 0] -= shifted[:, 0].max()
    shifted[:, 1] -= shifted[:, 1].max()
    assert np.array_equal(pcl.points, shifted)
    ref = oracle.pipeline_batched(shifted, 100)
    assert_curvature(K, H, ref["K"], ref["H"])
    np.random.seed(3)
    ref_conv, _ = oracle.neighbor_study(shifted, np.random.randint(0, 6000, 500))
    assert converged == ref_conv
    assert len(K) == 6000 and not np.isnan(K).any() and pcl.k_neighbors == converged
    out = tmp_path / "out.ply"
    pcl.export_ply_with_curvatures(str(out))                      # utils.py:538-551
    lines = out.read_text().splitlines()
    assert lines[2] == "element vertex 6000" and lines[9] == f"{shifted[0][0]} {shifted[0][1]} {shifted[0][2]} {K[0]} {H[0]}"


# ------------------------------------------------- next row N1: neighbour study
def test_neighbor_study_matches_reference_golden(gpu, golden):
    """explicit_quadratic_neighbor_study (pct:732-800) with the reference's own seeded draw (G8)."""
    g2, g8 = golden("g2_torus4k_k50.npz"), golden("g8_neighbor_study.npz")
    pc = gpu["PointCloud"](points=g2["points"], normals=np.zeros((4000, 0)))
    pc.plant_kdtree(50)
    np.random.seed(0)
    res = pc.explicit_quadratic_neighbor_study(sample_size=60)
    assert res == int(g8["result"])
    assert pc.k_neighbors == 50 and pc.neighbor_indices.shape == (4000, 50)       # planted table untouched
    pc.plant_kdtree(100)                                                          # validate_shape's order (utils.py:484-487)
    np.random.seed(0)
    assert pc.explicit_quadratic_neighbor_study(sample_size=60) == int(g8["result"])


def test_neighbor_study_curvature_table_vs_oracle(gpu):
    """K(n) for every n, including the under-determined n+1 < 6 prefixes (minimum-norm lstsq, pct:359)."""
    rng = np.random.default_rng(3)
    xy = rng.uniform(-1, 1, size=(3000, 2))
    pts = np.column_stack([xy, 0.15 * xy[:, 0] ** 2 - 0.1 * xy[:, 1] ** 2 + 0.05 * xy[:, 0] * xy[:, 1]]).astype(np.float32)
    h = gpu["capi"].Handle(0)
    h.set_points(pts)
    h.knn(40, 0.0, gpu["capi"].KNN_GRID)
    rows = np.array([5, 77, 1500, 2999])
    Kn = h.neighbor_study_curvatures(rows, 3, 40)
    h.close()
    from scipy.spatial import cKDTree
    tree = cKDTree(pts)
    for r, i in enumerate(rows):
        for n in (3, 4, 5, 6, 7, 20, 40):
            nb = pts[tree.query(pts[i], n + 1)[1]]
            ref = oracle.quadric_curvatures(oracle.quadric_fit(oracle.plane_align(nb - pts[i])))[0]
            got = Kn[r, n - 3]
            assert abs(got - ref) <= 1e-5 * max(abs(ref), 1e-2), (i, n, got, ref)


def test_neighbor_study_on_a_plane_converges_low(gpu):
    """Exact plane: |K(n+1)-K(n)| < tol everywhere, the bisection walks down to the lower bound."""
    rng = np.random.default_rng(9)
    pts = np.column_stack([rng.uniform(-1, 1, size=(5000, 2)), np.zeros(5000)]).astype(np.float32)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((5000, 0)))
    np.random.seed(1)
    res = pc.explicit_quadratic_neighbor_study(sample_size=40)
    np.random.seed(1)
    sample = np.random.randint(0, 5000, 40)
    ref, _ = oracle.neighbor_study(pts, sample)
    assert res == ref


@pytest.mark.parametrize("config,force,issued", [("c4", "", None), ("c4", "allgather", "allgather"), ("c4", "bcast", "broadcast_groups"),
                                                 ("c4", "padded", "padded_allgather"), ("c5", "padded", "padded_allgather")])
def test_distributed_step_on_one_rank(gpu, tmp_path, config, force, issued):
    """The multi-GPU code path of bench.py -- RCCL communicator behind the C ABI (no PyTorch in the process), unique id
    through the TCP rendezvous, exchange into a device buffer on the exchange stream, zero-copy hand-over, owned-range
    sweep, double-buffered exchange -- with a world of one rank, on BASELINE configs[3] (egg carton 5 M) and configs[4]
    (bunny x 557, 20 M points, k = 80, eps), checked against the sampled reference goldens.  With one rank and no
    PCT_COMM_FORCE the driver has nothing to exchange and issues no collective (the line says so); PCT_COMM_FORCE makes
    it issue the real thing inside every step: ncclAllGather straight into the gather buffer, the group of per-rank
    ncclBroadcast, and the padded ncclAllGather + compaction that unequal shards take (C5 over 8 ranks)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCT_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PCT_COMM_FORCE"):
        env.pop(v, None)
    if force:
        env["PCT_COMM_FORCE"] = force
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--config", config, "--verify",
                        "--no-extras"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 1e7 and out["stage_ms"]["knn"] > 0 and out["scaling"] == "strong"
    assert out["verified"] == {"rows_checked": 2000, "ranks_seen": 1}
    # C4 (egg carton, rows in no spatial order) runs with ownership by slab: cut, pack, records, second exchange, scatter
    # -- all of it also with one rank, whose one slab is the cloud; C5 (tile after tile) keeps index ranges
    assert out["config"]["ownership"] == ("slab" if config == "c4" else "range")
    got = out["config"]["collectives_issued"]
    if issued is None:
        assert got["allgather"] == got["padded_allgather"] == got["broadcast_groups"] == 0
        assert "no collective" in out["config"]["parallelism"]
    else:
        per_step = 2 if config == "c4" else 1                 # slab ownership: coordinates out, records back
        assert got[issued] >= 4 * per_step and sum(got[k] for k in ("allgather", "padded_allgather", "broadcast_groups")) == got[issued]
        assert "RCCL" in out["config"]["parallelism"]


def test_rccl_exchange_of_unequal_shards_in_process(gpu, monkeypatch):
    """pct_comm_* on one rank inside this process: communicator, every form of the exchange (ncclAllGather in place,
    padded ncclAllGather + compaction, group of per-rank ncclBroadcast -- with one rank the shards are trivially equal,
    so PCT_COMM_FORCE picks the form), the one-exchange-in-flight guard, device-side wait, reductions; the sharded
    driver gives what the plain handle gives, with and without a forced collective."""
    capi = gpu["capi"]
    from point_cloud_toolbox_amd.dist import RcclExchange, ShardedCurvature
    pts = gpu["shapes"].torus_random(30_001, seed=12)
    h = capi.Handle(0)
    ex = RcclExchange(h, 0, 1, addr="127.0.0.1", port=29547)
    assert ex.allreduce([3.0, -1.0], "max").tolist() == [3.0, -1.0] and ex.allreduce([2.5], "sum").tolist() == [2.5]
    ex.barrier()
    send, recv = h.device_alloc(pts.nbytes), h.device_alloc(pts.nbytes)
    h.device_upload(send, pts)
    for force, name in (("", "allgather"), ("allgather", "allgather"), ("padded", "padded_allgather"), ("bcast", "broadcast_groups")):
        if force:
            monkeypatch.setenv("PCT_COMM_FORCE", force)
        else:
            monkeypatch.delenv("PCT_COMM_FORCE", raising=False)
        before = h.comm_counters()
        h.device_upload(recv, np.zeros_like(pts))
        ticket = ex.begin(send, recv, [pts.size])
        with pytest.raises(ValueError, match="in flight"):
            ex.begin(send, recv, [pts.size])              # one completion event per handle: refused, not silently wrong
        ex.end(ticket)
        h.comm_synchronize()
        back = np.empty_like(pts)
        h.device_download(recv, back)
        assert np.array_equal(back, pts), force
        after = h.comm_counters()
        assert after[name] == before[name] + 1 and sum(after.values()) == sum(before.values()) + 1, (force, before, after)
    h.device_free(send)
    h.device_free(recv)
    ref = capi.Handle(0)
    ref.set_points(pts)
    ref.curvature(30)
    _, K0, H0, _ = ref.get_fit(0, len(pts), coefs=False, H2=False)
    for force in ("", "padded", "bcast"):
        if force:
            monkeypatch.setenv("PCT_COMM_FORCE", force)
        else:
            monkeypatch.delenv("PCT_COMM_FORCE", raising=False)
        sc = ShardedCurvature(len(pts), 30, 0, 1, handle=h, exchange=ex)
        assert sc.collective == bool(force)
        K, H = sc.step(pts)
        K2, H2 = sc.step(pts)                             # the second gather buffer
        sc.close()
        assert np.array_equal(K, K0) and np.array_equal(H, H0) and np.array_equal(K2, K0) and np.array_equal(H2, H0), force
    monkeypatch.delenv("PCT_COMM_FORCE", raising=False)
    with pytest.raises(ValueError):
        h.comm_init(0, 1, b"x" * 128)                    # one communicator per handle
    ex.close()
    ref.close()
    h.close()


def test_sharded_driver_on_four_handles_with_a_loopback_exchange(gpu):
    """What a 4-GPU job does, on one GPU: four handles, each driven by its own ShardedCurvature (rank r of 4, unequal
    shards -- n is not a multiple of 4), through the DEVICE path of the driver (send buffer, two gather buffers, cloud
    read in place, owned range, culling of the far points); the exchange is an in-process stand-in that moves the
    shards between the handles' buffers the way the all-gather does (RCCL refuses two ranks on one device; the
    communicator itself is covered above with one rank and on the driver's 8-GPU node).  Two clouds through the same
    drivers, so that both gather buffers are used.  Rows against the plain whole-cloud handle, bit for bit."""
    capi, shapes = gpu["capi"], gpu["shapes"]
    from point_cloud_toolbox_amd.dist import ShardedCurvature, shard_range
    world, n, k = 4, 120_003, 30

    class RankHandle(capi.Handle):
        def comm_synchronize(self):                   # no exchange stream here: the copies below are blocking
            self.synchronize()

    class Loopback:
        def __init__(self, hub, rank):
            self.hub, self.rank = hub, rank

        def begin(self, send, recv, counts):
            self.hub[self.rank] = (send, recv, [int(c) for c in counts])
            return self.rank

        def end(self, rank):
            _, recv, counts = self.hub[rank]
            off = 0
            for src in range(world):
                s_ptr = self.hub[src][0]
                host = np.empty(counts[src], np.float32)
                handles[src].device_download(s_ptr, host)
                handles[rank].device_upload(recv + 4 * off, host)
                off += counts[src]
            return recv

    handles = [RankHandle(0) for _ in range(world)]
    hub = {}
    drivers = [ShardedCurvature(n, k, r, world, handle=handles[r], exchange=Loopback(hub, r)) for r in range(world)]
    ref = capi.Handle(0)
    for step, seed in enumerate((31, 32)):
        parts = [shapes.torus_scan_order(n, world, r, seed=seed) for r in range(world)]
        for r in range(world):
            lo, hi = shard_range(n, r, world)
            assert len(parts[r]) == hi - lo
            drivers[r].upload_shard(parts[r])
        tickets = [drivers[r].begin_exchange(step) for r in range(world)]
        got_K, got_H = [], []
        for r in range(world):
            drivers[r].run_device(drivers[r].end_exchange(tickets[r]))
            K, H = drivers[r].download()
            got_K.append(K)
            got_H.append(H)
            assert handles[r].timings()["grid_points"] < n            # the far three quarters were left out of the cell list
        ref.set_points(np.concatenate(parts))
        ref.curvature(k, 0.0, capi.KNN_GRID)
        _, K0, H0, _ = ref.get_fit(0, n, coefs=False, H2=False)
        assert np.array_equal(np.concatenate(got_K), K0) and np.array_equal(np.concatenate(got_H), H0), step
    for d in drivers:
        d.close()
    for h in handles + [ref]:
        h.close()


def test_slab_driver_on_four_handles_with_a_loopback_exchange(gpu):
    """The same four-rank job on a cloud in NO spatial order, ownership by slab: every rank holds an index range, cuts
    the gathered cloud like the others, answers its slab, and gets the rows of its index range back out of everybody's
    records -- the driver's own sequence (gather, pass, records, second exchange, scatter), one thread per rank, the
    exchange an in-process stand-in with a barrier where the collective has one.  Two clouds, the gather of the second
    in flight while the first is answered (bench.py's pipeline).  Bit for bit the whole-cloud handle's values."""
    import threading
    capi, shapes = gpu["capi"], gpu["shapes"]
    from point_cloud_toolbox_amd.dist import ShardedCurvature, shard_range
    world, n, k = 4, 150_003, 30

    class RankHandle(capi.Handle):
        def comm_synchronize(self):
            self.synchronize()

    gate = threading.Barrier(world)

    class Loopback:
        def __init__(self, hub, rank):
            self.hub, self.rank, self.serial = hub, rank, 0

        def begin(self, send, recv, counts):
            self.serial += 1
            self.hub[(self.rank, self.serial)] = (send, recv, [int(c) for c in counts])
            return self.serial

        def end(self, serial):
            handles[self.rank].synchronize()          # (what the send buffer holds has been written)
            gate.wait()                               # every rank has begun this exchange
            _, recv, counts = self.hub[(self.rank, serial)]
            off = 0
            for src in range(world):
                host = np.empty(counts[src], np.float32)
                if counts[src]:
                    handles[self.rank].device_download(self.hub[(src, serial)][0], host)
                    handles[self.rank].device_upload(recv + 4 * off, host)
                off += counts[src]
            gate.wait()                               # nobody's send buffer changes while another rank reads it
            return recv

    handles = [RankHandle(0) for _ in range(world)]
    hub = {}
    drivers = [ShardedCurvature(n, k, r, world, handle=handles[r], exchange=Loopback(hub, r), ownership="slab") for r in range(world)]
    assert all(d.slab and d.collective for d in drivers)
    clouds = [shapes.egg_carton_random(n, seed=seed) for seed in (41, 42)]
    results, errors = {}, []

    def rank_main(r):
        try:
            d = drivers[r]
            lo, hi = shard_range(n, r, world)
            d.upload_shard(clouds[0][lo:hi])
            ticket = d.begin_exchange(0)
            for step in range(2):
                cur = d.end_exchange(ticket)
                if step == 0:
                    d.upload_shard(clouds[1][lo:hi])
                    ticket = d.begin_exchange(1)                  # in flight during the pass below
                d.run_device(cur)
                results[(r, step)] = d.download() + (handles[r].timings()["grid_points"],)
        except BaseException as e:                                # noqa: BLE001 -- reported by the main thread
            errors.append((r, repr(e)))
            gate.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    ref = capi.Handle(0)
    for step in range(2):
        ref.set_points(clouds[step])
        ref.curvature(k, 0.0, capi.KNN_GRID)
        _, K0, H0, _ = ref.get_fit(0, n, coefs=False, H2=False)
        K = np.concatenate([results[(r, step)][0] for r in range(world)])
        H = np.concatenate([results[(r, step)][1] for r in range(world)])
        assert np.array_equal(K.view(np.uint32), K0.view(np.uint32)) and np.array_equal(H.view(np.uint32), H0.view(np.uint32)), step
        assert all(results[(r, step)][2] < 0.45 * n for r in range(world))       # a quarter and its margin, not the cloud
    for d in drivers:
        d.close()
    for h in handles + [ref]:
        h.close()


@pytest.mark.parametrize("algo", ["levels", "tree"])
@pytest.mark.parametrize("kind", ["blobs", "outliers", "shell_and_core", "quantised"])
def test_chain_of_cell_lists_equals_exhaustive_sweep(gpu, kind, algo):
    """PCT_KNN_GRID_LEVELS (each pass owns what the previous one could not answer) and PCT_KNN_TREE (the hierarchical
    cell list) against the exhaustive sweep, bit for bit, neighbours and curvatures, also for a shard (the tree
    takes whole clouds: a shard asked of it goes down the chain) and with the eps bound."""
    capi = gpu["capi"]
    which = capi.KNN_GRID_LEVELS if algo == "levels" else capi.KNN_TREE
    for n, k, eps in ((30_000, 30, 0.0), (20_000, 70, 0.0), (25_000, 25, 0.05)):
        pts = _stress_cloud(kind, np.random.default_rng([len(kind), n, 7]), n)
        h = capi.Handle(0)
        h.set_points(pts)
        h.curvature(k, eps, capi.KNN_BRUTE)
        ib, db, cb = h.get_neighbors(0, n, want_count=True)
        cfb, Kb, Hb, _ = h.get_fit(0, n)
        h.curvature(k, eps, which)
        assert h.timings()["algo"] == which and (algo == "tree" or h.timings()["levels"] >= 1)
        il, dl, cl = h.get_neighbors(0, n, want_count=True)
        cfl, Kl, Hl, _ = h.get_fit(0, n)
        assert np.array_equal(ib, il) and np.array_equal(db, dl) and np.array_equal(cb, cl), (kind, n, k, eps)
        assert np.array_equal(cfb, cfl, equal_nan=True) and np.array_equal(Kb, Kl, equal_nan=True) and np.array_equal(Hb, Hl, equal_nan=True)
        lo, hi = n // 3, n // 3 + n // 4
        h.set_query_range(lo, hi)
        h.knn(k, eps=eps, algo=which)
        assert h.timings()["algo"] == capi.KNN_GRID_LEVELS
        i2, d2, c2 = h.get_neighbors(lo, hi, want_count=True)
        assert np.array_equal(i2, ib[lo:hi]) and np.array_equal(d2, db[lo:hi]) and np.array_equal(c2, cb[lo:hi])
        h.close()


def test_auto_takes_the_hierarchical_cell_list_only_where_it_pays(gpu):
    """PCT_KNN_AUTO: a scan whose density falls off like 1/r^2 (one terrestrial laser station) goes through the
    hierarchical cell list, an even surface and a volume do not -- and whatever is chosen, the values are the same."""
    capi = gpu["capi"]
    rng = np.random.default_rng(8)
    n = 400_000
    r, a = 0.01 * 100 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n)
    lidar = np.stack([r * np.cos(a), r * np.sin(a), 0.05 * np.sin(r * np.cos(a)) * np.cos(r * np.sin(a))], 1).astype(np.float32)
    clouds = {"lidar": lidar, "torus": gpu["shapes"].torus_random(n, seed=4), "blob": rng.normal(size=(n, 3)).astype(np.float32)}
    for name, pts in clouds.items():
        h = capi.Handle(0)
        h.set_points(pts)
        h.curvature(40, 0.0, capi.KNN_AUTO)
        chosen = h.timings()["algo"]
        assert chosen == (capi.KNN_TREE if name == "lidar" else capi.KNN_GRID), (name, chosen)
        rows = np.arange(0, n, 41)
        ia, da, _ = h.get_neighbor_rows(rows)
        _, Ka, Ha, _ = h.get_fit(0, n, coefs=False, H2=False)
        h.curvature(40, 0.0, capi.KNN_GRID)
        assert h.timings()["algo"] == capi.KNN_GRID
        ib, db, _ = h.get_neighbor_rows(rows)
        _, Kb, Hb, _ = h.get_fit(0, n, coefs=False, H2=False)
        assert np.array_equal(ia, ib) and np.array_equal(da, db)
        assert np.array_equal(Ka, Kb, equal_nan=True) and np.array_equal(Ha, Hb, equal_nan=True)
        h.close()


def test_hierarchical_cell_list_on_awkward_clouds(gpu, monkeypatch):
    """PCT_KNN_TREE against the exhaustive sweep where its parts are strained: clouds smaller than a work item, every
    point identical (one Morton code: the level cannot go below 0), a line (one non-empty stencil cell in three), the
    reference's theta x phi lattice (equal keys everywhere), a lone point far from everything (its own cell is empty
    at every level: found by the 10:1 density probe, where one such query scanned the whole cloud), k up to 127 and
    the eps bound, float64 clouds (native queries against float32-rounded candidates; far from the origin the rounding
    distance is no longer small against the cells and most queries end in the exact sweep); and its exact sweep alone
    (PCT_TREE_EXACT_ONLY) on one of them."""
    capi, shapes = gpu["capi"], gpu["shapes"]
    rng = np.random.default_rng(77)
    plane = np.concatenate([rng.uniform(-1, 0, (60_000, 2)), rng.uniform(0, 1, (6_000, 2))])
    two = np.stack([plane[:, 0], plane[:, 1], 0.05 * np.sin(plane[:, 0]) * np.cos(plane[:, 1])], 1).astype(np.float32)
    two[-1] = (0.9, -0.9, 0.0)                        # alone in its quadrant
    line = np.stack([np.linspace(0, 1, 5000), np.zeros(5000), np.zeros(5000)], 1).astype(np.float32)
    cases = [
        ("tiny", rng.normal(size=(7, 3)).astype(np.float32), 5, 0.0),
        ("small", rng.normal(size=(130, 3)).astype(np.float32), 60, 0.0),
        ("identical", np.ones((3000, 3), np.float32), 20, 0.0),
        ("line", line, 30, 0.0),
        ("lattice", shapes.torus_grid(150), 50, 0.0),
        ("two densities + loner", two, 50, 0.0),
        ("two densities, k=127", two, 127, 0.0),
        ("two densities, eps", two, 40, 0.02),
        ("torus, eps smaller than the spacing", shapes.torus_random(20_000, seed=3), 10, 0.004),
        ("two densities, float64", two.astype(np.float64) + rng.normal(scale=1e-9, size=two.shape), 50, 0.0),
        ("float64 far from the origin", shapes.torus_random(30_000, seed=5).astype(np.float64) * 3.0 + 4000.0, 30, 0.0),
    ]
    for name, pts, k, eps in cases:
        n = len(pts)
        h = capi.Handle(0)
        h.set_points(pts)
        h.curvature(k, eps, capi.KNN_BRUTE)
        ib, db, cb = h.get_neighbors(0, n, want_count=True)
        _, Kb, Hb, _ = h.get_fit(0, n, coefs=False, H2=False)
        h.curvature(k, eps, capi.KNN_TREE)
        assert h.timings()["algo"] == capi.KNN_TREE, name
        it, dt, ct = h.get_neighbors(0, n, want_count=True)
        _, Kt, Ht, _ = h.get_fit(0, n, coefs=False, H2=False)
        assert np.array_equal(ib, it) and np.array_equal(db, dt) and np.array_equal(cb, ct), name
        assert np.array_equal(Kb, Kt, equal_nan=True) and np.array_equal(Hb, Ht, equal_nan=True), name
        rows = np.arange(0, n, max(1, n // 50))       # the row-wise download goes through row_of
        ir, dr, _ = h.get_neighbor_rows(rows)
        assert np.array_equal(ir, ib[rows]) and np.array_equal(dr, db[rows]), name
        if name == "two densities + loner":
            monkeypatch.setenv("PCT_TREE_EXACT_ONLY", "1")
            h.knn(k, eps=eps, algo=capi.KNN_TREE)
            monkeypatch.delenv("PCT_TREE_EXACT_ONLY")
            ie, de, _ = h.get_neighbors(0, n)
            assert np.array_equal(ib, ie) and np.array_equal(db, de), name + " (exact sweep on the tree)"
        h.close()


def test_exact_sweeps_on_tiny_clouds_whose_last_batch_is_mostly_empty(gpu):
    """The scenario behind the GPU memory fault that aborted a suite run now and then for two rounds (found at last by
    `tools/fuzz_tiny.py` seed 13, case 244: n=198, k=96, every query through `k_knn_exact<2>`): the last 64-candidate
    batch of a run has lanes without a candidate, their registers hold an earlier candidate (or zeros) whose squared
    distance ties with the running (k+1)-th EXACTLY, and the tie-break looked up the public index at a position up to
    63 records past the end of the cloud.  The read never changed a value -- such lanes cannot pass -- but on clouds
    this small it left the allocation.  A cloud at the origin makes every empty lane tie (d2 = 0 = tau); the fuzz case
    is replayed as generated.  Values against the exhaustive sweep / the oracle."""
    capi = gpu["capi"]
    rng = np.random.default_rng([13, 244])
    n = int(rng.integers(2, 300)); k = int(rng.integers(1, min(127, n - 1) + 1)); kind = rng.integers(0, 4)
    assert (n, k, int(kind)) == (198, 96, 0)
    clouds = [(rng.normal(size=(n, 3)).astype(np.float32), k), (np.zeros((100, 3), np.float32), 20), (np.zeros((65, 3), np.float32), 63),
              (np.concatenate([np.zeros((40, 3)), rng.normal(size=(30, 3))]).astype(np.float32), 33)]
    for pts, kk in clouds:
        m = len(pts)
        h = capi.Handle(0)
        h.set_points(pts)
        h.knn(kk, algo=capi.KNN_BRUTE)
        ib, db, _ = h.get_neighbors(0, m)
        assert np.array_equal(db, oracle.knn(pts, kk)[1])              # (indices: exact ties are ordered by index here)
        for algo in (capi.KNN_GRID_EXACT, capi.KNN_GRID, capi.KNN_TREE):
            h.knn(kk, algo=algo)
            ig, dg, _ = h.get_neighbors(0, m)
            assert np.array_equal(ib, ig) and np.array_equal(db, dg), (m, kk, algo)
        idx, dist = h.query_points(pts[:5].astype(np.float64), kk)
        assert np.array_equal(dist[:, 1:].astype(np.float32), db[:5, :kk - 1])
        h.close()


def test_auto_remembers_an_uneven_cloud_and_forgets_when_a_different_one_follows(gpu):
    """A handle fed a stream of clouds: after the census has sent one to the hierarchical cell list, the next one of the
    same size AND the same bounding box (2 % per face: "the same scanner again") goes there directly, without a uniform
    build first; any other cloud is examined afresh."""
    capi = gpu["capi"]
    rng = np.random.default_rng(19)
    n = 300_000

    def scan(seed):
        g = np.random.default_rng(seed)
        r, a = 0.01 * 100 ** g.uniform(0, 1, n), g.uniform(0, 2 * np.pi, n)
        p = np.stack([r * np.cos(a), r * np.sin(a), np.zeros(n)], 1).astype(np.float32)
        p[:4] = [(-1, -1, 0), (1, 1, 0), (-1, 1, 0), (1, -1, 0)]      # same extent whatever the draw
        return p

    torus = gpu["shapes"].torus_random(n, seed=6)
    h = capi.Handle(0)
    seen, grid_ms = [], []
    for pts in (scan(1), scan(2), torus, torus, scan(3), scan(4)):
        h.set_points(pts)
        h.curvature(30, 0.0, capi.KNN_AUTO)
        t = h.timings()
        seen.append(t["algo"])
        grid_ms.append(t["grid_ms"])
    h.close()
    T, G = capi.KNN_TREE, capi.KNN_GRID
    assert seen == [T, T, G, G, T, T], seen
    del rng


def test_chain_window_and_histogram_agree_at_a_bin_boundary(gpu):
    """Found by tools/fuzz_gpu.py (seed 31, case 5827): a float64 cloud far from the origin, quantised by float32 into
    piles of identical points -- every pending query of the chained sweep wanted exactly the edge of a histogram bin
    boundary; floor() put them into a window that the [lo, hi) test then found empty, and a cell list was built for
    nobody (grid of INT_MIN cells, launch of no blocks).  Bit-identical to the exhaustive sweep now."""
    capi = gpu["capi"]
    _, pts, n, k, kind, eps = _tool("fuzz_gpu").make_case(31, 5827)
    assert (n, k, kind) == (36111, 25, 7) and pts.dtype == np.float64
    h = capi.Handle(0)
    h.set_points(pts)
    h.curvature(k, eps, capi.KNN_BRUTE)
    ib, db, _ = h.get_neighbors(0, n)
    _, Kb, Hb, _ = h.get_fit(0, n, coefs=False, H2=False)
    h.curvature(k, eps, capi.KNN_GRID_LEVELS)
    ig, dg, _ = h.get_neighbors(0, n)
    _, Kg, Hg, _ = h.get_fit(0, n, coefs=False, H2=False)
    h.close()
    assert np.array_equal(ib, ig) and np.array_equal(db, dg)
    assert np.array_equal(Kb, Kg, equal_nan=True) and np.array_equal(Hb, Hg, equal_nan=True)


def test_pointcloud_flow_on_a_lidar_like_scan(gpu):
    """The class surface on a cloud whose density falls off like 1/r^2 (default algorithm: the hierarchical cell list
    is taken by itself): planting, lazy neighbour download, neighbour study, separate fit, curvatures -- against the
    oracle on sampled rows."""
    rng = np.random.default_rng(18)
    n = 200_000
    r, a = 0.02 * 50 ** rng.uniform(0, 1, n), rng.uniform(0, 2 * np.pi, n)
    x, y = r * np.cos(a), r * np.sin(a)
    pts = np.stack([x, y, 0.1 * np.sin(2 * x) * np.cos(2 * y)], 1).astype(np.float32)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((n, 0)))
    pc.plant_kdtree(100)
    assert pc.last_timings["algo"] == gpu["capi"].KNN_TREE
    np.random.seed(3)
    conv = pc.explicit_quadratic_neighbor_study(sample_size=30)
    np.random.seed(3)
    ref_conv, _ = oracle.neighbor_study(pts, np.random.randint(0, n, 30))
    assert conv == ref_conv
    pc.plant_kdtree(40)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    rows = np.sort(rng.choice(n, 1500, replace=False))
    ref = oracle.pipeline_batched(pts, 40, rows=rows)
    assert np.array_equal(pc.neighbor_indices[rows], ref["idx"]) and np.array_equal(pc.dists[rows], ref["dists"])
    assert_curvature(K[rows], H[rows], ref["K"], ref["H"])
    pc.close()


def test_random_cross_check(gpu):
    """tools/fuzz_gpu.py for a fixed seed: random clouds (torus, Gaussian at random scale, lattice with ties, blobs of
    uneven density, shifted egg carton, a line with a far sub-line), random k, eps, dtype, shard -- plain grid sweep,
    chained sweep and sharded sweep against the exhaustive sweep, bit for bit incl. coefficients and curvatures.  (The
    same tool run for minutes found the two bugs fixed in the commits that mention it.)"""
    done, bad = _tool("fuzz_gpu").run(seed0=7, budget=60.0, cases=250, verbose=False)
    assert bad is None, bad
    assert done >= 50


def test_asynchronous_calls_of_a_stream_of_clouds(gpu):
    """pct_set_async: pct_curvature on a whole-cloud handle returns once its kernels are enqueued; the next call does the
    previous one's bookkeeping after its own mid-build wait, any other entry point waits first.  Same values as the plain
    handle, timings of every call, through the uniform list, the hierarchical list, the chained sweep, with eps, with two
    list registers; an error leaves the handle usable; a sharded handle and a handle with statistics stay synchronous."""
    capi, shapes = gpu["capi"], gpu["shapes"]
    rng = np.random.default_rng(77)
    even = shapes.torus_random(120_000, seed=3)
    r = 10.0 ** rng.uniform(-2.0, 0.0, size=150_000)                     # density ~ 1/r^2: AUTO takes the hierarchical list
    th = rng.uniform(0, 2 * np.pi, size=len(r))
    scan = np.stack([r * np.cos(th), r * np.sin(th), 0.02 * np.sin(8 * r)], 1).astype(np.float32)
    ref, h = capi.Handle(0), capi.Handle(0)
    h.set_async(True)
    for pts, k, eps, algo in ((even, 50, 0.0, capi.KNN_GRID), (even, 50, 0.0, capi.KNN_AUTO), (even, 80, 0.0, capi.KNN_GRID),
                              (even, 30, 0.03, capi.KNN_GRID), (scan, 40, 0.0, capi.KNN_AUTO), (scan, 40, 0.0, capi.KNN_GRID_LEVELS),
                              (scan, 40, 0.0, capi.KNN_TREE), (even.astype(np.float64), 50, 0.0, capi.KNN_GRID)):
        ref.set_points(pts)
        ref.curvature(k, eps, algo)
        c0, K0, H0, _ = ref.get_fit(0, len(pts))
        t_ref = ref.timings()
        h.set_points(pts)
        for step in range(4):                                           # back to back: each call finishes the one before it
            h.curvature(k, eps, algo)
            t = h.stage_times_done()
            if step >= 1:
                assert t.knn_ms > 0 and t.fit_ms >= 0 and t.total_ms >= t.knn_ms, (step, t.as_dict())
                assert t.algo == t_ref["algo"] and t.grid_points == t_ref["grid_points"]
        c, K, H, _ = h.get_fit(0, len(pts))                              # (waits for the pending call)
        assert np.array_equal(c, c0, equal_nan=True) and np.array_equal(K, K0, equal_nan=True) and np.array_equal(H, H0, equal_nan=True)
        t = h.timings()
        assert t["knn_ms"] > 0 and t["total_ms"] > 0 and t["redone_queries"] == 0 and t["algo"] == t_ref["algo"]
        i0, d0, _ = ref.get_neighbors(0, 1000)
        i1, d1, _ = h.get_neighbors(0, 1000)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    # an error inside an asynchronous call: reported at once, the pending call is finished, the handle goes on
    h.set_points(even)
    h.curvature(50)
    with pytest.raises(ValueError):
        h.curvature(600)
    h.curvature(50)
    _, K, _, _ = h.get_fit(0, 100, coefs=False, H2=False)
    ref.set_points(even)
    ref.curvature(50)
    _, K0, _, _ = ref.get_fit(0, 100, coefs=False, H2=False)
    assert np.array_equal(K, K0)
    # synchronous where the verdict of the sweep is needed before returning
    h.set_query_range(1000, 50_000)
    h.curvature(50)
    assert h.stage_times_done().knn_ms == h.timings()["knn_ms"] > 0
    h.set_query_range(0, len(even))
    h.set_stats(True)
    h.curvature(50)
    assert h.timings()["redone_queries"] > 0
    h.set_stats(False)
    h.set_async(False)
    h.curvature(50)
    assert h.timings()["fit_ms"] > 0
    h.close()
    ref.close()


def test_curvature_stream_gives_every_cloud_its_own_values_in_order(gpu):
    """point_cloud_toolbox_amd.stream.curvature_stream: clouds of different sizes and dtypes through two handles in turn
    (asynchronous calls, transfers under the other cloud's kernels) -- the values of a plain handle, in input order; a
    generator as input; one cloud; none."""
    capi, shapes = gpu["capi"], gpu["shapes"]
    from point_cloud_toolbox_amd.stream import curvature_stream
    clouds = [shapes.torus_random(30_000, seed=1), shapes.egg_carton_random(45_000, seed=2), shapes.torus_random(8_000, seed=3).astype(np.float64),
              shapes.fibonacci_sphere(20_000), shapes.torus_random(30_000, seed=5), shapes.egg_carton_random(5_000, seed=6), shapes.torus_random(61_000, seed=7)]
    ref = capi.Handle(0)
    want = []
    for c in clouds:
        ref.set_points(c)
        ref.curvature(40)
        _, K, H, _ = ref.get_fit(0, len(c), coefs=False, H2=False)
        want.append((K, H))
    got = list(curvature_stream((c for c in clouds), 40))
    assert len(got) == len(want)
    for (K, H), (K0, H0) in zip(got, want):
        assert np.array_equal(K, K0, equal_nan=True) and np.array_equal(H, H0, equal_nan=True)
    assert len(list(curvature_stream(clouds[:1], 40))) == 1 and list(curvature_stream([], 40)) == []
    with pytest.raises(ValueError):
        list(curvature_stream([clouds[0], np.zeros((10, 2))], 40))
    # the handles went back to the pool in the plain state: the class works as before
    pc = gpu["PointCloud"](points=clouds[0], normals=np.zeros((len(clouds[0]), 0)))
    pc.plant_kdtree(40)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    assert np.array_equal(np.asarray(K, np.float32), want[0][0], equal_nan=True)
    pc.close()
    ref.close()


def test_random_cross_check_of_slab_ownership(gpu):
    """Fixed-seed slice of tools/fuzz_slab.py: the random clouds of fuzz_gpu (ties, blobs of uneven density, outliers, far
    offsets, anisotropic boxes; float32, >= 4096 points), random k, eps and number of slabs -- every slab on one handle,
    records scattered back, K and H of every row the bits of the unsharded call.  (33 909 cases clean in a 200 s run.)"""
    done, bad = _tool("fuzz_slab").run(seed0=3, budget=30.0, cases=400, verbose=False)
    assert bad is None, bad
    assert done >= 100


def _tool(name):
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_random_cross_check_large_and_tiny(gpu):
    """Fixed-seed slices of tools/fuzz_big.py (0.1 .. 0.5 M points: fast sweep + redo against the all-exact sweep of the
    same cell list, whole cloud and one random shard) and tools/fuzz_tiny.py (2 .. 300 points: grid, chained and
    all-exact sweeps against the exhaustive one), bit for bit."""
    done, bad = _tool("fuzz_big").run(seed0=11, budget=40.0, cases=10, verbose=False, n_max_log10=5.7)
    assert bad is None, bad
    assert done >= 3
    done, bad = _tool("fuzz_tiny").run(seed0=13, budget=30.0, cases=300)
    assert bad is None, bad
    assert done >= 100


def test_random_cross_checks_with_exact_size_buffers(gpu, monkeypatch):
    """The same tools with PCT_NO_HEADROOM=1: every device buffer is allocated at exactly the size asked for and moves on
    every growing request, so that an over-read past a table or a pointer kept across a reserve cannot hide in
    pct_reserve's spare bytes (size / 8 + 256) -- which is where the tie-break over-read of round 2 hid for two rounds."""
    monkeypatch.setenv("PCT_NO_HEADROOM", "1")
    done, bad = _tool("fuzz_tiny").run(seed0=29, budget=25.0, cases=250)
    assert bad is None, bad
    assert done >= 80
    done, bad = _tool("fuzz_gpu").run(seed0=17, budget=25.0, cases=60, verbose=False)
    assert bad is None, bad
    assert done >= 15
    done, bad = _tool("fuzz_stream").run(seed0=9, budget=20.0, cases=120, verbose=False)
    assert bad is None, bad
    assert done >= 30


def test_random_stream_of_clouds_through_one_handle(gpu):
    """Fixed-seed slice of tools/fuzz_stream.py: the warm-start state a handle keeps between calls (cell-edge hint,
    speculative box, reused culling box) never changes a result -- same-size clouds that move, shrink or change
    density, owned ranges that change or stay, against the exhaustive sweep of a fresh handle."""
    done, bad = _tool("fuzz_stream").run(seed0=1, budget=40.0, cases=400, verbose=False)
    assert bad is None, bad
    assert done >= 100


def test_edge_calls(gpu):
    """Degenerate requests end in the reference's exceptions or in well-defined results, never in a crash."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(5000, seed=3)
    h = capi.Handle(0)
    h.set_points(pts)
    for bad_k in (0, 512):                                         # (rows of up to 511 neighbours are served, pct_knn_wide.hip)
        with pytest.raises(ValueError):
            h.knn(bad_k)
    h.knn(10, eps=-1.0)                                            # a non-positive eps means "no bound"
    h.set_query_range(100, 100)                                    # an empty owned range is legal
    h.curvature(10, 0.0, capi.KNN_GRID)
    assert h.get_fit(100, 100)[1].shape == (0,)
    h.set_query_range(100, 101)
    h.curvature(10, 0.0, capi.KNN_GRID)
    one = h.get_fit(100, 101)[1]
    with pytest.raises(ValueError):
        h.set_query_range(10, 5)
    h.set_query_range(0, 5000)
    h.curvature(10, 0.0, capi.KNN_GRID)
    assert h.get_fit(100, 101)[1] == one
    h.curvature(10, 1e30, capi.KNN_GRID)                           # an eps ball that holds the cloud
    assert np.array_equal(h.get_fit(100, 101)[1], one)
    h.curvature(10, 1e-30, capi.KNN_GRID)                          # ... and one that holds nothing
    _, _, cnt = h.get_neighbors(0, 5000, want_count=True)
    assert (cnt == 0).all() and np.isnan(h.get_fit(0, 5000)[1]).all()
    h.set_points(pts[:100])
    with pytest.raises(AttributeError):
        h.fit()                                                    # no table for the new cloud yet
    with pytest.raises(IndexError):
        h.knn(100)                                                 # k + 1 > N
    h.set_points(pts[:2])
    h.curvature(1, 0.0, capi.KNN_GRID)
    assert np.array_equal(h.get_neighbors(0, 2)[0].ravel(), [1, 0])
    h.set_points(np.zeros((500, 3), np.float32))                   # all points identical: ties by index
    h.curvature(20, 0.0, capi.KNN_GRID)
    assert np.array_equal(h.get_neighbors(0, 1)[0][0], np.arange(1, 21))
    with pytest.raises(ValueError):
        h.set_points(np.vstack([pts[:100], [[np.inf, 0, 0]]]).astype(np.float32))
        h.knn(5)
    h.close()
    with pytest.raises(ValueError):
        h.knn(5)                                                   # a closed handle is refused, not dereferenced


@pytest.mark.parametrize("scale", [1e-30, 1e-15, 1e18, 1e25])
def test_coordinates_whose_squares_leave_float32(gpu, scale):
    """Coordinate differences whose squares overflow or underflow float32 still give the exhaustive sweep's table,
    with and without an eps ball (the float32 pre-selection must step aside, not drop every candidate)."""
    capi = gpu["capi"]
    pts = (gpu["shapes"].torus_random(20000, seed=5).astype(np.float64) * scale).astype(np.float32)
    h = capi.Handle(0)
    h.set_points(pts)
    for k, eps in ((30, 0.0), (92, 0.07 * scale), (12, 0.01 * scale)):
        h.curvature(k, eps, capi.KNN_BRUTE)
        want = h.get_neighbors(0, 20000, want_count=True) + (h.get_fit(0, 20000)[0],)
        h.curvature(k, eps, capi.KNN_GRID)
        got = h.get_neighbors(0, 20000, want_count=True) + (h.get_fit(0, 20000)[0],)
        for w, g in zip(want, got):
            assert np.array_equal(w, g, equal_nan=True)
        if eps > 0:
            assert want[2].max() > 0
    h.close()





@pytest.mark.gpu
def test_handles_are_independent_across_threads(gpu):
    """no global mutable state -- several handles driven from different host threads at once (ctypes drops
    the GIL during a call) give exactly what each gives alone."""
    import threading
    capi, shapes = gpu["capi"], gpu["shapes"]
    jobs = [(shapes.torus_random(60_000, seed=11), 50, 0.0), (shapes.egg_carton_random(45_000, seed=12), 30, 0.0),
            (shapes.fibonacci_sphere(30_000), 80, 0.05), (shapes.torus_random(52_000, seed=13).astype(np.float64), 20, 0.0)]

    def run(job, rounds, out):
        pts, k, eps = job
        h = capi.Handle(0)
        try:
            for _ in range(rounds):
                h.set_points(pts)
                h.curvature(k, eps, capi.KNN_GRID)
                out.append(h.get_neighbors(0, len(pts), want_count=True) + h.get_fit(0, len(pts))[:3])
        except Exception as e:          # surfaced by the assert below
            out.append(e)
        finally:
            h.close()

    alone = []
    for job in jobs:
        res = []
        run(job, 1, res)
        alone.append(res[0])
    together = [[] for _ in jobs]
    threads = [threading.Thread(target=run, args=(job, 6, together[i])) for i, job in enumerate(jobs)]
    for t in threads: t.start()
    for t in threads: t.join()
    for want, got in zip(alone, together):
        assert len(got) == 6
        for res in got:
            assert not isinstance(res, Exception), res
            for w, g in zip(want, res):
                assert np.array_equal(w, g, equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tree_query_for_arbitrary_points(gpu, dtype):
    """``pc.kdtree.query(x, k)`` (pct:74; used at pct:625, 759, 844) against SciPy's tree of the float32 cloud:
    off-cloud points, cloud points themselves (distance 0 first, nothing dropped), k = 1 squeeze, k > 64, an upper
    bound, k > N padding.  Distances are float64 and must agree bit for bit; indices too (no ties in these clouds)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(3)
    pts = gpu["shapes"].torus_random(30_000, seed=21).astype(dtype)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.plant_kdtree(20)
    ref = cKDTree(pts.astype(np.float32))
    q = np.vstack([rng.normal(size=(200, 3)), pts[rng.choice(len(pts), 100, replace=False)].astype(np.float64),
                   pts[:50].astype(np.float64) + 1e-4])
    for k in (1, 5, 64, 65, 100, 128):
        d, i = pc.kdtree.query(q, k)
        rd, ri = ref.query(q, k)
        assert d.dtype == np.float64 and d.shape == rd.shape and i.shape == ri.shape
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
    d, i = pc.kdtree.query(q[3], 7)                                  # a single point: (k,) arrays
    rd, ri = ref.query(q[3], 7)
    assert d.shape == (7,) and np.array_equal(d, rd) and np.array_equal(i, ri)
    d, i = pc.kdtree.query(q[3])                                     # k = 1: scalars
    rd, ri = ref.query(q[3])
    assert np.ndim(d) == 0 and float(d) == float(rd) and int(i) == int(ri)
    d, i = pc.kdtree.query(q, 30, distance_upper_bound=0.05)
    rd, ri = ref.query(q, 30, distance_upper_bound=0.05)
    assert np.array_equal(d, rd) and np.array_equal(i, ri) and np.isinf(d).any() and (i == len(pts)).any()
    # the neighbour study's own query (pct:759-761): the point itself comes first
    d, i = pc.kdtree.query(pc.points[17], 31)
    assert i[0] == 17 and d[0] == 0.0 and np.array_equal(i[1:21], pc.neighbor_indices[17])
    small = gpu["PointCloud"](points=pts[:40], normals=np.zeros((40, 0)))
    small.plant_kdtree(5)
    d, i = small.kdtree.query(q[:4], 50)                             # k > N: padded with N / inf
    rd, ri = cKDTree(pts[:40].astype(np.float32)).query(q[:4], 50)
    assert np.array_equal(d, rd) and np.array_equal(i, ri)
    with pytest.raises(ValueError):
        pc.kdtree.query(np.array([0.0, np.nan, 0.0]), 3)


def test_float64_diagnostics_fit(gpu):
    """pct_fit_indices_f64: the unrounded solution rounds to the float32 coefficients bit for bit,
    agrees with a float64 least-squares solve of the same float32 design rows, leaves the resident results alone,
    and its float64 curvatures sit closer to the closed form of the surface than one float32 ulp of the formulas."""
    capi, shapes = gpu["capi"], gpu["shapes"]
    pts, Kt, Ht = shapes.torus_random(40_000, seed=31, with_truth=True)
    h = capi.Handle(0)
    h.set_points(pts)
    h.curvature(50, 0.0, capi.KNN_GRID)
    rows = np.arange(0, 40_000, 97, dtype=np.int64)
    idx, _, _ = h.get_neighbor_rows(rows)
    c32, K32, H32, _ = h.get_fit(0, 40_000)
    c64, K64, H64 = h.fit_indices_f64(idx, query=rows)
    assert c64.dtype == np.float64 and c64.shape == (len(rows), 6)
    assert np.array_equal(c64.astype(np.float32), c32[rows])                 # pct:359's cast, nothing else
    assert np.allclose(K64, K32[rows], rtol=2e-5, atol=1e-5) and np.allclose(H64, H32[rows], rtol=2e-5, atol=1e-5)
    again = h.get_fit(0, 40_000)                                             # resident table and results untouched
    assert np.array_equal(again[0], c32) and np.array_equal(again[1], K32)
    assert np.array_equal(h.get_neighbor_rows(rows)[0], idx)
    for r, nb in zip(rows[:60], idx[:60]):                                   # float64 lstsq on the oracle's design rows
        rot = oracle.plane_align(pts[nb] - pts[r]).astype(np.float32)
        X = np.column_stack([rot[:, 0] ** 2, rot[:, 1] ** 2, rot[:, 0] * rot[:, 1], rot[:, 0], rot[:, 1],
                             np.ones(len(rot), np.float32)]).astype(np.float64)
        sol = np.linalg.lstsq(X, rot[:, 2].astype(np.float64), rcond=None)[0]
        got = c64[np.searchsorted(rows, r)]
        assert np.allclose(got, sol, rtol=1e-7, atol=1e-9 * np.abs(sol).max())
    with pytest.raises(ValueError):
        h.fit_indices_f64(idx[:, :3] * 0 + 40_000)                           # out-of-range index
    # rows with 2..5 valid neighbours: lstsq's minimum-norm answer (pct:359), as the float32 path; fewer: NaN
    # (five, not three: three points lie IN their plane, the orientation test's dot product is rounding noise)
    c, K, H = h.fit_indices_f64(idx[:3], count=np.array([5, 50, 1], np.int32), query=rows[:3])
    few = oracle.quadric_fit(oracle.plane_align(pts[idx[0, :5]] - pts[rows[0]]))
    assert np.allclose(c[0], few, rtol=1e-5, atol=1e-6 * np.abs(few).max()) and np.isfinite(c[1]).all()
    assert np.isnan(c[2]).all() and np.isnan(K[2])
    h.close()


@pytest.mark.parametrize("offset,scale", [(3.0, 0.01), (40.0, 0.2), (0.0, 1.0)])
def test_float64_cloud_whose_float32_rounding_is_coarse(gpu, offset, scale):
    """Float64 queries against the float32-rounded tree data (pct:74, 83) where the rounding distance of a query is a
    visible fraction of its neighbour distances: the pre-selecting sweep's bounds must absorb it (Q64 variant), with
    and without an eps ball, k below and above 64 -- bit-identical to the exhaustive sweep."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(60_000, seed=77, dtype=np.float64) * scale + offset
    h = capi.Handle(0)
    h.set_points(pts)
    spacing = scale * 0.02
    for k, eps in ((50, 0.0), (80, 0.0), (30, 2.5 * spacing), (100, 6.0 * spacing)):
        h.curvature(k, eps, capi.KNN_BRUTE)
        want = h.get_neighbors(0, len(pts), want_count=True) + h.get_fit(0, len(pts))[:3]
        h.curvature(k, eps, capi.KNN_GRID)
        got = h.get_neighbors(0, len(pts), want_count=True) + h.get_fit(0, len(pts))[:3]
        for w, g in zip(want, got):
            assert np.array_equal(w, g, equal_nan=True)
        if eps:
            assert 0 < want[2].min() < k or want[2].max() == k
    idx, d = oracle.knn(pts, 20)
    h.knn(20, algo=capi.KNN_GRID)
    gi, gd, _ = h.get_neighbors(0, len(pts))
    assert np.array_equal(gd, d)
    for r in np.where((gi != idx).any(1))[0]:          # points that coincide after the float32 rounding: exact ties,
        c = np.where(gi[r] != idx[r])[0]                # ordered by index here, arbitrarily by SciPy
        assert sorted(gi[r][c]) == sorted(idx[r][c]) or (c[-1] == 19 and (d[r][c] == d[r][c[0]]).all())
    assert (gi != idx).any(1).sum() < 20
    h.close()


def test_tiny_shard_after_the_cloud_moved(gpu):
    """A handle that owns a handful of rows reuses the neighbourhood box of its previous, similar call; when the next
    cloud of the same size lies elsewhere that box holds none of it, the kept part (the owned rows alone) cannot fill
    a row of k neighbours, and the sweep must take every point instead of handing the fused fit a table with missing
    entries (tools/fuzz_stream.py found the out-of-bounds gather this used to end in)."""
    capi, shapes = gpu["capi"], gpu["shapes"]
    n, lo, hi, k = 17_294, 7639, 7653, 20
    h = capi.Handle(0)
    a = shapes.egg_carton_random(n, seed=1) * np.float32(0.017)
    b = shapes.torus_random(n, seed=2) + np.array([59.8, 43.0, 25.4], np.float32)
    for pts in (a, b, a, b):
        h.set_points(pts)
        h.set_query_range(lo, hi)
        h.curvature(k, 0.0, capi.KNN_GRID)
        got = h.get_neighbors(lo, hi) + h.get_fit(lo, hi)[:3]
        f = capi.Handle(0)
        f.set_points(pts); f.set_query_range(lo, hi); f.curvature(k, 0.0, capi.KNN_BRUTE)
        want = f.get_neighbors(lo, hi) + f.get_fit(lo, hi)[:3]
        f.close()
        for w, g in zip(want[:2] + want[3:], got[:2] + got[3:]):
            assert np.array_equal(w, g, equal_nan=True)
    h.close()


def test_fit_results_survive_a_helper_call_on_a_sharded_handle(gpu):
    """pct_voxel_downsample borrows the cell list's scratch and drops the neighbour table; the fit results of a
    sharded handle must still be addressed by cloud row afterwards (they were mistaken for row-aligned results)."""
    capi = gpu["capi"]
    pts = gpu["shapes"].torus_random(20_000, seed=9)
    h = capi.Handle(0)
    h.set_points(pts)
    h.set_query_range(5000, 9000)
    h.curvature(30, 0.0, capi.KNN_GRID)
    before = h.get_fit(6000, 7000)
    kept = h.voxel_downsample(pts, 0.05)
    assert 0 < len(kept) < len(pts)
    after = h.get_fit(6000, 7000)
    for b, a in zip(before, after):
        assert np.array_equal(b, a)
    with pytest.raises(AttributeError):
        h.get_neighbors(6000, 7000)                      # the table is gone, as documented
    with pytest.raises(ValueError):
        h.get_fit(0, 100)                                # still outside the owned range
    h.close()


def test_random_call_sequences(gpu):
    """Fixed-seed slice of tools/fuzz_api.py: random valid sequences of C-ABI calls on one handle, every result equal
    to a fresh handle's exhaustive answer for that request alone."""
    done, bad = _tool("fuzz_api").run(seed0=5, budget=40.0, cases=6000)
    assert bad is None, bad
    assert done >= 1000


def test_attributes_outlive_the_calls_that_follow(gpu):
    """Host-state regressions found by tools/fuzz_pointcloud.py: what the reference keeps as plain attributes stays
    readable here whatever is called next -- coefficients across `neighbor_indices = ...` + re-planting (Q16),
    coefficients and curvatures across close(), `dists` after a fit from a caller-supplied table."""
    pts = gpu["shapes"].torus_random(3000, seed=4)
    ref15, ref30 = oracle.pipeline_batched(pts, 15), oracle.pipeline_batched(pts, 30)
    pc = gpu["PointCloud"](points=pts, normals=np.zeros((len(pts), 0)))
    pc.plant_kdtree(15)
    pc.fit_explicit_quadratic_surfaces_to_neighborhoods()
    pc.neighbor_indices = ref15["idx"].copy()
    pc.plant_kdtree(30)                                               # utils.py:495-501: re-plant without re-fitting
    assert (np.asarray(pc.quadratic_coefficients) == ref15["coefs"]).all(1).mean() > 0.99
    K, H = pc.calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points()
    assert_curvature(K, H, ref15["K"], ref15["H"])
    assert np.array_equal(pc.neighbor_indices, ref30["idx"])

    pc.compute_pointwise_explicit_quadratic_curvature()               # k = 30 results on the device ...
    pc.close()                                                        # ... and the handle goes away
    assert (np.asarray(pc.quadratic_coefficients) == ref30["coefs"]).all(1).mean() > 0.99
    assert_curvature(pc.K_quadratic, pc.H_quadratic, ref30["K"], ref30["H"])
    K, H = pc.calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points()
    assert_curvature(K, H, ref30["K"], ref30["H"])

    pc.plant_kdtree(15)
    pc.neighbor_indices = ref15["idx"].copy()
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()        # fit from the caller's table
    assert_curvature(K, H, ref15["K"], ref15["H"])
    assert np.array_equal(pc.dists, ref15["dists"])                   # the planted distances are still there
    pc.close()


def test_random_pointcloud_call_sequences(gpu):
    """Fixed-seed slice of tools/fuzz_pointcloud.py: the class's methods and attributes in random valid order against
    the oracle."""
    done, bad = _tool("fuzz_pointcloud").run(seed0=2, budget=40.0, cases=1500)
    assert bad is None, bad
    assert done >= 300


def test_handles_give_their_memory_back(gpu):
    """Create / use / destroy: the device memory of a handle (cloud, cell list, table, results, side buffers) returns
    to the card (tools/leak_probe.py runs the longer version)."""
    import ctypes
    capi = gpu["capi"]
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    pts = gpu["shapes"].torus_random(1_000_000, seed=3)

    def once(i):
        h = capi.Handle(0)
        h.set_points(pts if i % 2 else pts[:400_000])
        h.set_query_range(0, 300_000 if i % 3 == 0 else h.n)
        h.curvature(50, 0.0, capi.KNN_GRID if i % 4 else capi.KNN_GRID_LEVELS)
        h.get_fit(0, 100); h.get_neighbors(0, 100)
        h.query_points(pts[:3].astype(np.float64), 5)
        h.voxel_downsample(pts[:50_000], 0.05)
        h.close()

    once(1)                                    # warm the allocator's own caches
    before = free_bytes()
    for i in range(30):
        once(i)
    assert before - free_bytes() < 256 * 2**20     # one handle holds about 0.7 GB here: a leak would show 30-fold
