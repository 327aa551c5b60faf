"""Host-side mirror of the PointCloud surface (no GPU compute)."""
import os

import numpy as np
import pytest


def test_constructor_surface(built):
    PC = built["PointCloud"]
    import inspect
    sig = inspect.signature(PC.__init__)
    names = list(sig.parameters)[1:9]
    # pointCloudToolbox.py:26
    assert names == ["file_path", "points", "normals", "downsample", "voxel_size", "k_neighbors",
                     "output_path", "max_points_per_voxel"]
    d = {n: sig.parameters[n].default for n in names}
    assert d == dict(file_path=None, points=None, normals=None, downsample=False, voxel_size=0,
                     k_neighbors=20, output_path='./output/', max_points_per_voxel=1)
    with pytest.raises(ValueError, match="Either file_path or points and normals must be provided"):
        PC()
    with pytest.raises(ValueError):
        PC(points=np.zeros((4, 3)))                      # normals are required too (pct:37)


def test_constructor_attributes(built):
    PC = built["PointCloud"]
    rng = np.random.default_rng(1)
    P = rng.normal(size=(50, 3))
    pc = PC(points=P, normals=np.zeros((50, 0)), k_neighbors=7)
    assert pc.points is P and pc.points.dtype == np.float64      # kept as given (pct:38)
    assert pc.num_points == 50 and pc.num_features == 3 and pc.k_neighbors == 7
    # one transposed pass instead of numpy's three (pointcloud._matrix_norms): equal up to rounding
    assert np.isclose(pc.l1_norm, np.linalg.norm(P, 1), rtol=1e-12, atol=0)
    assert np.isclose(pc.l2_norm, np.linalg.norm(P, 2), rtol=1e-12, atol=0)
    assert pc.infinity_norm == np.linalg.norm(P, np.inf)
    assert pc.random_indexes == [] and pc.output_path == './output/'


def test_file_constructor_matches_reference(built, golden, tmp_path):
    g = golden("g4_bunny4k_file_k30.npz")
    f = tmp_path / "scan.txt"
    np.savetxt(f, g["raw"])
    pc = built["PointCloud"](str(f))
    assert pc.points.dtype == np.float32 and pc.normals.shape == (4000, 0)
    assert np.array_equal(pc.points, g["points"])                # float32 max-shift of x and y (pct:56-57)
    assert pc.points[:, 0].max() == 0 and pc.points[:, 1].max() == 0
    assert pc.x_domain[1] == 0 and pc.file_path == str(f)
    with pytest.raises(AttributeError, match="downsample_point_cloud_by_grid"):
        built["PointCloud"](str(f), downsample=True)            # pct:59-60 (method is commented out)


def test_fit_before_planting_raises(built):
    pc = built["PointCloud"](points=np.zeros((10, 3), np.float32), normals=np.zeros((10, 0)))
    with pytest.raises(AttributeError):
        pc.neighbor_indices


def test_shapes_are_reproducible_and_shardable(built):
    sh = built["shapes"]
    full = sh.torus_random(3_000_000 // 2, seed=9)
    part = sh.torus_random(3_000_000 // 2, seed=9, lo=1_048_000, hi=1_049_500)
    assert np.array_equal(full[1_048_000:1_049_500], part)
    assert full.dtype == np.float32
    e = sh.egg_carton_random(10_000, seed=3)
    assert np.array_equal(e, sh.egg_carton_random(10_000, seed=3))
    assert np.array_equal(e[100:200], sh.egg_carton_random(10_000, seed=3, lo=100, hi=200))
    p, K, H = sh.torus_random(1000, seed=1, with_truth=True)
    rho = np.hypot(p[:, 0], p[:, 1])
    assert np.allclose((rho - 1) ** 2 + p[:, 2] ** 2, 1 / 9, atol=1e-6)
    s = sh.fibonacci_sphere(500)
    assert np.allclose(np.linalg.norm(s, axis=1), 1, atol=1e-6)
    t = sh.tile_cloud(np.random.default_rng(0).random((10, 3)) * 0.1, 5)
    assert t.shape == (50, 3) and t.dtype == np.float32


def test_tree_wrapper_checks_its_arguments_before_touching_the_device(built):
    """`PointCloud.kdtree` mirrors SciPy's `query` signature; what the device sweep cannot answer is refused up
    front (no silent host fallback), malformed input raises as SciPy does."""
    from point_cloud_toolbox_amd.pointcloud import _DeviceTree

    class Cloud:                          # stands in for a PointCloud: the checks run before any device call
        num_points = 10
        points = np.zeros((10, 3), np.float32)
        def _ctx(self):
            raise AssertionError("argument checks must come first")

    t = _DeviceTree(Cloud())
    assert t.n == 10 and t.m == 3 and t.data.dtype == np.float64 and t.data.shape == (10, 3)
    with pytest.raises(NotImplementedError):
        t.query(np.zeros(3), 2, eps=0.1)                  # approximate search
    with pytest.raises(NotImplementedError):
        t.query(np.zeros(3), 2, p=1)                      # another metric
    with pytest.raises(ValueError):
        t.query(np.zeros(4), 2)
    with pytest.raises(ValueError):
        t.query(np.zeros(3), 0)
    with pytest.raises(ValueError):
        t.query(np.zeros(3), 2.5)


def test_staticmethods_validate_before_touching_the_device(built):
    """pct:273-274, 351-357: the reference's own checks and messages, raised by the host wrapper (no GPU needed)."""
    PC = built["PointCloud"]
    with pytest.raises(ValueError, match="Non-finite values in input points"):
        PC.get_best_fit_plane_and_rotate(np.array([[0.0, 0, 0], [1, np.inf, 0], [0, 1, 0]]))
    with pytest.raises(ValueError, match=r"Input points must have shape \(N, 3\)"):
        PC.fit_quadratic_surface(np.zeros((4, 2)))
    with pytest.raises(ValueError, match=r"Input points must have shape \(N, 3\)"):
        PC.fit_quadratic_surface(np.zeros(3))
    with pytest.raises(ValueError, match="Input contains non-finite values."):
        PC.fit_quadratic_surface(np.array([[0.0, 0, 0], [1, np.nan, 0], [0, 1, 0]]))


def test_plane_rotation_of_malformed_blocks_raises_what_numpy_raises():
    """pct:277-280 on a block the device path does not take: np.cov of one point (or none) is NaN and np.linalg.svd
    raises LinAlgError("SVD did not converge"); a 1-D array makes np.cov a scalar, which svd refuses.  (Checked against
    NumPy 2.2 itself below: the messages are NumPy's.)  No device is touched on these paths."""
    import numpy as np
    import warnings
    from pointCloudToolbox import PointCloud
    for block in (np.ones((1, 3)), np.zeros((0, 3)), np.ones(3)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                np.linalg.svd(np.cov(block, rowvar=False))
                raise AssertionError("numpy accepted the block")
            except np.linalg.LinAlgError as e:
                want = str(e)
        try:
            PointCloud.get_best_fit_plane_and_rotate(block)
            raise AssertionError("no error")
        except np.linalg.LinAlgError as e:
            assert str(e) == want, (block.shape, str(e), want)
