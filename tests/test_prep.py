"""N4: voxel-grid down-sampling and the PCA surface-variation estimator.

Pinned: tests/golden/g11_prep.npz holds inputs and outputs of the reference's own two function bodies
(convert_asc_to_ply.py:20-51, utils.py:778-829), compiled out of the reference files by oracle/make_goldens_prep.py;
the restatements in oracle/pct_oracle.py are checked against it here, the device path against both."""
import numpy as np
import pytest

import pct_oracle as oracle
import pointCloudToolbox  # noqa: F401  (registers point_cloud_toolbox_amd)


DS_CASES = ["f64", "f32", "lattice_f32", "lattice_f32_v01", "tuples"]


@pytest.mark.parametrize("tag", DS_CASES)
def test_oracle_downsample_equals_the_reference(golden, tag):
    g = golden("g11_prep.npz")
    out = oracle.voxel_downsample(g[f"ds_{tag}_in"], float(g[f"ds_{tag}_voxel"]))
    assert out.dtype == g[f"ds_{tag}_out"].dtype and np.array_equal(out, g[f"ds_{tag}_out"])


@pytest.mark.parametrize("tag", ["torus2k_f32", "torus2k_f64", "torus150_f32"])
def test_oracle_surface_variation_as_written_equals_the_reference(golden, tag):
    g = golden("g11_prep.npz")
    ref = g[f"ec_{tag}_out"]
    out = oracle.surface_variation(g[f"ec_{tag}_in"], as_written=True)
    assert out.dtype == ref.dtype
    tiny = 1e-6 if ref.dtype == np.float32 else 1e-14
    assert np.abs(ref).max() < tiny and np.abs(out).max() < tiny      # round-off around an exactly-zero eigenvalue


@pytest.mark.gpu
@pytest.mark.parametrize("tag", DS_CASES)
def test_gpu_downsample_equals_the_reference(gpu, golden, tag):
    from point_cloud_toolbox_amd.prep import downsample
    g = golden("g11_prep.npz")
    pts = g[f"ds_{tag}_in"]
    if tag == "tuples":
        pts = [tuple(r) for r in pts]                                # what parse_asc_file hands over
    got = downsample(pts, float(g[f"ds_{tag}_voxel"]))
    assert got.dtype == g[f"ds_{tag}_out"].dtype and np.array_equal(got, g[f"ds_{tag}_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["torus2k_f32", "torus2k_f64", "torus150_f32"])
def test_gpu_estimate_curvature_default_follows_the_code_as_written(gpu, golden, tag):
    from point_cloud_toolbox_amd.prep import estimate_curvature
    g = golden("g11_prep.npz")
    ref = g[f"ec_{tag}_out"]
    got = estimate_curvature(g[f"ec_{tag}_in"])
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.abs(got - ref).max() < (1e-6 if ref.dtype == np.float32 else 1e-14)
    with pytest.raises(ValueError):
        estimate_curvature(g[f"ec_{tag}_in"][:4])                    # k = 5 > 4 points: sklearn refuses too


def test_estimate_curvature_default_needs_no_device(golden):
    """The code as written returns round-off around an exactly-zero eigenvalue: the default answer is that zero, in the
    dtype NumPy would give, for any cloud the reference's function accepts."""
    from point_cloud_toolbox_amd.prep import estimate_curvature
    g = golden("g11_prep.npz")
    for tag in ("torus2k_f32", "torus2k_f64", "torus150_f32"):
        out = estimate_curvature(g[f"ec_{tag}_in"])
        assert out.dtype == g[f"ec_{tag}_out"].dtype and out.shape == g[f"ec_{tag}_out"].shape and not out.any()


def test_oracle_downsample_semantics():
    pts = np.array([[0.05, 0.05, 0.05], [0.06, 0.01, 0.09], [0.15, 0.0, 0.0], [-0.01, 0.0, 0.0], [0.19, 0.09, 0.01]])
    out = oracle.voxel_downsample(pts, 0.1)
    assert np.array_equal(out, pts[[0, 2, 3]])                     # first point of each voxel, in order of first occurrence


@pytest.mark.gpu
@pytest.mark.parametrize("n,voxel", [(50_000, 0.05), (200_000, 0.013), (1000, 1e-4)])
def test_gpu_downsample_matches_restatement(gpu, n, voxel):
    from point_cloud_toolbox_amd.prep import downsample
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(n, 3)) * 0.7                             # float64, negative coordinates included
    if n == 200_000:
        pts = pts.astype(np.float32)                                # binned in float32, as NumPy would
    got, idx = downsample(pts, voxel, return_indices=True)
    ref = oracle.voxel_downsample(pts, voxel)
    assert np.array_equal(got, ref) and (np.diff(idx) > 0).all()


def test_reference_surface_variation_quirk():
    """utils.py:822 contracts the coordinate axis: k x k Gram matrix, smallest eigenvalue == 0 up to round-off."""
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(400, 3)).astype(np.float32)
    written = oracle.surface_variation(pts, as_written=True)          # k = 10
    intended = oracle.surface_variation(pts)
    assert np.abs(written).max() < 1e-5 and intended.min() > 1e-3


@pytest.mark.gpu
def test_gpu_surface_variation_matches_restatement(gpu):
    from point_cloud_toolbox_amd.prep import estimate_curvature
    pts = gpu["shapes"].torus_random(6000, seed=5)
    got = estimate_curvature(pts, as_written=False)                 # k = min(max(5, 150), 100) = 100
    ref = oracle.surface_variation(pts)                             # the documented 3 x 3 estimator, float64
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.abs(got - ref).max() < 1e-6
    assert 0 <= got.min() and got.max() < 1 / 3 + 1e-6
    small = pts[:150]                                               # k = 5
    assert np.abs(estimate_curvature(small, as_written=False) - oracle.surface_variation(small)).max() < 1e-6
