"""N4: voxel-grid down-sampling and the PCA surface-variation estimator.

Restated from convert_asc_to_ply.py:20-51 and utils.py:778-829 (neither importable here); parity unpinned by a
reference run, checked against the restatements in oracle/pct_oracle.py."""
import numpy as np
import pytest

import pct_oracle as oracle


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
    got = estimate_curvature(pts)                                   # k = min(max(5, 150), 100) = 100
    ref = oracle.surface_variation(pts)                             # the documented 3 x 3 estimator, float64
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.abs(got - ref).max() < 1e-6
    assert 0 <= got.min() and got.max() < 1 / 3 + 1e-6
    small = pts[:150]                                               # k = 5
    assert np.abs(estimate_curvature(small) - oracle.surface_variation(small)).max() < 1e-6
