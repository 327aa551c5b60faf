"""Scan preparation steps that share building blocks with the curvature path."""
from __future__ import annotations

import numpy as np

from . import _capi

__all__ = ["downsample", "estimate_curvature"]


def downsample(coordinates, voxel_size=0.1, device=0, return_indices=False):
    """Voxel-grid down-sampling, same signature and result as /root/reference/convert_asc_to_ply.py:20-51:
    one point per voxel ``floor(coordinate / voxel_size)`` -- evaluated in the array's dtype, as NumPy does: float32
    coordinates are binned in float32 -- the first in input order, in order of first occurrence.  Pinned by
    tests/golden/g11_prep.npz (outputs of the reference's own function body)."""
    coordinates = np.array(coordinates)
    h = _capi.Handle(device)
    try:
        idx = h.voxel_downsample(coordinates, voxel_size)
    finally:
        h.close()
    return (coordinates[idx], idx) if return_indices else coordinates[idx]


def estimate_curvature(points, k_fraction=0.025, max_neighbors=100, device=0, as_written=True):
    """/root/reference/utils.py:778-829, same signature.

    ``as_written=True`` (default): what the reference's code returns.  Its einsum subscripts ``'nik,njk->nij'``
    (utils.py:822) contract the COORDINATE axis, so the "covariance" is the k x k Gram matrix of the centred
    neighbourhood: rank <= 3 with k >= 5, smallest eigenvalue exactly zero.  ``eigenvalues[:, 0] / (sums + 1e-10)`` is
    therefore 0 -- the reference prints LAPACK's round-off around it (|value| < 1e-7 in float32, < 1e-15 in float64,
    either sign: tests/golden/g11_prep.npz); this returns the exact zeros, in the dtype NumPy would give.
    ``as_written=False``: the estimator the docstring and comments of the reference describe -- smallest eigenvalue of
    the 3 x 3 covariance of the k nearest points (the point itself included) over the eigenvalue sum -- computed on the
    device from the sweep's neighbour table."""
    points = np.asarray(points)
    num_points = len(points)
    k = min(max(5, int(k_fraction * num_points)), max_neighbors)          # utils.py:807
    if k > num_points:                                                    # sklearn's kneighbors says the same
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {num_points}, "
                         f"n_samples = {num_points}")
    if as_written:
        return np.zeros(num_points, dtype=np.result_type(points.dtype, np.float32))
    h = _capi.Handle(device)
    try:
        h.set_points(points.astype(np.float32, copy=False))
        return h.surface_variation(k)
    finally:
        h.close()
