"""Scan preparation steps that share building blocks with the curvature path."""
from __future__ import annotations

import numpy as np

from . import _capi

__all__ = ["downsample", "estimate_curvature"]


def downsample(coordinates, voxel_size=0.1, device=0, return_indices=False):
    """Voxel-grid down-sampling, same signature and result as /root/reference/convert_asc_to_ply.py:20-51:
    one point per voxel ``floor(coordinate / voxel_size)`` -- the first in input order -- in order of first occurrence."""
    coordinates = np.array(coordinates)
    h = _capi.Handle(device)
    try:
        idx = h.voxel_downsample(coordinates, voxel_size)
    finally:
        h.close()
    return (coordinates[idx], idx) if return_indices else coordinates[idx]


def estimate_curvature(points, k_fraction=0.025, max_neighbors=100, device=0):
    """PCA surface variation per point, same signature as /root/reference/utils.py:778-829:
    lambda_min / (sum lambda + 1e-10) of the 3 x 3 covariance of the k nearest points (the point itself included).

    The reference's einsum subscripts (utils.py:822) actually build the k x k Gram matrix, whose smallest eigenvalue is
    zero up to round-off; this returns the estimator its docstring describes."""
    points = np.asarray(points)
    num_points = len(points)
    k = min(max(5, int(k_fraction * num_points)), max_neighbors)          # utils.py:807
    h = _capi.Handle(device)
    try:
        h.set_points(points.astype(np.float32, copy=False))
        return h.surface_variation(k)
    finally:
        h.close()
