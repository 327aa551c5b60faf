"""Mesh energy integrals -- the consumer of the curvature path.

Mirrors ``load_mesh_compute_energies`` (/root/reference/utils.py:702-765): bending energy
sum(mean(H^2) * area), stretching energy sum(mean(K) * area), total area, computed by a GPU reduction
(``pct_mesh_energies``) instead of the reference's per-triangle Python loop.
"""
from __future__ import annotations

import logging

import numpy as np

from . import _capi

__all__ = ["mesh_energies", "load_mesh_compute_energies"]


def mesh_energies(vertices, triangles, gaussian_curvature, mean_curvature, device=0):
    """(bending_energy, stretching_energy, total_area) of a triangle mesh with per-vertex K and H."""
    h = _capi.Handle(device)
    try:
        return h.mesh_energies(vertices, triangles, gaussian_curvature, mean_curvature)
    finally:
        h.close()


def load_mesh_compute_energies(mesh, device=0):
    """Same call as the reference's function for a PyVista-like mesh object.

    ``mesh`` needs ``points`` (V,3), ``faces`` (PyVista's flat ``[3, a, b, c, ...]`` layout or a (T,3) array) and
    ``point_data`` with 'gaussian_curvature' / 'mean_curvature' (zeros if missing, utils.py:747-751).
    """
    if mesh is None:
        logging.error("Error: Mesh conversion failed.")
        return 0, 0, 0                                               # utils.py:707-709
    verts = np.asarray(mesh.points, dtype=np.float64)
    faces = np.asarray(mesh.faces)
    if faces.ndim == 1:
        if faces.size % 4 or (faces.size and not np.all(faces[::4] == 3)):
            raise ValueError("only triangle meshes are supported")
        faces = faces.reshape(-1, 4)[:, 1:]
    if len(faces) == 0:
        logging.error("Mesh has no valid triangles.")
        return 0, 0, 0                                               # utils.py:711-713, 719-721
    pd = mesh.point_data
    if 'gaussian_curvature' in pd and 'mean_curvature' in pd:
        K, H = np.asarray(pd['gaussian_curvature']), np.asarray(pd['mean_curvature'])
    else:
        logging.warning("Curvature data missing. Setting curvatures to zero.")
        K, H = np.zeros(len(verts)), np.zeros(len(verts))
    bend, stretch, area = mesh_energies(verts, faces, K, H, device)
    if area == 0:
        logging.error("Error: Computed areas are all zero.")
        return 0, 0, 0                                               # utils.py:730-732
    return bend, stretch, area
