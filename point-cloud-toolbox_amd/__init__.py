"""MI355X-native per-point curvature path (import name ``point_cloud_toolbox_amd``).

Import through the repository-root module ``pointCloudToolbox`` (the drop-in for
the reference's module of that name) or load this directory as a package.
"""
from . import _capi, shapes  # noqa: F401
from .pointcloud import PointCloud  # noqa: F401

__all__ = ["PointCloud", "shapes"]
