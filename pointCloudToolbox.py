"""Drop-in replacement module for the reference's ``pointCloudToolbox``.

The reference's callers do ``from pointCloudToolbox import *`` and use the
``PointCloud`` class (/root/reference/utils.py:9, main_scans.py:8).  This
module exposes the MI355X-backed class of the same name from the package
directory ``point-cloud-toolbox_amd/`` (not a valid Python identifier, so it is
registered under the import name ``point_cloud_toolbox_amd``).
"""
import importlib.util
import os
import sys

_PKG_NAME = "point_cloud_toolbox_amd"
_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "point-cloud-toolbox_amd")


def _load_package():
    if _PKG_NAME in sys.modules:
        return sys.modules[_PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        _PKG_NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


_load_package()

from point_cloud_toolbox_amd.pointcloud import PointCloud  # noqa: E402

__all__ = ["PointCloud"]
