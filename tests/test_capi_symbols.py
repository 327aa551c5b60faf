"""The C-ABI library loads and exports every symbol include/pct_hip.h declares (no compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pct_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pct_[a-z0-9_]+)\s*\(", text)))


def test_header_lists_the_boundary():
    syms = declared_symbols()
    for must in ("pct_create", "pct_set_points_f32", "pct_knn", "pct_fit", "pct_curvature",
                 "pct_get_neighbors", "pct_get_fit", "pct_get_timings", "pct_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built["capi"].LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in pct_hip.h but not exported"


def test_binding_table_matches_header(built):
    assert sorted(built["capi"].SIGNATURES) == declared_symbols()


def test_timings_struct_layout(built):
    # pct_timings: 7 floats, 2 int32 (+ 4 bytes of padding to 8-byte alignment), 7 int64, 1 double, 1 int64,
    # 2 int32, 1 double, 1 int64, 2 int32  (include/pct_hip.h) -- and what the library itself was compiled with
    assert ctypes.sizeof(built["capi"].Timings) == 7 * 4 + 2 * 4 + 4 + 7 * 8 + 8 + 8 + 2 * 4 + 8 + 8 + 2 * 4
    assert built["capi"].load().pct_timings_size() == ctypes.sizeof(built["capi"].Timings)


def test_no_cpu_fallback_without_device(built):
    capi = built["capi"]
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.HipExtensionError):
        capi.Handle(0)
    pc = built["PointCloud"](points=np.random.rand(64, 3).astype(np.float32), normals=np.zeros((64, 0)))
    with pytest.raises(capi.HipExtensionError):
        pc.plant_kdtree(5)


def test_missing_library_fails_loudly(built, monkeypatch):
    capi = built["capi"]
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(capi.HipExtensionError, match="no CPU fallback"):
        capi.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "point-cloud-toolbox_amd")
    for fn in os.listdir(pkg) + ["../pointCloudToolbox.py"]:
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "pct_oracle" not in src and "oracle" not in src.replace("the oracle", ""), fn
            assert "cKDTree" not in src, fn


def test_product_runs_without_pytorch():
    """north_star: no PyTorch on the path -- the multi-GPU exchange is RCCL behind the C ABI (pct_comm_*); torchrun may
    start the processes, nothing imports torch."""
    pkg = os.path.join(ROOT, "point-cloud-toolbox_amd")
    for path in [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py")] + [os.path.join(ROOT, "bench.py"),
                                                                                      os.path.join(ROOT, "pointCloudToolbox.py")]:
        src = open(path).read()
        assert not re.search(r"^\s*(import torch|from torch)", src, flags=re.M), path
    comm = open(os.path.join(pkg, "csrc", "pct_comm.hip")).read()
    for sym in ("ncclCommInitRank", "ncclAllGather", "ncclAllReduce", "ncclGetUniqueId"):
        assert sym in comm
    assert "pct_comm_allgather_f32" in declared_symbols()
