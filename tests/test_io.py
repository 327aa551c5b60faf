"""N2: native ingest of text scans and the ASCII PLY egress (host code, no GPU)."""
import os

import numpy as np
import pytest


def test_text_loader_is_bit_identical_to_loadtxt(built, tmp_path):
    capi = built["capi"]
    rng = np.random.default_rng(0)
    cases = {
        "scan3.txt": ("%.6g", rng.normal(size=(20_000, 3)) * 0.1),                 # bunny.txt style
        "scan6.txt": ("%.18e", rng.normal(size=(5_000, 6))),                       # np.savetxt default (utils.py:372)
        "wide.txt": ("%.9f", rng.uniform(-1e4, 1e4, size=(3_000, 3))),
        "tiny.txt": ("%.17g", rng.normal(size=(7, 3)) * 1e-30),
    }
    for name, (fmt, arr) in cases.items():
        f = tmp_path / name
        np.savetxt(f, arr, fmt=fmt)
        got, ref = capi.load_text(str(f)), np.loadtxt(f)
        assert got.dtype == np.float64 and got.shape == ref.shape
        assert np.array_equal(got, ref), name                                      # every value correctly rounded
    f = tmp_path / "messy.txt"
    f.write_text("# header\n1 2 3\n\n  +4.5e0\t-6\t7.25   # trailing\n8,9,10\n")
    assert np.array_equal(capi.load_text(str(f)), np.array([[1, 2, 3], [4.5, -6, 7.25], [8, 9, 10.0]]))
    (tmp_path / "ragged.txt").write_text("1 2 3\n4 5\n")
    with pytest.raises(ValueError):
        capi.load_text(str(tmp_path / "ragged.txt"))
    with pytest.raises(ValueError):
        capi.load_text(str(tmp_path / "missing.txt"))


def test_number_formatting_matches_the_fstring(built):
    capi = built["capi"]
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.normal(size=30_000) * 10.0 ** rng.integers(-14, 22, 30_000),
                           [0.0, -0.0, 1.0, 3.0, 1e-4, 9.9999e-5, 1e16, 9.999999e15, np.inf, -np.inf, np.nan,
                            123456790000.0, 1e-5, 0.1, 1 / 3, 16777216.0, 1.17549435e-38, 1e-45]]).astype(np.float32)
    for v in vals:
        assert capi.format_float(v) == f"{v}" == repr(float(v)), float(v)     # utils.py:549: f-string of a np.float32
    for v in rng.normal(size=5000) * 10.0 ** rng.integers(-300, 300, 5000):
        assert capi.format_float(v) == repr(float(v))


def test_ply_writer_reproduces_the_reference_loop(built, tmp_path):
    capi = built["capi"]
    rng = np.random.default_rng(2)
    n = 5000
    pts = (rng.normal(size=(n, 3)) * 0.2).astype(np.float32)
    K = rng.normal(size=n).astype(np.float32)
    H = rng.normal(size=n).astype(np.float32)
    K[5] = np.nan
    out = tmp_path / "out.ply"
    capi.write_ply_ascii(str(out), pts, K, H)
    # what /root/reference/utils.py:538-551 writes (restated here as the test's expectation)
    exp = ["ply", "format ascii 1.0", f"element vertex {n}", "property float x", "property float y", "property float z",
           "property float gaussian_curvature", "property float mean_curvature", "end_header"]
    exp += [f"{pts[i][0]} {pts[i][1]} {pts[i][2]} {K[i]} {H[i]}" for i in range(n)]
    same = out.read_text() == "\n".join(exp) + "\n"
    assert same


def test_file_constructor_uses_the_native_loader(built, golden, tmp_path):
    g = golden("g4_bunny4k_file_k30.npz")
    f = tmp_path / "scan.txt"
    np.savetxt(f, g["raw"])
    pc = built["PointCloud"](str(f))
    assert np.array_equal(pc.points, g["points"])
