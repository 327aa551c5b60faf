"""Reference goldens for ill-conditioned / rank-deficient neighbourhoods (build container only; TEST INFRASTRUCTURE).

pointCloudToolbox.py:359 solves the quadric with numpy.linalg.lstsq(rcond=None) = LAPACK gelsd: singular values below
eps * max(m, 6) * sigma_1 are cut off and the minimum-norm solution is returned.  These fixtures hold what the
UNMODIFIED reference returns where that matters:

G6b  unit neighbourhoods through the reference's staticmethods: exactly collinear points (axis-aligned and oblique),
     a line with 1e-6 noise, a planar curve, six distinct points repeated five times each, four / five points
     (under-determined), a 1 : 20 anisotropic lattice neighbourhood.
G10  scan-line clouds (point pitch 0.005 along the line, line pitch 0.1 / 0.02 across: with k = 30 every neighbourhood
     of the first lies on ONE scan line) run through the whole reference class: plane, cylinder across the lines,
     wavy surface.
Run from the repo root:  MPLBACKEND=Agg python oracle/make_goldens_degenerate.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, load_reference, run_full  # noqa: E402


def scanline_cloud(n_lines, per_line, dx, dy, surf, jitter=0.0, seed=0):
    rng = np.random.default_rng(seed)
    X, Y = np.meshgrid(np.arange(per_line) * dx, np.arange(n_lines) * dy)
    X, Y = X.ravel(), Y.ravel()
    if jitter:
        X = X + rng.normal(0, jitter, X.shape)
        Y = Y + rng.normal(0, jitter, Y.shape)
    if surf == "plane":
        Z = 0.0 * X
    elif surf == "cylinder":                         # axis along y (the direction across the scan lines), radius 2
        Z = np.sqrt(np.maximum(4.0 - (X - 1.0) ** 2, 0.0))
    else:                                            # wavy
        Z = 0.05 * np.sin(3.0 * X) * np.cos(2.0 * Y)
    return np.stack([X, Y, Z], 1).astype(np.float32)


def unit_cases(ref):
    rng = np.random.default_rng(4242)
    t = np.sort(rng.uniform(-0.05, 0.05, 30))
    t = t[np.argsort(np.abs(t))]
    u = np.array([0.48, -0.6, 0.64])
    cases = {
        "line_axis": np.column_stack([t, 0 * t, 0 * t]),
        "line_oblique": t[:, None] * u[None, :],
        "line_noise1e6": t[:, None] * u[None, :] + rng.normal(0, 1e-6, (30, 3)),
        "planar_curve": np.column_stack([t, 0 * t, 4.0 * t * t]),
        "six_points_x5": np.repeat(np.column_stack([rng.uniform(-0.05, 0.05, (6, 2)), rng.uniform(-0.002, 0.002, 6)]), 5, 0),
        "four_points": np.column_stack([rng.uniform(-0.05, 0.05, (4, 2)), rng.uniform(-0.002, 0.002, 4)]),
        "five_points": np.column_stack([rng.uniform(-0.05, 0.05, (5, 2)), rng.uniform(-0.002, 0.002, 5)]),
        "line_axis_f32": np.column_stack([t, 0 * t, 0 * t]).astype(np.float32),
    }
    lat = scanline_cloud(9, 61, 0.005, 0.1, "wavy")
    c = lat[4 * 61 + 30]
    d = np.linalg.norm(lat - c, axis=1)
    cases["aniso_lattice_1to20"] = (lat[np.argsort(d, kind="stable")[1:31]] - c).astype(np.float32)
    out = {}
    for name, nb in cases.items():
        rot = ref.PointCloud.get_best_fit_plane_and_rotate(nb)
        cf = ref.PointCloud.fit_quadratic_surface(rot)
        cur = ref.PointCloud.calculate_explicit_quadratic_curvatures(cf)
        out[name + "_in"] = nb
        out[name + "_rot"] = rot
        out[name + "_coefs"] = np.asarray(cf)
        out[name + "_curv"] = np.array(cur, dtype=np.float32)
    return out


def main():
    ref = load_reference()
    np.savez_compressed(os.path.join(OUT, "g6b_degenerate_unit_cases.npz"), **unit_cases(ref))
    for tag, dy, surf in [("plane_1to20", 0.1, "plane"), ("cyl_1to20", 0.1, "cylinder"), ("wavy_1to20", 0.1, "wavy"),
                          ("wavy_1to4", 0.02, "wavy"), ("wavy_1to4_jitter", 0.02, "wavy")]:
        P = scanline_cloud(20, 400, 0.005, dy, surf, jitter=2e-4 if "jitter" in tag else 0.0)
        g = run_full(ref, 30, P)
        np.savez_compressed(os.path.join(OUT, f"g10_scanline_{tag}_k30.npz"), **g)
        print(tag, "K range", float(np.nanmin(g["K"])), float(np.nanmax(g["K"])), flush=True)


if __name__ == "__main__":
    main()
