"""Reference fixtures for the scan-preparation row N4 (build container only; TEST INFRASTRUCTURE).

Neither module imports here as a whole (convert_asc_to_ply.py converts files at import time, :101-105; utils.py needs
open3d / pyvista), so ONLY the two function definitions are compiled out of the reference files -- located by name
with ``ast``, source untouched -- and run on seeded inputs:
  downsample            /root/reference/convert_asc_to_ply.py:20-51   (needs numpy)
  estimate_curvature    /root/reference/utils.py:778-829              (needs numpy, scikit-learn)
Nothing but inputs and outputs is written to tests/golden/g11_prep.npz.
Run from the repo root:  python oracle/make_goldens_prep.py
"""
import ast
import os

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_function(path, name):
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name], (node.lineno, node.end_lineno)


def main():
    downsample, span_d = reference_function(os.path.join(REF, "convert_asc_to_ply.py"), "downsample")
    estimate_curvature, span_e = reference_function(os.path.join(REF, "utils.py"), "estimate_curvature")
    print("downsample at lines", span_d, "estimate_curvature at lines", span_e)
    rng = np.random.default_rng(20)
    out = {}
    # downsample: float64 scatter, float32 scatter, float32 lattice whose coordinates are multiples of the voxel
    # (x / voxel lands on integers: the dtype of the division decides the voxel), a list of tuples as parse_asc_file gives
    a = rng.normal(size=(6000, 3)) * 0.7
    b = (rng.normal(size=(6000, 3)) * 0.7).astype(np.float32)
    g = np.stack(np.meshgrid(np.arange(-20, 21), np.arange(-20, 21), np.arange(-2, 3), indexing="ij"), -1).reshape(-1, 3)
    c = (g * np.float32(0.05)).astype(np.float32)
    c = c[rng.permutation(len(c))]
    d = [tuple(float(v) for v in row) for row in (g[:500] * 0.05)]
    for tag, pts, vox in (("f64", a, 0.05), ("f32", b, 0.05), ("lattice_f32", c, 0.05), ("lattice_f32_v01", c, 0.1), ("tuples", d, 0.1)):
        res = downsample(pts, voxel_size=vox)
        out[f"ds_{tag}_in"] = np.array(pts)
        out[f"ds_{tag}_voxel"] = np.float64(vox)
        out[f"ds_{tag}_out"] = res
        print(tag, np.array(pts).dtype, len(pts), "->", res.shape, res.dtype)
    # estimate_curvature: float32 and float64 clouds (k = 50 and k = 5)
    import importlib.util
    spec = importlib.util.spec_from_file_location("pct_shapes", os.path.join(os.path.dirname(OUT), "..", "point-cloud-toolbox_amd", "shapes.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    for tag, pts in (("torus2k_f32", sh.torus_random(2000, seed=21)), ("torus2k_f64", sh.torus_random(2000, seed=21, dtype=np.float64)),
                     ("torus150_f32", sh.torus_random(150, seed=22))):
        res = estimate_curvature(pts)
        out[f"ec_{tag}_in"] = pts
        out[f"ec_{tag}_out"] = res
        print(tag, res.dtype, "min", res.min(), "max", res.max())
    np.savez_compressed(os.path.join(OUT, "g11_prep.npz"), **out)


if __name__ == "__main__":
    main()
