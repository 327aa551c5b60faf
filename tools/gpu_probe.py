"""Developer probe run on the GPU box: parity vs the oracle + stage timings.

usage: python tools/gpu_probe.py [N] [k]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import __graft_entry__ as ge  # noqa: E402

ge.build()
import pct_oracle as oracle  # noqa: E402
from point_cloud_toolbox_amd import _capi, shapes  # noqa: E402


def check(name, pts, k, algo, eps=0.0):
    h = _capi.Handle(0)
    h.set_points(pts)
    t0 = time.time()
    h.curvature(k, eps, algo)
    t1 = time.time()
    tm = h.timings()
    n = len(pts)
    idx, dist, cnt = h.get_neighbors(0, n, True, True, True)
    coefs, K, H, H2 = h.get_fit(0, n)
    ref = oracle.pipeline_batched(pts, k, eps=eps if eps > 0 else None)
    same_idx = (idx == ref["idx"]).all(1)
    same_d = (dist == ref["dists"]).all(1)
    fK = 1e-2 * np.nanmax(np.abs(ref["K"]))
    fH = 1e-2 * np.nanmax(np.abs(ref["H"]))
    okK = oracle.curvature_tolerance_ok(K, ref["K"], fK) | (np.isnan(K) & np.isnan(ref["K"]))
    okH = oracle.curvature_tolerance_ok(H, ref["H"], fH) | (np.isnan(H) & np.isnan(ref["H"]))
    print(f"[{name}] N={n} k={k} algo={algo} eps={eps}: idx rows equal {same_idx.mean():.6f} dist rows equal {same_d.mean():.6f} "
          f"K ok {okK.mean():.6f} H ok {okH.mean():.6f} coefs exact {(coefs == ref['coefs']).all(1).mean():.4f} "
          f"K exact {(K == ref['K']).mean():.4f} wall {1e3 * (t1 - t0):.2f} ms")
    print("    timings:", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in tm.items()})
    if eps > 0:
        print("    count equal:", (cnt == ref["count"]).mean(), "min/max count", cnt.min(), cnt.max())
    bad = np.where(~same_idx)[0]
    if len(bad):
        i = bad[0]
        print("    first bad row", i, "\n     gpu", idx[i], "\n     ref", ref["idx"][i], "\n     gpu d", dist[i][:8], "\n     ref d", ref["dists"][i][:8])
    h.close()
    return same_idx.all() and same_d.all() and okK.all() and okH.all()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    ok = True
    pts = shapes.torus_random(n, seed=3)
    ok &= check("torus/brute", pts[:5000].copy(), k, _capi.KNN_BRUTE)
    ok &= check("torus/grid", pts, k, _capi.KNN_GRID)
    ok &= check("sphere/grid", shapes.fibonacci_sphere(n), 30, _capi.KNN_GRID)
    ok &= check("egg/grid k=80", shapes.egg_carton_random(n, seed=4), 80, _capi.KNN_GRID)
    ok &= check("egg/grid eps", shapes.egg_carton_random(n, seed=4), 50, _capi.KNN_GRID, eps=0.06)
    ok &= check("torus f64/grid", shapes.torus_random(n, seed=3, dtype=np.float64), k, _capi.KNN_GRID)
    print("ALL OK" if ok else "MISMATCH")
    # timing at scale
    for big in (1_000_000,):
        pts = shapes.torus_random(big, seed=1234)
        h = _capi.Handle(0)
        h.set_points(pts)
        for it in range(4):
            t0 = time.time()
            h.curvature(50, 0.0, _capi.KNN_GRID)
            t1 = time.time()
            tm = h.timings()
            print(f"N={big} it={it} wall {1e3 * (t1 - t0):.2f} ms -> {big / (t1 - t0) / 1e6:.1f} Mpts/s ",
                  {a: (round(b, 4) if isinstance(b, float) else b) for a, b in tm.items()})
        h.close()


if __name__ == "__main__":
    main()
