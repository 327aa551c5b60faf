"""Randomised cross-check of ownership by slab (pct_set_query_slab) against the unsharded handle (developer tool).
Random clouds of tools/fuzz_gpu.py (float32 ones of >= 4096 points), random k / eps / number of parts: every part on one
handle, the records scattered back -- K and H of every row must be the bits of the plain pct_curvature call.
python tools/fuzz_slab.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi
import fuzz_gpu


def run(seed0, budget=None, cases=None, verbose=True):
    t_end = time.time() + (budget if budget is not None else 1e9)
    it = done = 0
    h = _capi.Handle(0)
    while time.time() < t_end and (cases is None or done < cases):
        rng, pts, n, k, kind, eps = fuzz_gpu.make_case(seed0, it)
        it += 1
        if n < 4096:
            continue
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        if not np.isfinite(pts).all():
            continue
        parts = int(rng.integers(1, 10))
        if verbose: print(f"case {it - 1}: n={n} k={k} kind={kind} eps={eps:.4g} parts={parts}", flush=True)
        h.set_points(pts)
        h.curvature(k, eps)
        _, K0, H0, _ = h.get_fit(0, n, coefs=False, H2=False)
        rec, Kd, Hd = h.device_alloc(n * 12), h.device_alloc(n * 4), h.device_alloc(n * 4)
        try:
            off = 0
            for part in range(parts):
                h.set_query_slab(part, parts)
                h.curvature(k, eps)
                counts = h.slab_counts(parts)
                rows = h.slab_records(rec + off * 12, n - off)
                if rows != counts[part] or sum(counts) != n:
                    return done, f"counts seed=({seed0},{it - 1}) part {part}/{parts}: rows {rows}, counts {counts}"
                off += rows
            h.scatter_records(rec, n, 0, n, Kd, Hd)
            K, H = np.empty(n, np.float32), np.empty(n, np.float32)
            h.device_download(Kd, K)
            h.device_download(Hd, H)
        finally:
            for p_ in (rec, Kd, Hd):
                h.device_free(p_)
        if not (np.array_equal(K.view(np.uint32), K0.view(np.uint32)) and np.array_equal(H.view(np.uint32), H0.view(np.uint32))):
            bad = np.flatnonzero((K.view(np.uint32) != K0.view(np.uint32)) | (H.view(np.uint32) != H0.view(np.uint32)))
            return done, f"values seed=({seed0},{it - 1}) n={n} k={k} kind={kind} eps={eps} parts={parts}: {len(bad)} rows differ, first {bad[:5]}"
        done += 1
    h.close()
    return done, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
    print(f"done: {n_done} cases, {'no mismatch' if bad is None else 'MISMATCH ' + bad}")
    sys.exit(0 if bad is None else 1)
