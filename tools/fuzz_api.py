"""Random sequences of C-ABI calls on one handle (developer tool): whatever order a caller uses the entry points in --
new clouds, owned ranges, sweeps with any algorithm, separate or fused fits, host-supplied rows, side queries, voxel
and mesh helpers, statistics and cell-size knobs in between -- every result equals what a fresh handle computes for
that request alone with the exhaustive sweep.  Only valid calls are issued (the state rules of include/pct_hip.h are
tracked here).  python tools/fuzz_api.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from point_cloud_toolbox_amd import _capi, shapes

ALGOS = [_capi.KNN_GRID, _capi.KNN_GRID, _capi.KNN_BRUTE, _capi.KNN_GRID_EXACT, _capi.KNN_GRID_LEVELS, _capi.KNN_TREE]


def new_cloud(rng):
    n = int(rng.integers(300, 30_000))
    kind = rng.integers(0, 4)
    if kind == 0: p = shapes.torus_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64)
    elif kind == 1: p = shapes.egg_carton_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64)
    elif kind == 2: p = rng.normal(size=(n, 3)) * [1, 1, 0.05]
    else:
        p = shapes.torus_random(n, seed=int(rng.integers(1 << 30))).astype(np.float64); p[: n // 3] *= 0.25
    if rng.random() < 0.3: p = p * 10.0 ** rng.uniform(-2, 2) + rng.normal(size=3) * 10.0 ** rng.uniform(-1, 2)
    return np.ascontiguousarray(p, dtype=np.float32 if rng.random() < 0.8 else np.float64)


class Reference:
    """Fresh-handle answers for the current cloud, cached per (k, eps)."""
    def __init__(self, pts):
        self.pts, self.cache = pts, {}
    def table(self, k, eps):
        key = (k, eps)
        if key not in self.cache:
            f = _capi.Handle(0); f.set_points(self.pts); f.curvature(k, eps, _capi.KNN_BRUTE)
            n = len(self.pts)
            self.cache = {key: f.get_neighbors(0, n, want_count=True) + f.get_fit(0, n)}
            f.close()
        return self.cache[key]


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def run(seed0=0, budget=None, cases=None, verbose=False):
    t_end = time.time() + (budget if budget is not None else 1e9)
    rng = np.random.default_rng([seed0, 777])
    h = _capi.Handle(0)
    pts = ref = None
    knn = fit = None            # knn: (k, eps, lo, hi) of the resident table; fit: "cloud" | ("rows", rows idx) | None
    lo = hi = 0
    it = 0
    log = []
    def fail(msg):
        h.close()
        return it, f"seed={seed0} step {it}: {msg}; last ops: {log[-8:]}"
    while time.time() < t_end and (cases is None or it < cases):
        ops = ["cloud"] if pts is None else ["cloud", "range", "knn", "knn", "curv", "curv", "stats", "factor", "query", "voxel", "survar", "async"]
        if knn: ops += ["fit", "get_nbr", "get_rows", "rows_fit", "rows_fit64", "study"]
        if fit: ops += ["get_fit", "get_fit"]
        w = np.array([0.25 if o in ("cloud", "range") else 0.5 if o in ("stats", "factor", "voxel", "async") else 1.0 for o in ops]) if pts is not None else None
        op = str(rng.choice(ops, p=None if w is None else w / w.sum())); it += 1
        log.append(op)
        if verbose: print(it, op, knn, fit[0] if fit else None, flush=True)
        if op == "cloud":
            pts = new_cloud(rng); ref = Reference(pts); h.set_points(pts); n = len(pts); lo, hi = 0, n; knn = fit = None
        elif op == "range":
            lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
            if rng.random() < 0.3: lo, hi = 0, n
            h.set_query_range(lo, hi); knn = fit = None
        elif op in ("knn", "curv"):
            k = min(int(rng.choice([3, 10, 30, 50, 64, 100])), n - 1)
            eps = float(np.ptp(pts, axis=0).max()) * 10.0 ** rng.uniform(-2.2, -0.8) if rng.random() < 0.25 else 0.0
            algo = int(rng.choice(ALGOS))
            (h.knn if op == "knn" else h.curvature)(k, eps, algo)
            knn = (k, eps, lo, hi); fit = ("cloud", k, eps, lo, hi) if op == "curv" else None
        elif op == "fit":
            h.fit(); fit = ("cloud",) + knn
        elif op == "stats": h.set_stats(bool(rng.integers(0, 2)))
        elif op == "async": h.set_async(bool(rng.integers(0, 2)))       # fused calls return early; everything else waits for them
        elif op == "factor": h.set_grid_param(float(rng.choice([0.0, 0.0, 0.35, 0.5, 0.8])))
        elif op == "get_nbr":
            k, eps, klo, khi = knn
            b = int(rng.integers(klo, khi)); e = int(rng.integers(b, khi + 1))
            got = h.get_neighbors(b, e, want_count=True); want = ref.table(k, eps)
            if not (same(got[0], want[0][b:e]) and same(got[1], want[1][b:e]) and same(got[2], want[2][b:e])): return fail(f"get_neighbors [{b},{e}) of {knn}")
        elif op == "get_rows":
            k, eps, klo, khi = knn
            rows = rng.integers(klo, khi, size=int(rng.integers(1, 40)))
            got = h.get_neighbor_rows(rows); want = ref.table(k, eps)
            if not (same(got[0], want[0][rows]) and same(got[1], want[1][rows]) and same(got[2], want[2][rows])): return fail(f"get_neighbor_rows of {knn}")
        elif op == "get_fit":
            if fit[0] == "cloud":
                _, k, eps, klo, khi = fit
                b = int(rng.integers(klo, khi)); e = int(rng.integers(b, khi + 1))
                got = h.get_fit(b, e); want = ref.table(k, eps)[3:]
                if not all(same(g, w[b:e]) for g, w in zip(got, want)): return fail(f"get_fit [{b},{e}) of {knn}")
            else:
                rows, k, eps = fit[1], fit[2], fit[3]
                got = h.get_fit(0, len(rows)); want = ref.table(k, eps)[3:]
                if not all(same(g, w[rows]) for g, w in zip(got, want)): return fail("get_fit of host-supplied rows")
        elif op in ("rows_fit", "rows_fit64"):
            k, eps, klo, khi = knn
            rows = np.sort(rng.choice(np.arange(klo, khi), size=min(khi - klo, int(rng.integers(1, 60))), replace=False))
            idx, _, cnt = h.get_neighbor_rows(rows)
            idx = np.where(idx >= n, 0, idx)                   # padding entries lie beyond the count
            if op == "rows_fit":
                h.fit_indices(idx, count=cnt, query=rows); fit = ("rows", rows, k, eps)
            else:
                c64, K64, H64 = h.fit_indices_f64(idx, count=cnt, query=rows); want = ref.table(k, eps)
                if not same(c64.astype(np.float32), want[3][rows]): return fail("fit_indices_f64 does not round to the float32 coefficients")
        elif op == "study":
            k, eps, klo, khi = knn
            if eps == 0.0 and k >= 12:
                rows = rng.integers(klo, khi, size=5); n_hi = int(rng.integers(6, k + 1)); n_lo = int(rng.integers(3, n_hi + 1))
                got = h.neighbor_study_curvatures(rows, n_lo, n_hi)
                f = _capi.Handle(0); f.set_points(pts); f.knn(k, 0.0, _capi.KNN_BRUTE); want = f.neighbor_study_curvatures(rows, n_lo, n_hi); f.close()
                if not same(got, want): return fail("neighbor_study_curvatures")
        elif op == "query":
            q = np.vstack([rng.normal(size=(5, 3)) * np.ptp(pts, axis=0).max() + pts.mean(0), pts[rng.integers(0, n, 5)].astype(np.float64)])
            kq = int(rng.choice([1, 7, 64, 65, 128]))
            got = h.query_points(q, kq)
            f = _capi.Handle(0); f.set_points(pts); want = f.query_points(q, kq); f.close()
            if not (same(got[0], want[0]) and same(got[1], want[1])): return fail("query_points")
        elif op == "survar":                                        # PCA surface variation: plants its own table, drops the fit
            kt = min(int(rng.choice([8, 20, 40])), n - 1)
            if lo == 0 and hi == n:
                got = h.surface_variation(kt)
                f = _capi.Handle(0); f.set_points(pts); want = f.surface_variation(kt); f.close()
                if not same(got, want): return fail("surface_variation")
                knn = (kt - 1, 0.0, lo, hi); fit = None
        elif op == "voxel":
            v = float(np.ptp(pts, axis=0).max()) * 10.0 ** rng.uniform(-2, -0.5)
            got = h.voxel_downsample(pts, v)
            f = _capi.Handle(0); want = f.voxel_downsample(pts, v); f.close()
            if not same(got, want): return fail("voxel_downsample")
            knn = None                                          # the helper reuses the cell list's scratch (pct_hip.h); fit results stay
    h.close()
    return it, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 60.0,
                      verbose=os.environ.get("FUZZ_VERBOSE") == "1")
    if bad:
        print("MISMATCH", bad, flush=True); sys.exit(1)
    print(f"api fuzz ok: {n_done} steps")
