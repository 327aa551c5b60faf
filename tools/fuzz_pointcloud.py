"""Random call sequences on the PointCloud class (developer tool): the reference's methods and attributes in any valid
order -- re-planting, lazy downloads, caller-supplied neighbour tables and coefficients, separate and combined fit /
curvature calls, tree queries, closing and reopening the device handle -- against the oracle for the same cloud.
python tools/fuzz_pointcloud.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as ge
ge.build()
from pointCloudToolbox import PointCloud
from point_cloud_toolbox_amd import shapes
import pct_oracle as oracle


def run(seed0=0, budget=None, cases=None, verbose=False):
    t_end = time.time() + (budget if budget is not None else 1e9)
    rng = np.random.default_rng([seed0, 31337])
    it = 0
    pc = None
    log = []
    def tol_ok(x, r):
        return oracle.curvature_tolerance_ok(np.asarray(x), r, 1e-2 * np.abs(r).max()).all()
    while time.time() < t_end and (cases is None or it < cases):
        if pc is None or rng.random() < 0.03:
            if pc is not None: pc.close()
            n = int(rng.integers(400, 5000))
            pts = (shapes.torus_random if rng.random() < 0.5 else shapes.egg_carton_random)(n, seed=int(rng.integers(1 << 30)))
            if rng.random() < 0.2: pts = pts.astype(np.float64)
            pc = PointCloud(points=pts, normals=np.zeros((n, 0)))
            refs = {}
            stale = None                     # the cloud the table in force was planted on, once self.points moved on
            planted = fitted = None          # planted: k of the table in force; fitted: k the coefficients came from
            curv = False
            log = ["new"]
        def ref(k):
            if k not in refs:
                refs.clear()
                if stale is None:
                    refs[k] = oracle.pipeline_batched(pts, k)
                else:                        # table of the old cloud, coordinates of the new one
                    idx, dists = oracle.knn(stale, k)
                    coefs, K, H, H2 = oracle.curvature_batched(pts, idx, np.arange(n), None)
                    refs[k] = dict(idx=idx, dists=dists, coefs=coefs, K=K, H=H, H2=H2)
            return refs[k]
        ops = ["plant", "plant", "assign_points", "edit_points"]
        if planted: ops += ["idx", "dists", "fit", "compute", "set_idx", "close"] + (["tree"] if stale is None else [])
        # (after the cloud has changed, the reference's table of the last planting is still an attribute and a fit
        # gathers the NEW coordinates with it, pct:640: table_of = the cloud the table in force was planted on)
        if fitted: ops += ["coefs", "curv", "set_coefs"]
        if curv: ops += ["read_curv"]
        op = str(rng.choice(ops)); it += 1; log.append(op)
        if verbose: print(it, op, planted, fitted, flush=True)
        bad = None
        if op in ("assign_points", "edit_points"):
            # pct:74 / pct:640 read self.points as they are at the call: a new array assigned, or the array rewritten in
            # place, must be what the next planting and the next fit see
            if planted and stale is None: stale = pts
            if op == "assign_points":
                pts = (pts * np.float32(1.0 + 0.01 * rng.random()) + np.float32(0.001)).astype(pts.dtype)
                pc.points = pts
            else:
                pts = pts.copy() if stale is pts else pts
                if pc.points is not pts: pc.points = pts
                pts *= pts.dtype.type(1.0 + 0.01 * rng.random())       # in place
            refs.clear()
            fitted = None; curv = False      # (what was fitted before belongs to a cloud this tool no longer holds)
        elif op == "plant":
            if stale is not None: fitted = None; curv = False      # (a fit made with the old table on the new cloud: this tool's reference for it goes with the table)
            stale = None; refs.clear()
            k = min(int(rng.choice([10, 15, 30, 50, 70])), n - 1)     # (k = 6 is an exactly determined fit: its agreement with
                                                                       # the oracle's SVD solve is a matter of conditioning, not of state)
            pc.plant_kdtree(k, algorithm=str(rng.choice(["auto", "grid", "brute", "tree"])))
            planted = k
            if pc.k_neighbors != k: bad = "k_neighbors not overwritten (Q15)"
        elif op == "idx":
            if not np.array_equal(pc.neighbor_indices, ref(planted)["idx"]): bad = "neighbor_indices"
        elif op == "dists":
            if not np.array_equal(pc.dists, ref(planted)["dists"]): bad = "dists"
        elif op == "fit":
            pc.fit_explicit_quadratic_surfaces_to_neighborhoods(); fitted = planted; curv = False
        elif op == "compute":
            K, H = pc.compute_pointwise_explicit_quadratic_curvature(); fitted = planted; curv = True
            r = ref(planted)
            if not (tol_ok(K, r["K"]) and tol_ok(H, r["H"])): bad = "compute K/H"
        elif op == "set_idx":                      # the caller replaces the table (same k): later fits must use it
            pc.neighbor_indices = ref(planted)["idx"].copy()
        elif op == "tree":
            i = int(rng.integers(0, n)); kk = int(rng.integers(1, min(n, 60)))
            d, j = pc.kdtree.query(pc.points[i], kk)
            if int(np.atleast_1d(j)[0]) != i and float(np.atleast_1d(d)[0]) != 0.0: bad = "tree query does not start at the point"
        elif op == "close":
            pc.close(); planted = None           # the neighbour table is gone; coefficients and stored curvatures stay
        elif op == "coefs":
            c = np.asarray(pc.quadratic_coefficients); r = ref(fitted)
            if c.shape != (n, 6) or (c == r["coefs"]).all(1).mean() < 0.98: bad = "quadratic_coefficients"
        elif op == "curv":
            K, H = pc.calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points(); curv = True
            r = ref(fitted)
            if not (tol_ok(K, r["K"]) and tol_ok(H, r["H"])): bad = "curvatures from the fit"
        elif op == "set_coefs":
            pc.quadratic_coefficients = ref(fitted)["coefs"].copy()
        elif op == "read_curv":
            r = ref(fitted)
            if not (tol_ok(pc.K_quadratic, r["K"]) and tol_ok(pc.H_quadratic, r["H"]) and len(pc.K_H_sq_quadratic) == n): bad = "stored curvature attributes"
        if bad:
            return it, f"seed={seed0} step {it}: {bad}; last ops {log[-10:]} planted={planted} fitted={fitted}"
    if pc is not None: pc.close()
    return it, None


if __name__ == "__main__":
    n_done, bad = run(int(sys.argv[2]) if len(sys.argv) > 2 else 0, budget=float(sys.argv[1]) if len(sys.argv) > 1 else 60.0,
                      verbose=os.environ.get("FUZZ_VERBOSE") == "1")
    if bad:
        print("MISMATCH", bad, flush=True); sys.exit(1)
    print(f"PointCloud fuzz ok: {n_done} steps")
