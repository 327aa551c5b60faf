"""End-to-end time of the drop-in class on a 1 M-point cloud, stage by stage (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from pointCloudToolbox import PointCloud
from point_cloud_toolbox_amd import shapes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts = shapes.torus_random(n, seed=1)
for rep in range(3):
    t0 = time.perf_counter()
    pc = PointCloud(points=pts, normals=np.zeros((n, 0)))
    t1 = time.perf_counter()
    pc.plant_kdtree(50)
    t2 = time.perf_counter()
    pc.fit_explicit_quadratic_surfaces_to_neighborhoods()
    t3 = time.perf_counter()
    K, H = pc.calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points()
    K = np.asarray(K); H = np.asarray(H)
    t4 = time.perf_counter()
    idx = pc.neighbor_indices
    t5 = time.perf_counter()
    print(f"rep {rep}: ctor {1e3*(t1-t0):.1f} ms  plant {1e3*(t2-t1):.1f}  fit {1e3*(t3-t2):.1f}  curvatures+download {1e3*(t4-t3):.1f}  neighbor_indices download {1e3*(t5-t4):.1f}  | K[0]={K[0]:.4f}", flush=True)
    t0 = time.perf_counter()
    pc2 = PointCloud(points=pts, normals=np.zeros((n, 0)))
    K2, H2 = pc2.compute_curvature_fused(50) if hasattr(pc2, "compute_curvature_fused") else (None, None)
    K2 = np.asarray(K2)
    t1 = time.perf_counter()
    print(f"        fused: ctor+compute+download {1e3*(t1-t0):.1f} ms", flush=True)
