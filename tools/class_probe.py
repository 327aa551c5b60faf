"""Developer tool: wall time of the class-surface sequence on fresh PointCloud objects in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.build()
from pointCloudToolbox import PointCloud
from point_cloud_toolbox_amd import shapes, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts = shapes.torus_random(n, seed=1234)
for i in range(5):
    t0 = time.perf_counter(); pc = PointCloud(points=pts, normals=np.zeros((n, 0)))
    t1 = time.perf_counter(); pc.plant_kdtree(50, algorithm="grid")
    t2 = time.perf_counter(); K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    t3 = time.perf_counter(); pc.close()
    t4 = time.perf_counter()
    tm = pc.last_timings
    print(f"object {i}: ctor {1e3*(t1-t0):.2f} plant {1e3*(t2-t1):.2f} (upload {tm.get('upload_ms',0):.2f} grid {tm['grid_ms']:.2f} knn {tm['knn_ms']:.2f} passes {tm['grid_iters']}) curvature {1e3*(t3-t2):.2f} close {1e3*(t4-t3):.2f} ms", flush=True)
import cProfile, pstats
pc = PointCloud(points=pts, normals=np.zeros((n, 0)))
pr = cProfile.Profile(); pr.enable()
pc.plant_kdtree(50, algorithm="grid")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
