"""Random cross-check of the voxel down-sampling against the NumPy restatement (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as ge
ge.build()
import pct_oracle as oracle
import pointCloudToolbox  # noqa: F401
from point_cloud_toolbox_amd.prep import downsample
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
it = 0
while time.time() < t_end:
    rng = np.random.default_rng([3, it])
    n = int(10 ** rng.uniform(1, 5.5))
    scale = 10.0 ** rng.uniform(-3, 3)
    pts = rng.normal(size=(n, 3)) * scale + rng.uniform(-1, 1, 3) * scale * 10 ** rng.uniform(-1, 3)
    if rng.random() < 0.3: pts = np.round(pts / scale * 8) * scale / 8          # duplicates, points on voxel faces
    voxel = scale * 10.0 ** rng.uniform(-3, 1)
    got, idx = downsample(pts, voxel, return_indices=True)
    ref = oracle.voxel_downsample(pts, voxel)
    if not (np.array_equal(got, ref) and (np.diff(idx) > 0).all()):
        print("MISMATCH", it, n, scale, voxel, len(got), len(ref)); sys.exit(1)
    it += 1
print("prep fuzz ok:", it)
