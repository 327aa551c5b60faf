"""Sampled reference goldens for the reference's OWN lattice inputs (build container only; TEST INFRASTRUCTURE).

G9a  the 1000 x 1000 (theta, phi) torus of utils.py:883-896 (generate_torus_points(1_000_000): grid_size = 1000,
     nothing re-drawn), float32, k = 50: the reference's staticmethods on 2000 fixed sample rows.
G9b  sample_scans/egg_carton.txt (316 x 316 lattice, 99 856 points) through the reference's FILE constructor
     (float32 cast + max shift, pct:50-57), k = 30: the whole run, 4000 sampled rows kept; the shifted float32 cloud
     is stored with it (a data file of the reference: the GPU box has no /root/reference).
G9c  the 1000 x 1000 egg carton of utils.py:906-914, float32, k = 50, 2000 sample rows.
Lattices are full of (near-)equal distances: tests compare distances bit for bit and indices outside exact ties.
Run from the repo root:  MPLBACKEND=Agg python oracle/make_goldens_lattice.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, REF, load_reference, load_shapes, run_sampled  # noqa: E402


def main():
    ref = load_reference()
    sh = load_shapes()
    rng = np.random.default_rng(777)

    P = sh.torus_grid(1000)
    rows = np.sort(rng.choice(len(P), 2000, replace=False))
    np.savez_compressed(os.path.join(OUT, "g9_torusgrid1m_k50_sample.npz"), **run_sampled(ref, P, 50, rows))
    print("torus grid done", flush=True)

    P = sh.egg_carton_grid(1000)
    rows = np.sort(rng.choice(len(P), 2000, replace=False))
    np.savez_compressed(os.path.join(OUT, "g9_egggrid1m_k50_sample.npz"), **run_sampled(ref, P, 50, rows))
    print("egg grid done", flush=True)

    pc = ref.PointCloud(os.path.join(REF, "sample_scans", "egg_carton.txt"))
    pc.plant_kdtree(30)
    K, H = pc.compute_pointwise_explicit_quadratic_curvature()
    rows = np.sort(rng.choice(pc.num_points, 4000, replace=False))
    np.savez_compressed(os.path.join(OUT, "g9_eggcarton_file_k30_sample.npz"),
                        points=np.asarray(pc.points, np.float32), rows=rows, k=np.int32(30),
                        idx=pc.neighbor_indices[rows], dists=pc.dists[rows],
                        coefs=np.stack(pc.quadratic_coefficients).astype(np.float32)[rows], K=K[rows], H=H[rows])
    print("egg_carton.txt done", flush=True)


if __name__ == "__main__":
    main()
