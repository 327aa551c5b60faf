"""G4-full: the WHOLE sample_scans/bunny.txt (35 947 rows) through the unmodified reference's FILE constructor at k = 30
(BASELINE.md's first plumbing line; G4 holds rows 0-3999 only): tests/golden/g4_bunny_full_file_k30_sample.npz.

TEST INFRASTRUCTURE, build container only:  MPLBACKEND=Agg python oracle/make_goldens_bunny_full.py   (~15 s)

The whole class runs on the whole file (max-shift of pct:56-57 included); 3 000 sampled rows of every output are kept
(indices, float32 distances, coefficients, K, H, H^2) together with the shifted float32 cloud the class holds
(`points`, 431 KB) -- the scan's rows as data: tests/golden/bunny_xyz_f32.npy is the unshifted file.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import load_reference, run_full, OUT, REF      # noqa: E402


def main():
    ref = load_reference()
    g = run_full(ref, 30, file_path=os.path.join(REF, "sample_scans", "bunny.txt"))
    n = len(g["points"])
    rows = np.sort(np.random.default_rng(404).choice(n, 3000, replace=False))
    out = dict(points=g["points"].astype(np.float32), rows=rows.astype(np.int64), k=g["k"], n=np.int64(n),
               idx=np.asarray(g["idx"])[rows], dists=np.asarray(g["dists"])[rows], coefs=g["coefs"][rows],
               K=np.asarray(g["K"])[rows], H=np.asarray(g["H"])[rows], H2=g["H2"][rows])
    path = os.path.join(OUT, "g4_bunny_full_file_k30_sample.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", n, "rows, sample", len(rows))


if __name__ == "__main__":
    main()
