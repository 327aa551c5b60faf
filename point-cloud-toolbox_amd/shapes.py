"""Seeded synthetic clouds for parity tests and the bench.

The formulas restate the reference's shape generators
(/root/reference/utils.py:858-866 Fibonacci sphere, :888-896 torus with R=1,
r=1/3, :911-914 egg carton 0.1 sin(pi x) cos(pi y)); sampling is this build's
own: float64 maths, ``np.random.default_rng(seed)`` (PCG64), one cast to
float32 at the end.  Closed-form curvatures are returned next to the points so
tests can compare against them (plot_shape_validation_results.py:28-45).
"""
from __future__ import annotations

import numpy as np

TORUS_R = 1.0
TORUS_r = 1.0 / 3.0


def fibonacci_sphere(n, radius=1.0, dtype=np.float32):
    """Fibonacci lattice (utils.py:859-865): K = 1/r^2, H = 1/r."""
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    p = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1) * radius
    return p.astype(dtype)


def torus_angles(n, seed=1234, lo=0, hi=None):
    """(theta, phi) ~ U[0, 2pi)^2 for rows [lo, hi) of the n-point cloud.

    The stream is drawn in fixed blocks so that any index range can be
    regenerated without materialising the whole cloud (multi-GPU shards).
    """
    hi = n if hi is None else hi
    blk = 1 << 20
    out = np.empty((hi - lo, 2), dtype=np.float64)
    b0, b1 = lo // blk, (max(hi, lo + 1) - 1) // blk
    for b in range(b0, b1 + 1):
        rng = np.random.default_rng([seed, b])
        m = min(blk, n - b * blk)
        ang = rng.uniform(0.0, 2.0 * np.pi, size=(m, 2))
        s, e = max(lo, b * blk), min(hi, b * blk + m)
        if e > s:
            out[s - lo:e - lo] = ang[s - b * blk:e - b * blk]
    return out


def torus_from_angles(ang, R=TORUS_R, r=TORUS_r, dtype=np.float32):
    th, ph = ang[:, 0], ang[:, 1]
    w = R + r * np.cos(ph)
    p = np.stack([w * np.cos(th), w * np.sin(th), r * np.sin(ph)], 1)
    return p.astype(dtype)


def torus_random(n, seed=1234, R=TORUS_R, r=TORUS_r, dtype=np.float32, lo=0, hi=None, with_truth=False):
    """Random-parameter torus (primary, tie-free bench input)."""
    ang = torus_angles(n, seed, lo, hi)
    p = torus_from_angles(ang, R, r, dtype)
    if not with_truth:
        return p
    cph = np.cos(ang[:, 1])
    K = cph / (r * (R + r * cph))
    H = (R + 2.0 * r * cph) / (2.0 * r * (R + r * cph))
    return p, K, H


def torus_scan_order(n, parts, part, seed=1234, R=TORUS_R, r=TORUS_r, dtype=np.float32):
    """Rows of part ``part`` of an n-point random torus whose index order follows the major angle.

    The reference's generator walks the surface theta-major (utils.py:888-891) and scanners emit points along
    their sweep, so contiguous index ranges are spatially compact.  This is the random-parameter equivalent:
    part p (rows [n*p//parts, n*(p+1)//parts)) holds theta ~ U[2 pi p/parts, 2 pi (p+1)/parts), phi ~ U[0, 2 pi),
    in random order inside the part.  ``parts == 1`` is not ``torus_random`` (different stream).
    """
    lo, hi = (n * part) // parts, (n * (part + 1)) // parts
    rng = np.random.default_rng([seed, parts, part])
    u = rng.uniform(0.0, 1.0, size=(hi - lo, 2))
    ang = np.stack([2.0 * np.pi * (part + u[:, 0]) / parts, 2.0 * np.pi * u[:, 1]], 1)
    return torus_from_angles(ang, R, r, dtype)


def torus_grid(n_side, R=TORUS_R, r=TORUS_r, dtype=np.float32):
    """The reference's own torus (utils.py:883-896 generate_torus_points with num_points = n_side^2, so that no
    rows are re-drawn): theta x phi lattice, ``linspace(0, 2 pi, n_side, endpoint=False)`` on both axes, phi-major
    row order (``meshgrid`` then ``ravel``).  Every point has symmetric partners at (nearly) equal distances."""
    t = np.linspace(0.0, 2.0 * np.pi, n_side, endpoint=False)
    th, ph = np.meshgrid(t, t)
    return torus_from_angles(np.stack([th.ravel(), ph.ravel()], 1), R, r, dtype)


def egg_carton_grid(n_side, amp=0.1, dtype=np.float32):
    """The reference's own egg carton (utils.py:906-914 generate_egg_carton_points, num_points = n_side^2):
    [-1, 1]^2 lattice with both end points, z = amp sin(pi x) cos(pi y)."""
    t = np.linspace(-1.0, 1.0, n_side)
    X, Y = np.meshgrid(t, t)
    Z = amp * np.sin(X * np.pi) * np.cos(Y * np.pi)
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(dtype)


def egg_carton_random(n, seed=1234, amp=0.1, dtype=np.float32, lo=0, hi=None, with_truth=False):
    """(x, y) ~ U[-1, 1]^2, z = amp sin(pi x) cos(pi y) (utils.py:911-914)."""
    hi = n if hi is None else hi
    blk = 1 << 20
    xy = np.empty((hi - lo, 2), dtype=np.float64)
    b0, b1 = lo // blk, (max(hi, lo + 1) - 1) // blk
    for b in range(b0, b1 + 1):
        rng = np.random.default_rng([seed, 7, b])
        m = min(blk, n - b * blk)
        u = rng.uniform(-1.0, 1.0, size=(m, 2))
        s, e = max(lo, b * blk), min(hi, b * blk + m)
        if e > s:
            xy[s - lo:e - lo] = u[s - b * blk:e - b * blk]
    x, y = xy[:, 0], xy[:, 1]
    z = amp * np.sin(np.pi * x) * np.cos(np.pi * y)
    p = np.stack([x, y, z], 1).astype(dtype)
    if not with_truth:
        return p
    pi = np.pi
    fx = amp * pi * np.cos(pi * x) * np.cos(pi * y)
    fy = -amp * pi * np.sin(pi * x) * np.sin(pi * y)
    fxx = -amp * pi * pi * np.sin(pi * x) * np.cos(pi * y)
    fyy = fxx
    fxy = -amp * pi * pi * np.cos(pi * x) * np.sin(pi * y)
    w = 1.0 + fx * fx + fy * fy
    K = (fxx * fyy - fxy * fxy) / w ** 2
    H = ((1 + fx * fx) * fyy - 2 * fx * fy * fxy + (1 + fy * fy) * fxx) / (2 * w ** 1.5)
    return p, K, H


def tile_cloud(base, n_tiles, pitch=0.25, lattice=(9, 8, 8), dtype=np.float32, only=None):
    """Translated copies of ``base`` on a lattice (bunny x 557).  ``only``: the tiles to produce (a rank
    of a sharded run builds just the ones its index range touches), concatenated in the given order."""
    base = np.asarray(base, dtype=np.float64)
    base = base - base.min(0)
    tiles = list(range(n_tiles)) if only is None else [int(t) for t in only]
    out = np.empty((len(tiles) * len(base), 3), dtype=dtype)
    nx, ny, _ = lattice
    for o, t in enumerate(tiles):
        if not 0 <= t < n_tiles:
            raise ValueError(f"tile {t} outside [0, {n_tiles})")
        off = np.array([t % nx, (t // nx) % ny, t // (nx * ny)], dtype=np.float64) * pitch
        out[o * len(base):(o + 1) * len(base)] = (base + off).astype(dtype)
    return out
