"""Curvatures of a STREAM of clouds at the kernel rate (a scanner's frames, a folder of scans): two device handles take
the clouds in turn with asynchronous calls (``pct_set_async``), so that while the device works on one cloud the host moves
the previous cloud's K / H out and the next cloud's coordinates in.  One thread, no extra copies; results come back in the
order of the input, bit-identical to ``PointCloud`` / a single handle (1 M-point clouds, k = 50: 0.79 ms per cloud with the
PCIe transfers against 1.25 ms one cloud at a time -- bench.py, ``end_to_end_stream``).

    from point_cloud_toolbox_amd.stream import curvature_stream
    for K, H in curvature_stream(frames, k=50):
        ...

The reference has no counterpart (its loop is one ``PointCloud`` per file, pct:26 / utils:481); this is the streaming form
of rows A3-A9 of SURVEY 8(a) for callers that have more than one cloud.
"""
import numpy as np

from . import _capi


def curvature_stream(clouds, k, eps=0.0, device=0, algorithm=None):
    """Yields ``(K, H)`` (float32 arrays of length N_i) for every ``(N_i, 3)`` array of ``clouds``, in order.

    ``clouds`` may be any iterable (a generator is consumed one cloud ahead); float64 clouds stay float64, as with the
    class (pct:74, 83).  ``algorithm``: a ``_capi.KNN_*`` constant, default the library's choice per cloud."""
    algo = _capi.KNN_AUTO if algorithm is None else int(algorithm)
    handles = [_capi.acquire_handle(device), _capi.acquire_handle(device)]
    for h in handles:
        h.set_async(True)
    pending = [None, None]                                   # rows of the cloud each handle is working on
    try:
        turn = 0
        for pts in clouds:
            h = handles[turn]
            if pending[turn] is not None:                    # this handle's previous cloud: wait, fetch, hand out
                _, K, H, _ = h.get_fit(0, pending[turn], coefs=False, H2=False)
                yield K, H
            pts = np.asarray(pts)
            if pts.ndim != 2 or pts.shape[1] != 3:
                raise ValueError("every cloud must have shape (N, 3)")
            h.set_points(pts if pts.dtype == np.float64 else np.ascontiguousarray(pts, dtype=np.float32))
            h.curvature(k, eps, algo)                        # returns once enqueued
            pending[turn] = len(pts)
            turn ^= 1
        for _ in range(2):                                   # drain, in input order
            if pending[turn] is not None:
                _, K, H, _ = handles[turn].get_fit(0, pending[turn], coefs=False, H2=False)
                pending[turn] = None
                yield K, H
            turn ^= 1
    finally:
        for h in handles:
            try:
                h.set_async(False)
            finally:
                _capi.release_handle(h)
