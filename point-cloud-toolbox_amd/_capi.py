"""ctypes binding of libpct_hip.so (C ABI: include/pct_hip.h).

There is no CPU fallback: if the shared library is missing or no gfx950
device is usable every entry point raises.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` (drives hipcc).
"""
from __future__ import annotations

import ctypes as C
import atexit
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCT_LIB") or os.path.join(_HERE, "libpct_hip.so")     # PCT_LIB: a developer's A/B build (tools/build_variant.py)

PCT_OK = 0
PCT_ERR_HIP = 1
PCT_ERR_NO_DEVICE = 2
PCT_ERR_INVALID = 3
PCT_ERR_NONFINITE = 4
PCT_ERR_K_TOO_LARGE = 5
PCT_ERR_OOM = 6
PCT_ERR_NO_NEIGHBORS = 7

KNN_AUTO, KNN_BRUTE, KNN_GRID, KNN_GRID_EXACT, KNN_GRID_LEVELS, KNN_TREE = 0, 1, 2, 3, 4, 5


class Timings(C.Structure):
    _fields_ = [
        ("upload_ms", C.c_float), ("grid_ms", C.c_float), ("knn_ms", C.c_float), ("knn_fast_ms", C.c_float),
        ("fit_ms", C.c_float), ("export_ms", C.c_float), ("total_ms", C.c_float),
        ("knn_launches", C.c_int32), ("grid_iters", C.c_int32),
        ("cells", C.c_int64), ("occupied_cells", C.c_int64),
        ("ring_fallbacks", C.c_int64), ("lds_overflows", C.c_int64),
        ("flushes", C.c_int64), ("candidate_steps", C.c_int64), ("redone_queries", C.c_int64),
        ("cell_size", C.c_double),
        ("grid_points", C.c_int64), ("limit_retries", C.c_int32), ("levels", C.c_int32),
        ("occupancy", C.c_double),
        ("fit_svd_rows", C.c_int64),
        ("algo", C.c_int32), ("reserved", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_p = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol declared in include/pct_hip.h
SIGNATURES = {
    "pct_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pct_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "pct_destroy": (None, [_p]),
    "pct_last_error": (C.c_char_p, [_p]),
    "pct_version": (C.c_char_p, []),
    "pct_set_points_f32": (C.c_int, [_p, _f32p, C.c_int64]),
    "pct_set_points_f64": (C.c_int, [_p, _f64p, C.c_int64]),
    "pct_set_points_device_f32": (C.c_int, [_p, _p, C.c_int64]),
    "pct_use_points_device_f32": (C.c_int, [_p, _p, C.c_int64]),
    "pct_set_query_range": (C.c_int, [_p, C.c_int64, C.c_int64]),
    "pct_set_query_slab": (C.c_int, [_p, C.c_int32, C.c_int32]),
    "pct_slab_counts": (C.c_int, [_p, C.POINTER(C.c_int64), C.c_int32]),
    "pct_slab_records": (C.c_int, [_p, _p, C.c_int64, C.POINTER(C.c_int64)]),
    "pct_scatter_records": (C.c_int, [_p, _p, C.c_int64, C.c_int64, C.c_int64, _p, _p]),
    "pct_set_grid_param": (C.c_int, [_p, C.c_double]),
    "pct_set_async": (C.c_int, [_p, C.c_int32]),
    "pct_get_timings_done": (C.c_int, [_p, C.POINTER(Timings)]),
    "pct_set_stats": (C.c_int, [_p, C.c_int32]),
    "pct_knn": (C.c_int, [_p, C.c_int32, C.c_double, C.c_int32]),
    "pct_get_neighbors": (C.c_int, [_p, C.c_int64, C.c_int64, _i32p, _f32p, _i32p]),
    "pct_get_neighbor_rows": (C.c_int, [_p, _i64p, C.c_int64, _i32p, _f32p, _i32p]),
    "pct_fit": (C.c_int, [_p]),
    "pct_fit_indices": (C.c_int, [_p, _i32p, _i32p, _i64p, C.c_int64, C.c_int32]),
    "pct_curvature": (C.c_int, [_p, C.c_int32, C.c_double, C.c_int32]),
    "pct_get_fit": (C.c_int, [_p, C.c_int64, C.c_int64, _f32p, _f32p, _f32p, _f32p]),
    "pct_curvatures_from_coefficients": (C.c_int, [_p, _f32p, C.c_int64, _f32p, _f32p, _f32p]),
    "pct_neighbor_study_curvatures": (C.c_int, [_p, _i64p, C.c_int64, C.c_int32, C.c_int32, _f32p]),
    "pct_fit_indices_f64": (C.c_int, [_p, _i32p, _i32p, _i64p, C.c_int64, C.c_int32, _f64p, _f64p, _f64p]),
    "pct_plane_rotate": (C.c_int, [_p, _p, C.c_int32, C.c_int64, C.c_int32, _f64p]),
    "pct_fit_quadric": (C.c_int, [_p, _f32p, C.c_int64, C.c_int32, _f32p]),
    "pct_query_points": (C.c_int, [_p, _f64p, C.c_int64, C.c_int32, C.c_double, _i32p, _f64p]),
    "pct_mesh_energies": (C.c_int, [_p, _f64p, C.c_int64, _i32p, C.c_int64, _p, _p, C.c_int32, _f64p]),
    "pct_voxel_downsample": (C.c_int, [_p, _f64p, C.c_int64, C.c_double, _i64p, _i64p]),
    "pct_voxel_downsample_f32": (C.c_int, [_p, _f32p, C.c_int64, C.c_double, _i64p, _i64p]),
    "pct_surface_variation": (C.c_int, [_p, C.c_int32, _f32p]),
    "pct_text_shape": (C.c_int, [C.c_char_p, _i64p, _i32p]),
    "pct_text_load": (C.c_int, [C.c_char_p, C.c_int64, C.c_int32, _f64p]),
    "pct_format_float": (C.c_int, [C.c_double, C.c_char_p]),
    "pct_matrix_norms_f32": (C.c_int, [_f32p, C.c_int64, _f64p]),
    "pct_matrix_norms_f64": (C.c_int, [_f64p, C.c_int64, _f64p]),
    "pct_write_ply_ascii": (C.c_int, [C.c_char_p, _f32p, _f32p, _f32p, C.c_int64]),
    "pct_comm_unique_id": (C.c_int, [_p]),
    "pct_comm_init": (C.c_int, [_p, C.c_int32, C.c_int32, _p]),
    "pct_comm_destroy": (C.c_int, [_p]),
    "pct_comm_allgather_f32": (C.c_int, [_p, _p, _p, _i64p]),
    "pct_comm_counters": (C.c_int, [_p, _i64p]),
    "pct_comm_wait": (C.c_int, [_p]),
    "pct_comm_synchronize": (C.c_int, [_p]),
    "pct_comm_allreduce_f64": (C.c_int, [_p, _f64p, C.c_int32, C.c_int32]),
    "pct_get_timings": (C.c_int, [_p, C.POINTER(Timings)]),
    "pct_timings_size": (C.c_int, []),
    "pct_device_alloc": (C.c_int, [_p, C.c_int64, C.POINTER(_p)]),
    "pct_device_free": (C.c_int, [_p, _p]),
    "pct_device_upload": (C.c_int, [_p, _p, _p, C.c_int64]),
    "pct_device_download": (C.c_int, [_p, _p, _p, C.c_int64]),
    "pct_synchronize": (C.c_int, [_p]),
    "pct_selftest": (C.c_int, [_p, _i32p]),
}

_lib = None


class HipExtensionError(RuntimeError):
    """The HIP extension is missing or unusable -- there is no CPU fallback."""


def load():
    """Load libpct_hip.so and attach prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run __graft_entry__.build()); this package has no CPU fallback")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise HipExtensionError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the ABI and this table diverge
        fn.restype = res
        fn.argtypes = args
    if lib.pct_timings_size() != C.sizeof(Timings):
        raise HipExtensionError(f"{LIB_PATH}: pct_timings is {lib.pct_timings_size()} bytes, this binding expects "
                                f"{C.sizeof(Timings)} (library and package out of step: rebuild)")
    _lib = lib
    return lib


def load_text(path):
    """np.loadtxt(path) for whitespace-separated numeric scans, multi-threaded native parser (float64 out)."""
    lib = load()
    rows, cols = C.c_int64(0), C.c_int32(0)
    bpath = os.fsencode(path)
    if lib.pct_text_shape(bpath, C.byref(rows), C.byref(cols)) != PCT_OK:
        raise ValueError(f"cannot parse {path!r} as a rectangular table of numbers")
    out = np.empty((rows.value, cols.value), np.float64)
    if rows.value and lib.pct_text_load(bpath, rows.value, cols.value, _ptr(out, _f64p)) != PCT_OK:
        raise ValueError(f"cannot parse {path!r} as a rectangular table of numbers")
    return out


def matrix_norm_sums(points):
    """One native pass over a C-contiguous (N, 3) float32 / float64 array: (column sums of |.|, largest row sum of |.|,
    3 x 3 Gram matrix) -- what np.linalg.norm(points, 1 | inf | 2) are made of (host code, no device)."""
    p = np.asarray(points)
    out = np.empty(10, np.float64)
    if p.dtype == np.float32:
        st = load().pct_matrix_norms_f32(_ptr(p, _f32p), len(p), _ptr(out, _f64p))
    else:
        st = load().pct_matrix_norms_f64(_ptr(p, _f64p), len(p), _ptr(out, _f64p))
    if st != PCT_OK:
        raise ValueError("matrix norms: bad array")
    g = out[4:]
    gram = np.array([[g[0], g[1], g[2]], [g[1], g[3], g[4]], [g[2], g[4], g[5]]])
    return out[:3], out[3], gram


def format_float(x):
    """repr(float(x)) computed natively (the number format of the PLY writer)."""
    buf = C.create_string_buffer(40)
    n = load().pct_format_float(float(x), buf)
    return buf.raw[:n].decode()


def write_ply_ascii(path, points, gaussian, mean):
    p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    g = np.ascontiguousarray(gaussian, dtype=np.float32)
    m = np.ascontiguousarray(mean, dtype=np.float32)
    if len(g) != len(p) or len(m) != len(p):
        raise ValueError("one curvature value per point is required")
    if load().pct_write_ply_ascii(os.fsencode(path), _ptr(p, _f32p), _ptr(g, _f32p), _ptr(m, _f32p), len(p)) != PCT_OK:
        raise OSError(f"cannot write {path!r}")


def device_count():
    n = C.c_int(0)
    load().pct_device_count(C.byref(n))
    return n.value


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (rank 0 calls this and hands the bytes to every rank)."""
    buf = C.create_string_buffer(128)
    if load().pct_comm_unique_id(C.cast(buf, _p)) != PCT_OK:
        raise HipExtensionError("pct_comm_unique_id failed: librccl.so cannot be opened or refused")
    return buf.raw


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


_handle_pool = {}          # device -> one idle Handle whose device buffers stay allocated (PointCloud.close puts it there)


def acquire_handle(device=0):
    """A device context for a PointCloud: the idle one of the pool if there is one (its buffers -- two dozen
    hipMallocs, ~4 ms for a million points -- are reused by the next cloud), else a new one."""
    h = _handle_pool.pop(int(device), None)
    if h is not None and getattr(h, "_h", None) and h._h.value:
        return h
    return Handle(device)


def release_handle(h):
    """Back to the pool (one idle handle per device is kept; a second one is closed).  PCT_NO_HANDLE_POOL=1 closes."""
    if h is None or not getattr(h, "_h", None) or not h._h.value:
        return
    if os.environ.get("PCT_NO_HANDLE_POOL") or h.device in _handle_pool:
        h.close()
        return
    try:
        h.set_stats(False)
        h.set_async(False)
        for attr in ("k", "eps"):
            if hasattr(h, attr):
                delattr(h, attr)
    except Exception:
        h.close()
        return
    _handle_pool[h.device] = h


atexit.register(lambda: drain_handle_pool())


def drain_handle_pool():
    """Close the idle handles (their device memory goes back to the runtime)."""
    while _handle_pool:
        _handle_pool.popitem()[1].close()


class Handle:
    """One device context (one HIP stream) -- thin, typed wrapper over pct_ctx."""

    def __init__(self, device=0):
        self._lib = load()
        self._h = _p()
        st = self._lib.pct_create(int(device), C.byref(self._h))
        if st != PCT_OK:
            self._h = _p()
            raise HipExtensionError(
                f"pct_create(device={device}) failed with status {st}: no usable MI355X (gfx950) device; "
                "this package has no CPU fallback")
        self.device = int(device)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.pct_destroy(self._h)
            self._h = _p()

    __del__ = close

    def _check(self, st):
        if st == PCT_OK:
            return
        msg = self._lib.pct_last_error(self._h).decode("utf-8", "replace")
        if st == PCT_ERR_NONFINITE:
            raise ValueError(msg or "Non-finite values in input points")     # pct:274
        if st == PCT_ERR_K_TOO_LARGE:
            raise IndexError(msg)                                            # reference: IndexError at pct:640
        if st == PCT_ERR_INVALID:
            raise ValueError(msg)
        if st == PCT_ERR_OOM:
            raise MemoryError(msg)
        if st == PCT_ERR_NO_NEIGHBORS:
            raise AttributeError(msg)                                        # reference: no self.neighbor_indices yet
        raise HipExtensionError(f"status {st}: {msg}")

    # -- cloud ------------------------------------------------------------
    def set_points(self, pts):
        pts = np.asarray(pts)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must have shape (N, 3)")
        if pts.dtype == np.float64:
            p = np.ascontiguousarray(pts)
            self._check(self._lib.pct_set_points_f64(self._h, _ptr(p, _f64p), len(p)))
        else:
            p = np.ascontiguousarray(pts, dtype=np.float32)
            self._check(self._lib.pct_set_points_f32(self._h, _ptr(p, _f32p), len(p)))
        self.n = len(p)

    def set_points_device(self, dev_ptr, n):
        self._check(self._lib.pct_set_points_device_f32(self._h, _p(int(dev_ptr)), int(n)))
        self.n = int(n)

    def use_points_device(self, dev_ptr, n):
        """Zero-copy variant: the caller keeps the buffer alive and unchanged while the handle uses it."""
        self._check(self._lib.pct_use_points_device_f32(self._h, _p(int(dev_ptr)), int(n)))
        self.n = int(n)

    def set_query_range(self, begin, end):
        self._check(self._lib.pct_set_query_range(self._h, int(begin), int(end)))

    # -- ownership by slab (multi-GPU, clouds in no spatial order; include/pct_hip.h) --------------------------------
    def set_query_slab(self, part, parts):
        self._check(self._lib.pct_set_query_slab(self._h, int(part), int(parts)))

    def slab_counts(self, parts):
        out = (C.c_int64 * int(parts))()
        self._check(self._lib.pct_slab_counts(self._h, out, int(parts)))
        return [int(v) for v in out]

    def slab_records(self, dev_ptr, capacity_rows):
        """(public index bits, K, H) of this slab's rows into the device buffer; returns the row count."""
        rows = C.c_int64(0)
        self._check(self._lib.pct_slab_records(self._h, _p(int(dev_ptr)), int(capacity_rows), C.byref(rows)))
        return rows.value

    def scatter_records(self, dev_records, n_records, begin, end, dev_K, dev_H):
        self._check(self._lib.pct_scatter_records(self._h, _p(int(dev_records)), int(n_records), int(begin), int(end),
                                                  _p(int(dev_K)), _p(int(dev_H))))

    def set_stats(self, enable=True):
        self._check(self._lib.pct_set_stats(self._h, int(bool(enable))))

    def set_grid_param(self, occupancy_factor):
        self._check(self._lib.pct_set_grid_param(self._h, float(occupancy_factor)))

    # -- path -------------------------------------------------------------
    def knn(self, k, eps=0.0, algo=KNN_AUTO):
        self._check(self._lib.pct_knn(self._h, int(k), float(eps or 0.0), int(algo)))
        self.k = int(k)

    def fit(self):
        self._check(self._lib.pct_fit(self._h))

    def curvature(self, k, eps=0.0, algo=KNN_AUTO):
        self._check(self._lib.pct_curvature(self._h, int(k), float(eps or 0.0), int(algo)))
        self.k = int(k)

    def get_neighbors(self, begin, end, want_idx=True, want_dist=True, want_count=False):
        rows = int(end) - int(begin)
        idx = np.empty((rows, self.k), np.int32) if want_idx else None
        dist = np.empty((rows, self.k), np.float32) if want_dist else None
        cnt = np.empty(rows, np.int32) if want_count else None
        self._check(self._lib.pct_get_neighbors(self._h, int(begin), int(end), _ptr(idx, _i32p),
                                                _ptr(dist, _f32p), _ptr(cnt, _i32p)))
        return idx, dist, cnt

    def query_points(self, q, k, eps=0.0):
        """The reference tree's ``query`` for arbitrary points: (m,3) float64 -> idx (m,k) int32 (missing: N), dist (m,k) float64 (missing: inf)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        if q.ndim != 2 or q.shape[1] != 3:
            raise ValueError("query points must have shape (m, 3)")
        idx = np.empty((len(q), int(k)), np.int32)
        dist = np.empty((len(q), int(k)), np.float64)
        self._check(self._lib.pct_query_points(self._h, _ptr(q, _f64p), len(q), int(k), float(eps or 0.0), _ptr(idx, _i32p), _ptr(dist, _f64p)))
        return idx, dist

    def get_neighbor_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        idx = np.empty((len(rows), self.k), np.int32)
        dist = np.empty((len(rows), self.k), np.float32)
        cnt = np.empty(len(rows), np.int32)
        self._check(self._lib.pct_get_neighbor_rows(self._h, _ptr(rows, _i64p), len(rows), _ptr(idx, _i32p),
                                                   _ptr(dist, _f32p), _ptr(cnt, _i32p)))
        return idx, dist, cnt

    def fit_indices(self, idx, count=None, query=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if idx.ndim != 2:
            raise ValueError("neighbour indices must have shape (rows, k)")
        cnt = None if count is None else np.ascontiguousarray(count, dtype=np.int32)
        qry = None if query is None else np.ascontiguousarray(query, dtype=np.int64)
        self._check(self._lib.pct_fit_indices(self._h, _ptr(idx, _i32p), _ptr(cnt, _i32p), _ptr(qry, _i64p),
                                              idx.shape[0], idx.shape[1]))
        return idx.shape[0]

    def fit_indices_f64(self, idx, count=None, query=None):
        """Diagnostics: unrounded float64 coefficients (rows,6) and float64 K, H of the given neighbourhoods."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if idx.ndim != 2:
            raise ValueError("neighbour indices must have shape (rows, k)")
        cnt = None if count is None else np.ascontiguousarray(count, dtype=np.int32)
        qry = None if query is None else np.ascontiguousarray(query, dtype=np.int64)
        rows = idx.shape[0]
        c, K, H = np.empty((rows, 6), np.float64), np.empty(rows, np.float64), np.empty(rows, np.float64)
        self._check(self._lib.pct_fit_indices_f64(self._h, _ptr(idx, _i32p), _ptr(cnt, _i32p), _ptr(qry, _i64p), rows, idx.shape[1],
                                                  _ptr(c, _f64p), _ptr(K, _f64p), _ptr(H, _f64p)))
        return c, K, H

    def get_fit(self, begin, end, coefs=True, K=True, H=True, H2=True):
        rows = int(end) - int(begin)
        c = np.empty((rows, 6), np.float32) if coefs else None
        # K, H, H^2 share one host block, back to back: the library moves what is adjacent on both sides in ONE copy
        # (two 4 MB copies cost two fixed set-ups: 0.65 -> 0.4 ms for a million points)
        want = [K, H, H2]
        block = np.empty((sum(want), rows), np.float32)
        it = iter(block)
        k, h, h2 = (next(it) if w else None for w in want)
        self._check(self._lib.pct_get_fit(self._h, int(begin), int(end), _ptr(c, _f32p), _ptr(k, _f32p),
                                          _ptr(h, _f32p), _ptr(h2, _f32p)))
        return c, k, h, h2

    def curvatures_from_coefficients(self, coefs):
        c = np.ascontiguousarray(coefs, dtype=np.float32).reshape(-1, 6)
        k = np.empty(len(c), np.float32)
        h = np.empty(len(c), np.float32)
        h2 = np.empty(len(c), np.float32)
        self._check(self._lib.pct_curvatures_from_coefficients(self._h, _ptr(c, _f32p), len(c), _ptr(k, _f32p),
                                                               _ptr(h, _f32p), _ptr(h2, _f32p)))
        return k, h, h2

    def plane_rotate(self, nbrs):
        """get_best_fit_plane_and_rotate for a block of neighbourhoods: (batch, m, 3) float32 / float64 -> float64."""
        a = np.asarray(nbrs)
        a = np.ascontiguousarray(a, dtype=np.float64 if a.dtype == np.float64 else np.float32)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("neighbourhoods must have shape (batch, m, 3)")
        out = np.empty(a.shape, np.float64)
        self._check(self._lib.pct_plane_rotate(self._h, a.ctypes.data_as(_p), int(a.dtype == np.float64), a.shape[0], a.shape[1],
                                               _ptr(out, _f64p)))
        return out

    def fit_quadric(self, pts):
        """fit_quadratic_surface for a block of rotated neighbourhoods: (batch, m, 3) float32 -> (batch, 6) float32."""
        a = np.ascontiguousarray(pts, dtype=np.float32)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("Input points must have shape (batch, N, 3)")
        out = np.empty((a.shape[0], 6), np.float32)
        self._check(self._lib.pct_fit_quadric(self._h, _ptr(a, _f32p), a.shape[0], a.shape[1], _ptr(out, _f32p)))
        return out

    def neighbor_study_curvatures(self, sample_rows, n_lo, n_hi):
        rows = np.ascontiguousarray(sample_rows, dtype=np.int64)
        out = np.empty((len(rows), int(n_hi) - int(n_lo) + 1), np.float32)
        self._check(self._lib.pct_neighbor_study_curvatures(self._h, _ptr(rows, _i64p), len(rows), int(n_lo), int(n_hi),
                                                           _ptr(out, _f32p)))
        return out

    def mesh_energies(self, vertices, triangles, gaussian, mean):
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        f64 = np.asarray(gaussian).dtype == np.float64 and np.asarray(mean).dtype == np.float64
        dt = np.float64 if f64 else np.float32
        g = np.ascontiguousarray(gaussian, dtype=dt)
        m = np.ascontiguousarray(mean, dtype=dt)
        if len(g) != len(v) or len(m) != len(v):
            raise ValueError("one curvature value per vertex is required")
        out = np.zeros(3, np.float64)
        self._check(self._lib.pct_mesh_energies(self._h, _ptr(v, _f64p), len(v), _ptr(t, _i32p), len(t), g.ctypes.data_as(_p),
                                               m.ctypes.data_as(_p), int(f64), _ptr(out, _f64p)))
        return float(out[0]), float(out[1]), float(out[2])

    def voxel_downsample(self, xyz, voxel_size):
        """Indices of the first point of every voxel.  float32 coordinates are binned in float32, everything else in
        float64 -- as ``np.floor(coordinates / voxel_size)`` evaluates (convert_asc_to_ply.py:34)."""
        xyz = np.asarray(xyz)
        f32 = xyz.dtype == np.float32
        p = np.ascontiguousarray(xyz, dtype=np.float32 if f32 else np.float64).reshape(-1, 3)
        idx = np.empty(len(p), np.int64)
        cnt = C.c_int64(0)
        if f32:
            self._check(self._lib.pct_voxel_downsample_f32(self._h, _ptr(p, _f32p), len(p), float(voxel_size), _ptr(idx, _i64p), C.byref(cnt)))
        else:
            self._check(self._lib.pct_voxel_downsample(self._h, _ptr(p, _f64p), len(p), float(voxel_size), _ptr(idx, _i64p), C.byref(cnt)))
        return idx[:cnt.value].copy()

    def surface_variation(self, k_total):
        out = np.empty(self.n, np.float32)
        self._check(self._lib.pct_surface_variation(self._h, int(k_total), _ptr(out, _f32p)))
        self.k = int(k_total) - 1
        return out

    def timings(self):
        t = Timings()
        self._lib.pct_get_timings(self._h, C.byref(t))
        return t.as_dict()

    def set_async(self, enable=True):
        """Streams of clouds: ``curvature`` on a whole-cloud handle returns once its kernels are enqueued; every other
        call first waits for it (include/pct_hip.h, pct_set_async)."""
        self._check(self._lib.pct_set_async(self._h, int(bool(enable))))

    def stage_times_done(self):
        """Timings of the most recent call whose kernels are known to have finished -- with ``set_async`` the call before
        the pending one; does not wait.  One struct kept on the handle, refilled and returned."""
        t = self.__dict__.get("_tm")
        if t is None:
            t = self._tm = Timings()
            self._tm_ref = C.byref(t)
        self._lib.pct_get_timings_done(self._h, self._tm_ref)
        return t

    def stage_times(self):
        """The same record as ``timings`` without building a dict: one struct kept on the handle, refilled and returned
        (read ``.grid_ms``, ``.knn_ms``, ...).  For loops that poll after every step while the GPU waits for the host."""
        t = self.__dict__.get("_tm")
        if t is None:
            t = self._tm = Timings()
            self._tm_ref = C.byref(t)
        self._lib.pct_get_timings(self._h, self._tm_ref)
        return t

    # -- multi-GPU exchange (RCCL behind the C ABI) ---------------------------
    def comm_init(self, rank, world, unique_id):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.pct_comm_init(self._h, int(rank), int(world), C.cast(buf, _p)))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        self._check(self._lib.pct_comm_destroy(self._h))

    def comm_allgather(self, dev_send, dev_recv, counts):
        """Start the all-gather of float32 shards (counts[r] floats from rank r) on the exchange stream."""
        c = np.ascontiguousarray(counts, dtype=np.int64)
        self._check(self._lib.pct_comm_allgather_f32(self._h, _p(int(dev_send)), _p(int(dev_recv)), _ptr(c, _i64p)))

    def comm_counters(self):
        """Collectives issued by this handle: {allgather, padded_allgather, broadcast_groups, allreduce}."""
        c = np.zeros(4, np.int64)
        self._check(self._lib.pct_comm_counters(self._h, _ptr(c, _i64p)))
        return dict(zip(("allgather", "padded_allgather", "broadcast_groups", "allreduce"), (int(v) for v in c)))

    def comm_wait(self):
        self._check(self._lib.pct_comm_wait(self._h))

    def comm_synchronize(self):
        self._check(self._lib.pct_comm_synchronize(self._h))

    def comm_allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._check(self._lib.pct_comm_allreduce_f64(self._h, _ptr(v, _f64p), len(v), {"sum": 0, "max": 2, "min": 3}[op]))
        return v

    def comm_barrier(self):
        self._check(self._lib.pct_comm_allreduce_f64(self._h, None, 0, 0))

    # -- raw device memory (multi-GPU all-gather target) ---------------------
    def device_alloc(self, nbytes):
        out = _p()
        self._check(self._lib.pct_device_alloc(self._h, int(nbytes), C.byref(out)))
        return out.value

    def device_free(self, ptr):
        self._check(self._lib.pct_device_free(self._h, _p(int(ptr))))

    def device_upload(self, ptr, host):
        host = np.ascontiguousarray(host)
        self._check(self._lib.pct_device_upload(self._h, _p(int(ptr)), host.ctypes.data_as(_p), host.nbytes))

    def device_download(self, ptr, host):
        assert host.flags["C_CONTIGUOUS"]
        self._check(self._lib.pct_device_download(self._h, host.ctypes.data_as(_p), _p(int(ptr)), host.nbytes))

    def selftest(self):
        n = C.c_int32(-1)
        self._check(self._lib.pct_selftest(self._h, C.byref(n)))
        return n.value

    def synchronize(self):
        self._check(self._lib.pct_synchronize(self._h))
