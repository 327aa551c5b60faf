"""MI355X-backed ``PointCloud``: the reference's class surface for the curvature path.

Mirrors /root/reference/pointCloudToolbox.py:24-1009 for the methods on the hot
path: same constructor signature, same attribute names,
same curvature-array outputs, same exception types and messages.  The work is
done by hand-written HIP kernels behind the C ABI of ``include/pct_hip.h``;
there is no CPU fallback -- without the shared library or a gfx950 device the
compute methods raise.

Deliberate, documented differences (none changes a value):
* ``neighbor_indices`` / ``dists`` stay on the device and are downloaded on first
  access (400 MB at 1 M points, k=50);
* ``quadratic_coefficients`` is an (N, 6) float32 array and ``K_quadratic`` /
  ``H_quadratic`` / ``K_H_sq_quadratic`` are (N,) float32 arrays instead of
  Python lists of the same float32 values (pct:637, pct:659-661); every use in
  the reference's callers (len, indexing, np.save, np.isnan, iteration) works
  on both.
"""
from __future__ import annotations

import atexit
import threading
import zlib

import numpy as np

from . import _capi

__all__ = ["PointCloud"]

_ALGORITHMS = {"auto": _capi.KNN_AUTO, "brute": _capi.KNN_BRUTE, "grid": _capi.KNN_GRID,
               "grid_exact": _capi.KNN_GRID_EXACT, "grid_levels": _capi.KNN_GRID_LEVELS, "tree": _capi.KNN_TREE}


def _matrix_norms(points):
    """``np.linalg.norm(points, ord)`` for ord = 1, 2, inf (pct:45-47; nothing in the reference reads them) from one
    transposed copy of the (N, 3) matrix instead of three strided passes and an SVD: column sums accumulated in
    float64, the spectral norm as the square root of the largest eigenvalue of the 3x3 Gram matrix, the row sums in
    numpy's own order.  Same values up to rounding (numpy's float32 column sums are themselves only good to ~1e-4),
    ~6x faster per million points; non-finite input raises the LinAlgError the SVD would raise."""
    p = np.asarray(points)
    if p.ndim != 2 or p.shape[0] == 0 or p.shape[1] != 3 or p.dtype.kind != "f":
        return np.linalg.norm(points, 1), np.linalg.norm(points, 2), np.linalg.norm(points, np.inf)
    if p.dtype in (np.float32, np.float64) and p.flags.c_contiguous:
        try:
            cols, row_max, gram = _capi.matrix_norm_sums(p)       # one native multi-threaded pass (12 -> ~1 ms per million points)
        except _capi.HipExtensionError:
            cols = None
        if cols is not None:
            if not np.isfinite(gram).all() or not np.isfinite(row_max):
                raise np.linalg.LinAlgError("SVD did not converge")
            l2 = np.sqrt(max(np.linalg.eigvalsh(gram)[-1], 0.0))
            return p.dtype.type(cols.max()), p.dtype.type(l2), p.dtype.type(row_max)
    pt = np.ascontiguousarray(p.T)
    a = np.abs(pt)
    l1 = a.sum(1, dtype=np.float64).max()
    linf = ((a[0] + a[1]) + a[2]).max()
    p64 = pt.astype(np.float64, copy=False)
    gram = p64 @ p64.T
    if not np.isfinite(gram).all():
        raise np.linalg.LinAlgError("SVD did not converge")
    l2 = np.sqrt(max(np.linalg.eigvalsh(gram)[-1], 0.0))
    return p.dtype.type(l1), p.dtype.type(l2), linf


class _DeviceTree:
    """``self.kdtree`` of the reference (SciPy's k-d tree of the float32 cloud, pct:74) as far as its callers use it
    (pct:625, 759, 844): ``query(x, k)`` for arbitrary points, answered by the exhaustive device sweep
    (``pct_query_points``).  Same return convention as SciPy: distances float64 ascending and indices, shape
    ``x.shape[:-1] + (k,)`` (the ``k`` axis squeezed for ``k == 1``), missing entries ``inf`` / ``n``."""

    def __init__(self, cloud):
        self._cloud = cloud
        self.n = cloud.num_points
        self.m = 3

    @property
    def data(self):
        return np.asarray(self._cloud.points, dtype=np.float32).astype(np.float64)

    def query(self, x, k=1, eps=0, p=2, distance_upper_bound=np.inf, workers=1):
        if eps != 0 or p != 2:
            raise NotImplementedError("only exact Euclidean queries (eps=0, p=2) run on the device")
        x = np.asarray(x, dtype=np.float64)
        if x.shape[-1:] != (3,):
            raise ValueError(f"x must consist of vectors of length 3 but has shape {x.shape}")
        if not isinstance(k, (int, np.integer)) or k < 1:
            raise ValueError("k must be an integer >= 1")
        flat = x.reshape(-1, 3)
        bound = float(distance_upper_bound)
        idx, dist = self._cloud._ctx().query_points(flat, int(k), 0.0 if not np.isfinite(bound) else bound)
        shape = x.shape[:-1] + ((int(k),) if k > 1 else ())
        return dist.reshape(shape), idx.astype(np.intp).reshape(shape)


class PointCloud:

    # pct:26 -- identical signature and defaults
    def __init__(self, file_path=None, points=None, normals=None, downsample=False, voxel_size=0,
                 k_neighbors=20, output_path='./output/', max_points_per_voxel=1, device=0):
        self.downsample = downsample
        self.k_neighbors = k_neighbors
        self.voxel_size = voxel_size
        self.max_points_per_voxel = max_points_per_voxel
        self.output_path = output_path
        self.random_indexes = []
        self._device = device
        self._handle = None
        self._cloud_on_device = False
        self._table_on_device = False   # the neighbour table of the last planting is resident (and not yet replaced)
        self._uploaded_fp = None        # fingerprint of the points array the device cloud was uploaded from
        self._n_dev = 0                 # rows of the device cloud
        self._nbr_cache = None
        self._user_neighbors = None
        self._fit_on_device = False
        self._coefs_cache = None
        self._user_coefs = None
        self._device_curv = None
        self._plant_token = 0
        self._fit_token = -1
        self.eps = None
        self.collect_stats = False      # sweep statistics in last_timings (costs atomics)

        if file_path:
            self.file_path = file_path
            self.read_from_file()
        elif points is not None and normals is not None:
            self.points = points
            self.normals = normals
        else:
            raise ValueError("Either file_path or points and normals must be provided")      # pct:41

        # pct:43-47
        self.num_points = len(self.points)
        self.num_features = len(self.points[0])
        self.l1_norm, self.l2_norm, self.infinity_norm = _matrix_norms(self.points)

    # pct:50-66
    def read_from_file(self):
        points = _capi.load_text(self.file_path)           # np.loadtxt semantics, native multi-threaded parser
        self.points = points[:, 0:3].astype(np.float32)
        self.normals = points[:, 3:6].astype(np.float32)
        self.points[:, 0] -= np.max(self.points[:, 0])
        self.points[:, 1] -= np.max(self.points[:, 1])
        if self.downsample:
            # pct:59-60 calls a method the reference has commented out (pct:159-193)
            raise AttributeError("'PointCloud' object has no attribute 'downsample_point_cloud_by_grid'")
        self.x_domain = [np.min(self.points[:, 0]), np.max(self.points[:, 0])]
        self.y_domain = [np.min(self.points[:, 1]), np.max(self.points[:, 1])]
        self.z_domain = [np.min(self.points[:, 2]), np.max(self.points[:, 2])]

    # ---------------------------------------------------------------- device
    def _fingerprint(self):
        """Cheap identity of ``self.points`` as it is NOW: the object, its buffer and layout, and a checksum of about a
        thousand rows spread over the array.  Assigning ``pc.points`` or rewriting the array in place changes it (an
        in-place edit of a few rows between the sample can escape; plant_kdtree does not rely on it)."""
        p = self.points
        a = np.asarray(p)
        step = max(1, len(a) // 1024)
        sample = np.ascontiguousarray(a[::step]) if a.ndim == 2 else a
        return (id(p), a.shape, a.dtype.str, a.__array_interface__["data"][0], a.strides, zlib.crc32(sample.tobytes()))

    def _upload(self):
        """The cloud as ``self.points`` holds it now -> device (the reference builds its tree from the array of the
        moment, pct:74, and gathers from the array of the moment, pct:640)."""
        if self._handle is None:
            self._handle = _capi.acquire_handle(self._device)      # raises without library / GPU
            self._handle.set_stats(self.collect_stats)
        pts = np.asarray(self.points)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise ValueError("points must have shape (N, 3)")
        if pts.dtype != np.float64:
            pts = pts.astype(np.float32, copy=False)
        self._handle.set_points(pts)
        self._table_on_device = False                      # a new cloud drops the device table
        self._cloud_on_device = True
        self._uploaded_fp = self._fingerprint()
        self._n_dev = len(pts)
        return self._handle

    def _ctx(self):
        if self._handle is None or not self._cloud_on_device:
            self._upload()
        return self._handle

    def close(self):
        """Release the device handle (an extension: the reference holds no device state).  Fitted coefficients that
        still live on the device are brought to the host first (24 B/point), so ``quadratic_coefficients`` and a later
        ``calculate_curvatures_...`` keep working; the neighbour table is dropped unless it has been read already."""
        if self._handle is not None:
            if self._fit_on_device and self._user_coefs is None:
                self._user_coefs = self.quadratic_coefficients
            _capi.release_handle(self._handle)              # (an idle context keeps its buffers for the next cloud)
            self._handle = None
            self._table_on_device = False
            self._cloud_on_device = False
            self._fit_on_device = False
            self._device_curv = None

    # ------------------------------------------------------------------ A3
    def plant_kdtree(self, k_neighbors, eps=None, algorithm="auto"):
        """k-NN table for every point (pct:69-89), computed on the GPU.

        ``eps`` (extension, README.md:8 / SURVEY A11): hybrid query, at most k
        neighbours with distance < eps; ``neighbor_counts`` then holds the
        number of valid entries per row.
        """
        self.k_neighbors = k_neighbors                      # pct:71
        self.eps = eps
        algo = _ALGORITHMS[algorithm]
        if self._fit_on_device and self._user_coefs is None and self._handle is not None:
            # the reference keeps the fitted coefficients across a re-planting (utils.py:495-501):
            # bring them to the host before the device results are invalidated
            self._user_coefs = self.quadratic_coefficients
        # pct:74 builds a NEW tree from self.points as they are at every planting: so does this (12 B/point over PCIe;
        # a cloud that was assigned or edited since the last call must not be answered from the old upload)
        h = self._upload()
        h.knn(k_neighbors, eps or 0.0, algo)
        self._table_on_device = True
        self.kdtree = _DeviceTree(self)                     # pct:74-75: the tree object later methods query
        self._nbr_cache = None
        self._user_neighbors = None
        self._fit_on_device = False
        self._plant_token += 1
        self.last_timings = h.timings()

    def _download_neighbors(self):
        if self._nbr_cache is None:
            if self._handle is None or not self._table_on_device:
                raise AttributeError("'PointCloud' object has no attribute 'neighbor_indices'")
            idx, dist, cnt = self._handle.get_neighbors(0, self._n_dev, True, True, True)
            self._nbr_cache = (idx, dist, cnt)
        return self._nbr_cache

    @property
    def neighbor_indices(self):
        """(N, k) int32, rows ascending by distance, self excluded (pct:79, 85)."""
        if self._user_neighbors is not None:
            return self._user_neighbors
        return self._download_neighbors()[0]

    @neighbor_indices.setter
    def neighbor_indices(self, value):
        self._user_neighbors = np.asarray(value)      # used by the next fit; what has been fitted already stays (pct:637)

    @property
    def dists(self):
        """(N, k) float32 (pct:78, 84)."""
        return self._download_neighbors()[1]

    @property
    def neighbor_counts(self):
        return self._download_neighbors()[2]

    # ------------------------------------------------------------------ A4
    def fit_explicit_quadratic_surfaces_to_neighborhoods(self):
        """Plane-align + quadric fit of every neighbourhood (pct:635-647)."""
        h = self._ctx()
        if self._fingerprint() != self._uploaded_fp:
            # self.points were assigned or rewritten after the upload: pct:640 gathers the CURRENT coordinates with the
            # table of the last planting -- the table comes to the host (indices, distances, counts: the reference keeps
            # them as attributes) before the new upload drops it, and goes back as rows for the new cloud
            if self._table_on_device:
                self._download_neighbors()
                if self._user_neighbors is None:
                    self._user_neighbors = self._nbr_cache[0]
            h = self._upload()
        if self._user_neighbors is not None:
            h.fit_indices(self._user_neighbors)
        else:
            h.fit()                                          # AttributeError if no table (pct:640)
        # results stay on the device: coefficients (24 B/point) are downloaded on first access, K/H/H^2 by
        # calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points()
        self._coefs_cache = None
        self._user_coefs = None
        self._device_curv = None
        self._fit_on_device = True
        self._fit_token = self._plant_token
        self.last_timings = h.timings()

    @property
    def quadratic_coefficients(self):
        """(N, 6) float32 [A, B, C, D, E, F] per point (pct:637-647), fetched from the device on first access."""
        if self._user_coefs is not None:
            return self._user_coefs
        if not self._fit_on_device:
            raise AttributeError("'PointCloud' object has no attribute 'quadratic_coefficients'")
        if self._coefs_cache is None:
            self._coefs_cache = self._handle.get_fit(0, self._n_dev, K=False, H=False, H2=False)[0]
        return self._coefs_cache

    @quadratic_coefficients.setter
    def quadratic_coefficients(self, value):
        self._user_coefs = value

    # ------------------------------------------------------------------ A8
    def calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points(self):
        """K, H, H^2 from the fitted coefficients (pct:657-674)."""
        if self._user_coefs is None and self._fit_on_device and self._handle is not None:
            if self._device_curv is None:                    # produced by the fused kernel, still on the device
                self._device_curv = self._handle.get_fit(0, self._n_dev, coefs=False)[1:]
            K, H, H2 = self._device_curv
        else:                                                # coefficients the caller supplied, or a re-planted table
            K, H, H2 = self._ctx().curvatures_from_coefficients(np.asarray(self.quadratic_coefficients))
        self.K_quadratic = K
        self.H_quadratic = H
        self.K_H_sq_quadratic = H2
        return self.K_quadratic, self.H_quadratic

    # ------------------------------------------------------------------ A9
    def compute_pointwise_explicit_quadratic_curvature(self):
        """pct:505-509."""
        self.fit_explicit_quadratic_surfaces_to_neighborhoods()
        K, H = self.calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points()
        return np.array(K), np.array(H)

    def export_ply_with_curvatures(self, filename='output_with_curvatures.ply'):
        """The ASCII PLY validate_shape writes after the curvature step (utils.py:538-551), same bytes."""
        _capi.write_ply_ascii(filename, np.asarray(self.points), self.K_quadratic, self.H_quadratic)

    # ----------------------------------------------------------------- A10
    def explicit_quadratic_neighbor_study(self, tol=1e-7, sample_size=500, lower_bound=3, upper_bound=99):
        """Neighbour count at which the Gaussian curvature stops changing (pct:732-800).

        Same draw (``np.random.randint`` on the global generator, pct:753), same per-point bisection
        (pct:772-789) and same return value ``int(mean) + 1`` (pct:800) as the reference.  The curvature
        K(n) of "the point itself plus its n nearest neighbours" (pct:759-761) is evaluated on the GPU for
        every n the bisection can ask for, from a k = upper_bound+1 neighbour table.
        """
        num_total = len(self.points)
        sample_size = min(sample_size, num_total)
        random_indexes = np.random.randint(0, num_total, sample_size)                  # pct:753
        if sample_size == 0:
            return 0                                                                   # pct:797-798
        need = upper_bound + 1
        h = self._ctx()
        own = self._table_on_device and getattr(h, "k", 0) >= need and not self.eps and self._user_neighbors is None
        if not own:      # keep the planted table untouched: a second context does the k=need sweep
            h = _capi.Handle(self._device)
            pts = np.asarray(self.points)
            h.set_points(pts if pts.dtype == np.float64 else pts.astype(np.float32, copy=False))
            h.knn(need)
        try:
            Kn = h.neighbor_study_curvatures(random_indexes, lower_bound, need)        # columns n = lower..upper+1
        finally:
            if not own:
                h.close()
        converged = []
        for row in Kn:
            lower, upper, best = lower_bound, upper_bound, None
            while lower <= upper:                                                      # pct:778-786
                mid = (lower + upper) // 2
                if abs(row[mid + 1 - lower_bound] - row[mid - lower_bound]) < tol:
                    best, upper = mid, mid - 1
                else:
                    lower = mid + 1
            converged.append(upper if best is None else best)                          # pct:787-788
        return int(np.mean(converged)) + 1                                             # pct:800

    # fused entry: k-NN -> fit -> curvature with nothing but K/H leaving the GPU
    def compute_curvature_fused(self, k_neighbors, eps=None, algorithm="auto"):
        self.k_neighbors = k_neighbors
        self.eps = eps
        algo = _ALGORITHMS[algorithm]
        h = self._upload()                                 # the cloud of the moment, as plant_kdtree (pct:74)
        h.curvature(k_neighbors, eps or 0.0, algo)
        self._table_on_device = True
        self._nbr_cache = None
        self._user_neighbors = None
        _, K, H, H2 = h.get_fit(0, self._n_dev, coefs=False)
        self.K_quadratic, self.H_quadratic, self.K_H_sq_quadratic = K, H, H2
        self._plant_token += 1
        self._fit_token = self._plant_token
        self._fit_on_device, self._user_coefs, self._coefs_cache, self._device_curv = True, None, None, (K, H, H2)
        self.last_timings = h.timings()
        return K, H

    # --------------------------------------------------------- staticmethods
    # The reference calls these once per point in a Python loop (pct:644-647, 668): one device context serves all
    # calls of the process (created on first use, released at exit), under a lock -- a handle has ONE stream.
    @staticmethod
    def get_best_fit_plane_and_rotate(points):
        """pct:270-321 for one neighbourhood: (k, 3) -> (k, 3) float64 with the best-fit normal on +z."""
        pts = np.asarray(points)
        if not np.all(np.isfinite(pts)):
            raise ValueError("Non-finite values in input points")                      # pct:273-274
        if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < 2:
            return _reference_small_case(pts)
        with _static_lock:
            rotated = _static_handle().plane_rotate(pts[None])[0]
        if not np.all(np.isfinite(rotated)):
            raise ValueError("Non-finite values after rotation")                       # pct:318-319
        return rotated

    @staticmethod
    def fit_quadratic_surface(points):
        """pct:331-360 for one neighbourhood: (N, 3) [a, b, z] -> float32 (6,) [A, B, C, D, E, F]."""
        points = np.array(points, dtype=np.float32)                                    # pct:350
        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("Input points must have shape (N, 3)")                    # pct:351-352
        if not np.all(np.isfinite(points)):
            raise ValueError("Input contains non-finite values.")                      # pct:356-357
        if points.shape[0] == 0:
            raise np.linalg.LinAlgError("0-dimensional array given")                   # what lstsq says to an empty system
        with _static_lock:
            return _static_handle().fit_quadric(points[None])[0]

    @staticmethod
    def calculate_explicit_quadratic_curvatures(coefficients):
        """pct:398-431 for one coefficient vector (device evaluation)."""
        c = np.asarray(coefficients, dtype=np.float32).reshape(1, 6)
        with _static_lock:
            K, H, H2 = _static_handle().curvatures_from_coefficients(c)
        disc = max(H[0] ** 2 - K[0], 0)                      # pct:425
        root = np.sqrt(disc)
        return K[0], H[0], H[0] + root, H[0] - root, H2[0]


_static_lock = threading.Lock()
_static = {"handle": None}


def _static_handle():
    if _static["handle"] is None:
        _static["handle"] = _capi.Handle(0)                 # raises without library / GPU: no CPU fallback
        atexit.register(_close_static_handle)
    return _static["handle"]


def _close_static_handle():
    h, _static["handle"] = _static["handle"], None
    if h is not None:
        h.close()


def _reference_small_case(pts):
    """What pct:277-280 does with a block the device path does not take (measured with NumPy 2.2): one point or none --
    np.cov has nothing to average, the covariance is NaN and np.linalg.svd raises LinAlgError("SVD did not converge");
    a 1-D array -- np.cov returns a scalar and svd refuses it; (m, c != 3) -- svd works but the rotation of pct:315
    cannot be formed (ValueError from the matrix product)."""
    if pts.ndim == 1:
        raise np.linalg.LinAlgError("0-dimensional array given. Array must be at least two-dimensional")
    if pts.ndim == 2 and pts.shape[1] == 3:
        raise np.linalg.LinAlgError("SVD did not converge")
    raise ValueError("Input points must have shape (N, 3)")
