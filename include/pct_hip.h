/*
 * pct_hip.h -- C ABI of the MI355X (gfx950) per-point curvature path.
 *
 * The reference has no FFI layer: its boundary is the Python class PointCloud
 * (/root/reference/pointCloudToolbox.py:24).  These entry points are what the
 * methods on that class's hot path bind to through ctypes; each one cites the
 * reference method it replaces.  Plain C types only, caller-allocated output
 * buffers, int status return (PCT_OK == 0), blocking semantics, one HIP stream
 * per handle, no global mutable state (several handles / ranks may coexist).
 */
#ifndef PCT_HIP_H
#define PCT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pct_ctx pct_ctx;

enum pct_status {
    PCT_OK = 0,
    PCT_ERR_HIP = 1,          /* a HIP runtime call failed (see pct_last_error)        */
    PCT_ERR_NO_DEVICE = 2,    /* no usable gfx950 device                               */
    PCT_ERR_INVALID = 3,      /* bad argument / call order                             */
    PCT_ERR_NONFINITE = 4,    /* NaN/Inf in the cloud (pointCloudToolbox.py:273-274)   */
    PCT_ERR_K_TOO_LARGE = 5,  /* k + 1 > N (reference: IndexError at pct:640)          */
    PCT_ERR_OOM = 6,
    PCT_ERR_NO_NEIGHBORS = 7  /* fit requested before a neighbour table exists         */
};

enum pct_knn_algo {
    PCT_KNN_AUTO = 0,
    PCT_KNN_BRUTE = 1,        /* exhaustive wave-per-query sweep                        */
    PCT_KNN_GRID = 2,         /* uniform cell list, LDS-staged 27-cell stencil          */
    PCT_KNN_GRID_EXACT = 3,   /* same cell list, every query through the exact sweep    */
    PCT_KNN_GRID_LEVELS = 4,  /* chain of cell lists, each sized for the queries the previous
                                 one could not answer (clouds of very uneven density)    */
    PCT_KNN_TREE = 5          /* hierarchical cell list: the cloud in Morton order, every
                                 query swept at the octree level that suits ITS density
                                 (whole clouds below 2^26 points; anything else asked for
                                 it takes GRID_LEVELS).  PCT_KNN_AUTO takes it
                                 by itself where a census of the uniform list's work items
                                 predicts that it pays: surface-like clouds whose density
                                 spans a decade or more                                  */
};

/* Per-stage device times of the most recent call, hipEvent milliseconds. */
typedef struct pct_timings {
    float upload_ms;          /* H2D of coordinates (0 when device-resident)            */
    float grid_ms;            /* bounding box + cell sizing + counting sort             */
    float knn_ms;             /* neighbour sweep kernel(s) only (fast + exact redo)     */
    float knn_fast_ms;        /* of which the dominant kernel k_knn_fast alone          */
    float fit_ms;             /* fused plane-align + quadric fit + curvature kernel     */
    float export_ms;          /* sorted-space -> public index translation               */
    float total_ms;
    int32_t knn_launches;
    int32_t grid_iters;       /* cell-size refinement passes                            */
    int64_t cells;            /* grid cells                                             */
    int64_t occupied_cells;   /* work items of the sweep (cells split into query chunks) */
    int64_t ring_fallbacks;   /* queries that needed more than the 27-cell stencil      */
    int64_t lds_overflows;    /* cells whose stencil exceeded the LDS staging capacity  */
    int64_t flushes;          /* wave-wide sort/merge passes of the sweep               */
    int64_t candidate_steps;  /* 64-candidate distance steps of the sweep               */
    int64_t redone_queries;   /* queries re-done by the exact sweep (float-key collisions) */
    double cell_size;
    int64_t grid_points;      /* points held by the cell list: the cloud, or (sharded handles) the part of it
                                 near the owned range                                    */
    int32_t limit_retries;    /* 1 = the sweep was repeated with every point because a query reached past
                                 the part kept                                            */
    int32_t levels;           /* passes of the density-adaptive sweep (0 = not used)       */
    double occupancy;         /* mean number of points sharing a point's cell (the cell-size search steers on it) */
    int64_t fit_svd_rows;     /* rows of the last fit solved by the SVD kernel (lstsq's gelsd semantics: ill-conditioned
                                 or under-determined design matrices) instead of the normal equations */
    int32_t algo;             /* the sweep that produced the table in place (pct_knn_algo; what PCT_KNN_AUTO chose) */
    int32_t reserved;
} pct_timings;

/* ---- lifetime ---------------------------------------------------------- */
int pct_device_count(int* count);
int pct_create(int device, pct_ctx** out);
void pct_destroy(pct_ctx* ctx);
const char* pct_last_error(const pct_ctx* ctx);    /* never NULL */
const char* pct_version(void);

/* ---- cloud upload: PointCloud.__init__ / read_from_file (pct:26-66) ----- */
/* (n,3) row-major host coordinates.  The float32 form is the file path's
 * dtype (pct:52); the float64 form keeps native-dtype queries and centring
 * (pct:83, pct:641) while the search structure holds float32-rounded
 * coordinates (pct:74). */
int pct_set_points_f32(pct_ctx* ctx, const float* xyz, int64_t n);
int pct_set_points_f64(pct_ctx* ctx, const double* xyz, int64_t n);
/* Same, from a device pointer on this handle's device (multi-GPU: the buffer
 * an RCCL all-gather has just filled). */
int pct_set_points_device_f32(pct_ctx* ctx, const void* dev_xyz, int64_t n);
/* Same without the copy: the handle reads the caller's buffer in place.  The
 * buffer must stay allocated and unchanged until the next pct_set_points_* /
 * pct_use_points_* call on this handle or pct_destroy, and whatever filled it
 * must have completed (the handle's stream does not wait for other streams). */
int pct_use_points_device_f32(pct_ctx* ctx, const void* dev_xyz, int64_t n);
/* Queries owned by this handle: global index range [begin, end).  Default all. */
int pct_set_query_range(pct_ctx* ctx, int64_t begin, int64_t end);
/* Ownership by SLAB instead of by index range (multi-GPU, clouds whose row order is not a spatial order: the
 * neighbourhood of a random index range is the whole cloud, so every rank would bin every point).  The gathered cloud
 * is cut into `parts` slabs of equal population along its longest axis -- from a 4096-bin histogram that every rank
 * computes alike, with the same arithmetic on the same bytes -- and this handle answers the points of slab `part`:
 * its cell list holds that slab and a margin (checked per query like an index range's box, pct_timings.limit_retries).
 * pct_curvature only (float32 clouds of >= 4096 points, one cell list: PCT_KNN_AUTO / PCT_KNN_GRID; k > 127 goes
 * through the exact sweep); the rows come back as records, not by index:
 *   pct_slab_counts    points per slab of the last pct_curvature (what every rank will send)
 *   pct_slab_records   (public index as int32 bits, K, H) x rows of this slab -> a DEVICE buffer of 3 floats per row
 *   pct_scatter_records   gathered records (device) -> K, H of the public rows [begin, end) (device); fails unless
 *                         exactly end - begin records fell into the range (the slabs of the ranks partition the cloud)
 * parts = 1 is the same path with the whole cloud as its one slab (a world of one rank); pct_set_query_range, a new
 * cloud or parts <= 0 return the handle to index ranges.  pct_get_fit / pct_get_neighbors and
 * the other by-index getters refuse while slab ownership is on. */
#define PCT_SLAB_PARTS_MAX 64
int pct_set_query_slab(pct_ctx* ctx, int32_t part, int32_t parts);
int pct_slab_counts(pct_ctx* ctx, int64_t* counts, int32_t parts);
int pct_slab_records(pct_ctx* ctx, float* dev_records, int64_t capacity_rows, int64_t* rows);
int pct_scatter_records(pct_ctx* ctx, const float* dev_records, int64_t n_records, int64_t begin, int64_t end,
                        float* dev_K, float* dev_H);
/* Cell-occupancy target of the grid search as a multiple of (k+1); <= 0 keeps
 * the default. */
int pct_set_grid_param(pct_ctx* ctx, double occupancy_factor);
/* Sweep statistics in pct_timings (ring_fallbacks ... redone_queries); off by default
 * because the counters cost same-address atomics. */
int pct_set_stats(pct_ctx* ctx, int32_t enable);

/* ---- plant_kdtree(k) (pct:69-89) ---------------------------------------- */
/* k nearest neighbours of every owned point, self dropped, rows ascending.
 * eps > 0 adds the hybrid bound "distance < eps" (SURVEY A11); rows then carry
 * a valid count, missing slots index N / distance +inf. */
int pct_knn(pct_ctx* ctx, int32_t k, double eps, int32_t algo);
/* Lazy download of the neighbour table for rows [begin,end) of the cloud:
 * idx (rows,k) int32, dist (rows,k) float32, count (rows) int32 (any may be NULL). */
int pct_get_neighbors(pct_ctx* ctx, int64_t begin, int64_t end,
                      int32_t* idx, float* dist, int32_t* count);

/* Same for an explicit list of cloud rows (sampled checks on clouds whose full table is tens of GB). */
int pct_get_neighbor_rows(pct_ctx* ctx, const int64_t* rows, int64_t n_rows,
                          int32_t* idx, float* dist, int32_t* count);

/* ---- fit_explicit_quadratic_surfaces_to_neighborhoods (pct:635-647)
 *      + calculate_curvatures_of_explicit_quadratic_surfaces_for_all_points
 *        (pct:657-674) ---------------------------------------------------- */
/* From the device-resident neighbour table left by pct_knn. */
int pct_fit(pct_ctx* ctx);
/* From host-supplied neighbours: idx (rows,k) int32 for query points
 * query[rows] (NULL = 0..rows-1), optional per-row valid count.  The results replace those of an
 * earlier fit; a resident neighbour table stays as it is (pct_get_neighbors / pct_fit keep working). */
int pct_fit_indices(pct_ctx* ctx, const int32_t* idx, const int32_t* count,
                    const int64_t* query, int64_t rows, int32_t k);
/* Diagnostics variant of pct_fit_indices: the same neighbourhoods through the same kernel, but the
 * solution of the normal equations is returned unrounded (float64, where pct:359 casts to float32) and K, H are the
 * formulas of pct:403-419 evaluated in float64 from it.  The design matrix stays float32 (pct:350, 358): this shows
 * what the float32 rounding of the last two stages costs, not a different algorithm.  coefs (rows,6), K, H (rows)
 * float64 host arrays.  Leaves the resident neighbour table and float32 results untouched. */
int pct_fit_indices_f64(pct_ctx* ctx, const int32_t* idx, const int32_t* count,
                        const int64_t* query, int64_t rows, int32_t k,
                        double* coefs, double* K, double* H);
/* plant_kdtree + compute_pointwise_explicit_quadratic_curvature (pct:505-509)
 * without materialising the neighbour table on the host. */
int pct_curvature(pct_ctx* ctx, int32_t k, double eps, int32_t algo);

/* Results of the last fit for cloud rows [begin,end) (pct_fit / pct_curvature)
 * or for the rows of the last pct_fit_indices call.  Any pointer may be NULL.
 * coefs (rows,6) float32 [A,B,C,D,E,F]; K, H, H2 (rows) float32. */
int pct_get_fit(pct_ctx* ctx, int64_t begin, int64_t end,
                float* coefs, float* K, float* H, float* H2);

/* calculate_explicit_quadratic_curvatures (pct:398-431) on caller-supplied
 * coefficients: coefs (rows,6) float32 host -> K, H, H2 (rows) float32 host. */
int pct_curvatures_from_coefficients(pct_ctx* ctx, const float* coefs, int64_t rows,
                                     float* K, float* H, float* H2);

/* The two per-neighbourhood staticmethods of the class on their own, batched (batch neighbourhoods of m points):
 * get_best_fit_plane_and_rotate (pct:270-321): nbrs (batch, m, 3) float32 (is_f64 == 0) or float64 ->
 * rotated (batch, m, 3) float64: np.cov in float64 (ddof 1), normal = direction of least variance, flipped by
 * points[-1] - points[0] (subtracted in the input's dtype), Rodrigues rotation of the normal onto +z, R p per point.
 * PCT_ERR_NONFINITE for NaN/Inf in the input (pct:273) -- the caller checks the output (pct:318).  m >= 2. */
int pct_plane_rotate(pct_ctx* ctx, const void* nbrs, int32_t is_f64, int64_t batch, int32_t m, double* rotated);
/* fit_quadratic_surface (pct:331-360): pts (batch, m, 3) float32 (the cast of pct:350 is the caller's) ->
 * coefs (batch, 6) float32: numpy.linalg.lstsq(rcond=None) of the float32 design rows [a^2, b^2, ab, a, b, 1] --
 * float64 SVD-based solve, singular values below eps max(m, 6) sigma_1 cut off, minimum-norm solution. */
int pct_fit_quadric(pct_ctx* ctx, const float* pts, int64_t batch, int32_t m, float* coefs);

/* self.kdtree.query(x, k[, distance_upper_bound=eps]) (the reference's tree object, pointCloudToolbox.py:74; called
 * with arbitrary points at pct:625, 759, 844): the k nearest CLOUD points of each of m caller-supplied float64 query
 * points -- nothing is dropped (a query that coincides with a cloud point gets that point first, distance 0).
 * The cloud is the float32-rounded one the tree is built from (pct:74); squared distances are accumulated in float64
 * as ((dx^2 + dy^2) + dz^2), rows ascending, exact ties by index.  idx (m,k) int32, dist (m,k) float64; missing
 * entries (k > N, or beyond eps when eps > 0): index N and +inf, as SciPy pads.  1 <= k <= 128.
 * Exhaustive sweep, one wave per query: meant for the hundreds of sample points of the neighbour study, not for
 * the per-point loop (that is pct_knn). */
int pct_query_points(pct_ctx* ctx, const double* q_xyz, int64_t m, int32_t k, double eps,
                     int32_t* idx, double* dist);

/* explicit_quadratic_neighbor_study (pct:732-800), the numeric part: for every sample row s and every
 * neighbour count n in [n_lo, n_hi], the Gaussian curvature of the quadric fitted to the point itself plus
 * its n nearest neighbours (pct:759-761).  Needs a resident plain k-NN table with k >= n_hi.
 * K_out: (n_samples, n_hi - n_lo + 1) float32.  The bisection over n stays on the host. */
int pct_neighbor_study_curvatures(pct_ctx* ctx, const int64_t* sample_rows, int64_t n_samples,
                                   int32_t n_lo, int32_t n_hi, float* K_out);

/* load_mesh_compute_energies (utils.py:702-765), the consumer of K/H: bending energy sum(mean(H^2) * area),
 * stretching energy sum(mean(K) * area) (both nansum) and total area over a triangle mesh.  vertices (V,3)
 * float64, triangles (T,3) int32, curvature arrays (V) float32 (curvature_is_f64 == 0, what the path
 * produces) or float64; face means are taken in that dtype as NumPy does.  out3 = {bending, stretching, area}. */
int pct_mesh_energies(pct_ctx* ctx, const double* vertices, int64_t n_vertices, const int32_t* triangles,
                      int64_t n_triangles, const void* gaussian, const void* mean, int32_t curvature_is_f64,
                      double* out3);

/* ---- scan preparation (next row N4) ---------------------------------------- */
/* Voxel-grid down-sampling of convert_asc_to_ply.py:20-51: voxel = floor(coordinate / voxel_size) evaluated in the
 * coordinates' own dtype as NumPy does (the _f32 form divides by (float)voxel_size in float32), the
 * first point of every voxel is kept.  indices (caller-allocated, n entries) receives the kept input indices in
 * increasing order (= the reference's order of first occurrence), *count their number.  Works on any handle, with or
 * without a resident cloud; it borrows the cell list's scratch buffers, so a resident neighbour table is dropped
 * (plant it again before pct_fit / pct_get_neighbors); fit results already computed stay readable. */
int pct_voxel_downsample(pct_ctx* ctx, const double* xyz, int64_t n, double voxel_size, int64_t* indices, int64_t* count);
int pct_voxel_downsample_f32(pct_ctx* ctx, const float* xyz, int64_t n, double voxel_size, int64_t* indices, int64_t* count);
/* PCA surface variation as utils.py:778-829 DOCUMENTS it, for the loaded cloud: k_total neighbours including the point
 * itself, out[i] = lambda_min / (lambda_0 + lambda_1 + lambda_2 + 1e-10) of their 3 x 3 covariance, (owned rows)
 * float32.  (As WRITTEN, utils.py:822 builds the k x k Gram matrix, whose smallest eigenvalue is zero: the function
 * returns LAPACK round-off around 0.  The Python wrapper's default follows the code as written -- zeros -- and offers
 * this estimator as an option.) */
int pct_surface_variation(pct_ctx* ctx, int32_t k_total, float* out);

/* ---- ingest / egress around the path (host code, no device needed) ------- */
/* The text scans read_from_file parses with np.loadtxt (pct:51): rows x cols of whitespace-separated numbers,
 * '#' comments and blank lines skipped.  Values are correctly rounded float64 (what Python's float() gives). */
int pct_text_shape(const char* path, int64_t* rows, int32_t* cols);
int pct_text_load(const char* path, int64_t rows, int32_t cols, double* out);
/* The constructor's three matrix norms (pct:45-47) from one multi-threaded pass over the (N, 3) array: out10 =
 * {sum|x|, sum|y|, sum|z| (float64), largest row sum (|x| + |y|) + |z| in the array's dtype, Gram matrix xx xy xz yy yz
 * zz (float64)}; the spectral norm is the square root of the Gram matrix' largest eigenvalue.  NaN input: NaN out. */
int pct_matrix_norms_f32(const float* xyz, int64_t n, double* out10);
int pct_matrix_norms_f64(const double* xyz, int64_t n, double* out10);
/* repr(float(x)) as Python prints it -- what an f-string gives for a np.float32 widened to double -- into out32
 * (>= 32 bytes, not NUL-terminated); returns the length. */
int pct_format_float(double x, char* out32);
/* The ASCII PLY of utils.py:538-551: header + one line 'x y z K H' per vertex, numbers as the reference's f-string
 * prints them. */
int pct_write_ply_ascii(const char* path, const float* xyz, const float* gaussian, const float* mean, int64_t n);

/* ---- multi-GPU exchange --------------------------------------------------------------------------
 * The reference is one process; here the cloud shards by point-index range over up to 8 GPUs, one process per GPU,
 * and the only exchange is an all-gather of the float32 coordinate shards (12 B / point): RCCL over xGMI, called from
 * this library (librccl.so is opened at run time; no PyTorch in the process).  Rank 0 creates the unique id and the
 * host code hands its 128 bytes to every rank (point-cloud-toolbox_amd/dist.py: a TCP socket on MASTER_ADDR). */
int pct_comm_unique_id(void* id128);
int pct_comm_init(pct_ctx* ctx, int32_t rank, int32_t world, const void* id128);
int pct_comm_destroy(pct_ctx* ctx);
/* Rank r contributes counts[r] floats at dev_send; dev_recv receives all shards back to back in rank order.
 * Asynchronous: runs on the handle's exchange stream, ordered after the work the compute stream holds at the call.
 * One exchange in flight per handle: a second call before pct_comm_wait / pct_comm_synchronize is PCT_ERR_INVALID.
 * Equal shards: ncclAllGather into dev_recv; unequal ones (zero-sized too): shards padded to the largest, one
 * ncclAllGather, one compaction pass.  PCT_COMM_FORCE=allgather|padded|bcast (environment) picks the form. */
int pct_comm_allgather_f32(pct_ctx* ctx, const void* dev_send, void* dev_recv, const int64_t* counts);
/* Collectives issued so far by this handle: out4 = {ncclAllGather in place, padded ncclAllGather + compaction, groups
 * of per-rank ncclBroadcast, ncclAllReduce}. */
int pct_comm_counters(pct_ctx* ctx, int64_t* out4);
/* The compute stream waits for the last exchange on the device (the host does not block) ... */
int pct_comm_wait(pct_ctx* ctx);
/* ... or the host does. */
int pct_comm_synchronize(pct_ctx* ctx);
/* values[0..n) reduced over the ranks in place (op 0 sum, 2 max, 3 min; n <= 8); blocking, also drains this handle's
 * compute stream first: with n == 0 it is the barrier of the bench's timed region. */
int pct_comm_allreduce_f64(pct_ctx* ctx, double* values, int32_t n, int32_t op);

/* ---- measurement -------------------------------------------------------- */
int pct_get_timings(const pct_ctx* ctx, pct_timings* out);
/* Streams of clouds.  With pct_set_async(ctx, 1) a pct_curvature call on a whole-cloud handle (no query range or slab,
 * statistics off) returns as soon as its kernels are enqueued: the host uploads and enqueues cloud i + 1 while the device
 * still fits cloud i.  Every other entry point (getters, pct_synchronize, pct_get_timings, a new cloud ...) first waits
 * for the pending call, so results are never observed early; an error of the pending call's kernels is reported by the
 * call that waits.  pct_get_timings_done does NOT wait: it returns the timings of the most recent call whose kernels are
 * known to have finished (the call before the pending one) -- what a loop polls after every step.  Off by default. */
int pct_set_async(pct_ctx* ctx, int32_t enable);
int pct_get_timings_done(const pct_ctx* ctx, pct_timings* out);
/* sizeof(pct_timings) as the library was built: a binding checks its own struct against it at load time. */
int pct_timings_size(void);
/* Device pointer helpers for zero-copy interop (multi-GPU all-gather target). */
int pct_device_alloc(pct_ctx* ctx, int64_t bytes, void** dev_ptr);
int pct_device_free(pct_ctx* ctx, void* dev_ptr);
int pct_device_upload(pct_ctx* ctx, void* dev_dst, const void* host_src, int64_t bytes);
int pct_device_download(pct_ctx* ctx, void* host_dst, const void* dev_src, int64_t bytes);
int pct_synchronize(pct_ctx* ctx);
/* Hardware self-test of the cross-lane primitives the sweep relies on
 * (DPP / ds_swizzle lane exchanges); *failures == 0 on a healthy gfx950. */
int pct_selftest(pct_ctx* ctx, int32_t* failures);

#ifdef __cplusplus
}
#endif
#endif /* PCT_HIP_H */
