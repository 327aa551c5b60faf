// Internal declarations shared by the HIP translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/pct_hip.h"

#define PCT_WAVE 64
#define PCT_K_MAX 511           // longest neighbour row: k + 1 <= 512 = 64 lanes x 8 list registers (wave-per-query sweeps, pct_knn_wide.hip)

// Every kernel launch of the library leaves its source position and kernel name here; the abort hook
// (pct_api.hip: install_abort_trace) prints it, so that the log of a GPU memory fault -- ROCr aborts the process from
// its own thread -- names the launch that preceded it (exactly the faulting one under HIP_LAUNCH_BLOCKING=1).
extern const char* pct_last_launch;
#define PCT_STR2(x) #x
#define PCT_STR(x) PCT_STR2(x)
// (a macro of the library's own at its launch sites: HIP's hipLaunchKernelGGL is left alone, also for the rocPRIM
// headers included after this one; the last-launch word is one relaxed atomic store, any handle's thread may write it)
#define PCT_LAUNCH(kernelName, ...)                                                          \
    do {                                                                                     \
        __atomic_store_n(&pct_last_launch, __FILE__ ":" PCT_STR(__LINE__) "  " #kernelName, __ATOMIC_RELAXED); \
        hipLaunchKernelGGL((kernelName), __VA_ARGS__);                                       \
    } while (0)

// ---------------------------------------------------------------------------
// Uniform cell list over the float32-rounded cloud.
// Cell id = (cz * ny + cy) * nx + cx; cell coordinates are computed in fp64 so
// that the "distance to the stencil boundary >= ring * cell" guarantee holds.
// ---------------------------------------------------------------------------
struct pct_grid {
    double ox, oy, oz;      // origin (bbox min)
    double cell;            // edge length
    double inv_cell;
    int32_t nx, ny, nz;
    int64_t ncell;
    // Faces beyond which points of the cloud were left out of this grid (a sharded handle keeps only the points
    // near its owned range), in cell units from the origin; -inf / +inf = nothing was left out on that side.
    double lim_lo[3], lim_hi[3];
};

struct pct_buf {
    void* p = nullptr;
    size_t cap = 0;
};

struct pct_comm;            // RCCL communicator + exchange stream (pct_comm.hip); null on a single-GPU handle

struct pct_ctx {
    int device = 0;
    pct_comm* comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[9] = {};         // [0..7] stage timings; [8]: the cell list's scan totals are in pinned memory (pct_build_grid)
    // Streams of clouds (pct_set_async): a fused call returns once its kernels are enqueued; its timing events, its
    // statistics words and its SVD row count wait in the set of the call's parity until the NEXT fused call has passed
    // its mid-build wait (everything of the previous call has completed by then) or until anything else is asked of the
    // handle (which first waits for the stream).  The host prepares step i + 1 while the device fits step i.
    hipEvent_t ev_prev[9] = {};    // the timing events of the pending call while a new one records into ev
    bool async_mode = false;
    bool pending = false;          // a fused call has been enqueued and its bookkeeping has not been done
    int pend_par = 0;              // parity of the pending call: which pinned slots its kernels write
    bool pend_sorted = false, pend_levels = false;
    int64_t pend_owned = 0;
    int fit_par = 0;               // which pinned slots (statistics mirror, SVD row count) the fits being launched write
    pct_timings tm_snap = {};      // the host-side fields of the pending call's timings
    pct_timings tm_done = {};      // timings of the last call whose bookkeeping has been done
    char err[512] = {0};

    int64_t n = 0;                 // cloud size (candidates)
    int64_t q_begin = 0, q_end = 0;
    // Points the grid holds: all n, or -- sharded handles -- the points inside the owned range's bounding box plus
    // a margin, compacted in public order (pts4.w keeps the public index).  g_begin = position of the first owned
    // point in that compacted array (== q_begin when nothing was left out).
    int64_t n_grid = 0, g_begin = 0;
    bool culled = false;           // the grid holds fewer than n points
    bool no_cull = false;          // set after a sweep met a query the kept points cannot answer
    float cull_box[6] = {0, 0, 0, 0, 0, 0};    // box of the last sharded pack (reused for the next similar cloud)
    bool cull_box_valid = false;
    int64_t cull_box_n = 0, cull_box_q0 = 0, cull_box_q1 = 0;
    int32_t retries = 0;
    // Ownership by slab (pct_set_query_slab): the cloud is cut into slab_parts slabs of equal population along its
    // longest axis and this handle answers the points of slab_part.  The cut comes from a 4096-bin histogram every
    // rank computes alike from the gathered cloud; q_begin / q_end then refer to the PACKED array (owned first).
    int32_t slab_part = 0, slab_parts = 0;
    int32_t slab_axis = 0, slab_bin_lo = 0, slab_bin_hi = 0;
    float slab_x0 = 0, slab_inv = 0;
    float slab_bbox[6] = {0, 0, 0, 0, 0, 0};
    int64_t slab_counts[PCT_SLAB_PARTS_MAX] = {};
    bool slab_split_valid = false;
    double hint_edge = 0, hint_guess = 0, hint_target = 0;   // warm start of the cell-size search (pct_build_grid)
    float spec_bbox[6] = {0, 0, 0, 0, 0, 0};   // grid box of the last plain build (trimmed), reused speculatively
    float spec_raw[6] = {0, 0, 0, 0, 0, 0};    // raw bounding box of that cloud
    int64_t spec_n = 0;
    bool spec_valid = false;
    // density-adaptive sweep (pct_levels.hip): the queries one pass could not answer are re-owned by the next
    // pass, which sizes its cells for THEM
    const void* own_flag = nullptr; // device float (n): wanted log2 cell edge of every unanswered point (NaN = answered); null = ownership by index range
    float own_lo = 0, own_hi = 0;   // the pass owns the points with own_lo <= wanted < own_hi
    float level_box[6] = {0, 0, 0, 0, 0, 0};   // bbox of the owned points of a fast level pass
    bool level_box_valid = false;
    int64_t own_count = 0;
    double level_edge = 0;          // > 0: first cell edge of this pass
    bool level_mode = false;        // a density-adaptive sweep is in progress (rows answered are recorded in row_done)
    bool levels_fuse_fit = false;   // pct_curvature: every pass fits the rows it answered (in its own cell order)
    bool levels_fitted = false;     // ... and did: pct_launch_fit_table has nothing left to do
    pct_buf lvl_src;                // float4 (n): the points in the cell order of the first pass (input of the later passes' builds)
    bool lvl_src_valid = false;
    float lvl_bbox[6] = {0, 0, 0, 0, 0, 0};
    // hierarchical cell list (pct_tree.hip)
    int tree_bits = 0;
    int64_t tree_segs = 0;
    double tree_two_level_share = 1.0;   // share of the points on the two most populated adjacent octree levels (same sample)
    int tree_level_spread = 0;      // octree levels between the 5th and 95th percentile of the points (a sample of the segments)
    pct_buf tree_codes, tree_vals;  // u64 / u32 (2 n): Morton codes and public positions, unsorted | sorted
    pct_buf tree_lvl;               // u8 (n): octree level every point is served at
    pct_buf tree_head, tree_marks;  // segment / item marks and their scans
    pct_buf tree_seg, tree_runs;    // int4 per segment {level, cx, cy, cz}; int2 x 27 per segment {first position, points}
    pct_buf tree_range;             // int2 per segment {first position, points} + int per segment: stencil population + device totals
    pct_buf tree_bucket;            // int (2^18 + 2): first position of every 18-bit code prefix (pct_code_lower_bound)
    pct_buf tree_tmp;
    pct_buf row_done;               // int32 (rows of the pass)
    pct_buf redo_m;                 // int32, parallel to redo: stencil population of the row's item
    pct_buf flag_buf;               // float (n): wanted log2 cell edge of every point still unanswered (NaN = answered)
    pct_buf dens_buf;               // float2 (n): log2 of the largest edge known too small / the smallest known too large
    pct_buf pub_pos, pub_dist, pub_cnt;   // public-space neighbour table the passes are merged into
    bool uneven = false;
    bool auto_probe = false;        // PCT_KNN_AUTO on a cloud the hierarchical list could take: pct_build_grid gives up (grid_skewed)
    bool auto_probe_tree = false;   // ... and the census that follows may send the call there
    bool grid_skewed = false;       // rather than build a uniform list of more than 16 cells per point
    float auto_tree_bbox[6] = {0, 0, 0, 0, 0, 0};   // ... whose bounding box was this (the next cloud must match it within 2 %)
    float tree_bbox[6] = {0, 0, 0, 0, 0, 0};        // bounding box of the cloud the hierarchical list was last built for
    bool tree_check_bbox = false, tree_hint_mismatch = false;
    int64_t auto_tree_n = 0;        // PCT_KNN_AUTO sent a cloud of this size to the hierarchical list: the next one of the same
    int32_t auto_tree_calls = 0;    // size goes there directly (no uniform build first); re-examined every 16th call
    bool last_levels = false;       // the table in place came from pct_knn_levels            // a plain grid sweep of this cloud left > 5 % of the queries to the exact kernel
    bool has_f64 = false;
    double occupancy_factor = 0.0; // 0 = default
    bool collect_stats = false;    // sweep statistics (costly same-address atomics)

    // coordinates
    pct_buf xyz;        // float  (n,3) public order (owned copy)
    const float* xyz_view = nullptr;   // the coordinates in use: xyz.p, or a caller's buffer (pct_use_points_device_f32)
    pct_buf pts4;       // float4 (n_grid) public order, w = public index bits
    pct_buf pts4d;      // double4 (n) public order (only when has_f64), w = index
    // grid
    pct_grid grid = {};
    pct_buf cell_of;    // int32 (n) cell id per public point
    pct_buf cell_cnt;   // int32 (ncell+1) exclusive cell starts in the sorted cloud
    pct_buf cell_own;   // int32 (ncell) owned points per cell (stored first inside the cell)
    pct_buf cell_oth;   // int32 (ncell) other points per cell (sharded handles only)
    pct_buf own_start;  // int32 (ncell+1) first neighbour-table row of every cell
    pct_buf cell_fill;  // int32 (n) arrival rank of every point in its cell and class
    pct_buf scan_tmp;   // block sums
    pct_buf occ;        // int2 (n_items) work items {cell id, chunk of items_q queries}
    pct_buf redo;       // int32 (n) queries the fast sweep handed to the exact sweep
    int64_t n_items = 0;
    int64_t nonempty_cells = 0;    // cells of the current cell list that hold at least one point
    int32_t items_q = 12;
    pct_buf sorted4;    // float4 (n) cell-sorted, w = public index bits
    pct_buf sorted4d;   // double4 (n) cell-sorted native coords (has_f64)
    pct_buf row_of;     // int32 (owned) neighbour-table row of public index q_begin + i
    pct_buf owned_pos;  // int32 (owned) sorted position of every table row
    pct_buf red;        // small reduction scratch
    // 4 KiB of pinned, device-visible host memory: kernels drop their few result words here so that a
    // read-back is one stream synchronisation, not a copy command.  [0,128) PackRed  [128,192) scan totals
    // [192,256) sweep counters  [256,1024) and [1024,1056) band statistics / band box of the density-adaptive sweep  [2048,2056) rows of the last fit that went to k_fit_svd  [2112,2144) work-item census of PCT_KNN_AUTO  [2176,2312) totals of the tree build
    unsigned char* pin = nullptr;
    int64_t n_occ = 0;
    bool grid_valid = false;
    bool pts4_valid = false;

    // neighbour table: one row per OWNED query (cell order, see own_start); entries are sorted positions
    pct_buf nbr_pos;    // int32 (n,k) sorted positions of the neighbours
    pct_buf nbr_dist;   // float (n,k)
    pct_buf nbr_cnt;    // int32 (n)
    int32_t k = 0;
    int32_t nbr_pitch = 0;   // row pitch of nbr_pos / nbr_dist in elements (k rounded up to 4)
    double eps = 0.0;
    bool counters_clean = false;   // the sweep's statistics words were zeroed by the cell-list build just enqueued (k_scan_tiles)
    int fit_parity = 0;            // which of the two counts of fit_flag the next fit launch uses (the launch zeroes the other one)
    void* fit_flag_seen = nullptr; // the fit_flag allocation (pointer and capacity) whose head has been zeroed
    size_t fit_flag_cap_seen = 0;
    bool stats_mirror_req = false; // pct_curvature: the fit about to be launched may mirror the sweep's statistics words
    bool stats_mirrored = false;   // the fused fit copied the sweep's statistics words to pinned memory (no D2H copy needed)
    bool knn_valid = false;
    bool knn_sorted_space = false; // false: rows/ids are public indices (brute force)
    bool skip_dist_req = false;    // the caller will not read distances from the table (the fused curvature call)
    bool dist_valid = true;        // nbr_dist holds the distances of the table in place (else: derived on demand)
    pct_buf counters;   // int64[4] device counters (fallbacks, overflows)

    // results, public order
    pct_buf coefs;      // float (n,6)
    pct_buf K, H, H2;   // float (n)
    int64_t fit_rows = 0;
    bool fit_valid = false;
    bool fit_row_order = false;    // results are in neighbour-table row order (grid sweep), not public order
    bool fit_cloud_aligned = false;// results belong to cloud rows [q_begin, q_end) (pct_fit / pct_curvature) rather than to the
                                   // rows of a pct_fit_indices call; kept apart from knn_valid, which helpers may clear

    pct_buf fit_flag;   // int: [0] number of rows k_fit handed to k_fit_svd, [16..] the rows
    // staging for downloads / host-index fits
    pct_buf stage_a, stage_b, stage_c, stage_d;
    pct_buf qpts4;      // float4 {x,y,z,index} of EVERY point in public order, for pct_query_points (built on first use)
    bool qpts4_valid = false;

    pct_timings tm = {};
};

int pct_fail(pct_ctx* ctx, int code, const char* fmt, ...);
// getenv() for the library's switches (PCT_*), cached: a fused call used to make ~20 getenv() scans of the environment
// (~6 us per step).  The cache is dropped whenever the environment block changes (a fingerprint of its string pointers,
// ~50 loads per lookup), so tests and tools that flip a switch between calls keep working.  `name` must be a literal.
const char* pct_getenv(const char* name);
void pct_comm_release(pct_ctx* ctx);

// The fast sweep sorts <= 64 survivors in one register per lane (R = 1) or <= 128 in two (R = 2).  R = 1 would hold
// k + 1 <= 64, but near that limit the window k+1 <= count <= 64 for the threshold gets narrow and the larger cells
// overflow the 512-slot staging area: from k + 1 > kFastR1Max on, R = 2 (768 slots, window up to 128) is faster.
// (Round 3, k_knn_pair against k_knn_duo on the 1 M torus: 0.51 | 0.60 ms at k = 56, 0.59 | 0.60 at 60, 0.80 | 0.61 at 63.)
#ifndef PCT_FAST_R1_MAX
#define PCT_FAST_R1_MAX 61
#endif
inline int pct_fast_r1_max() {
    static const int v = [] { const char* e = getenv("PCT_FAST_R1_MAX"); const int x = e ? atoi(e) : PCT_FAST_R1_MAX; return x < 2 ? 2 : x > 64 ? 64 : x; }();
    return v;
}
// cell occupancy (points sharing a point's cell) the grid is sized for, as a multiple of k + 1 (tools/tune_factor.py)
// R = 1: the optimum is set by the 512 staged slots, not by k: ~27 points per cell for k <= 47 (measured optimum
// 2.4 (k+1) at k = 10, 1.2-1.4 at 20, 0.85 at 30, 0.55-0.60 at 40), rising to ~29.5 from k = 52 on (0.55 at 50, 0.50 at
// 56 and 60): larger cells mean fewer work items and ring fallbacks until the 27-cell stencil of a surface outgrows
// the staging area.
inline double pct_default_factor(int k) {
    if (k + 1 <= pct_fast_r1_max()) {
        const double per_cell = k <= 47 ? 27.0 : k >= 52 ? 29.5 : 27.0 + 0.5 * (k - 47);
        return per_cell / (k + 1);
    }
    const double n = k + 1;                           // R = 2 (768 slots): 0.52 up to k ~ 84, 0.45 at 100, 0.40 at 127
    if (n > 128) return 0.35;                         // wave-per-query exact sweep only: ~3 (k+1) candidates in the 27 cells of a surface
    return n <= 85 ? 0.52 : n <= 101 ? 0.52 - 0.07 * (n - 85) / 16.0 : 0.45 - 0.05 * (n - 101) / 27.0;
}
int pct_reserve(pct_ctx* ctx, pct_buf* b, size_t bytes);
void pct_release(pct_buf* b);

// Morton codes of the hierarchical cell list: 21 bits per axis, x in bit 0 of every triple
__host__ __device__ inline unsigned long long pct_spread3(unsigned v) {         // bit i -> bit 3 i
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__host__ __device__ inline unsigned pct_compact3(unsigned long long x) {        // bit 3 i -> bit i
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return (unsigned)x;
}
// offset of stencil cell t (0..26) from the centre cell; t = 0 is the centre (staged / scanned first)
__host__ __device__ inline void pct_stencil_cell(int t, int* dx, int* dy, int* dz) {
    const int i = t == 0 ? 13 : t == 13 ? 0 : t;
    *dx = i % 3 - 1;
    *dy = (i / 3) % 3 - 1;
    *dz = i / 9 - 1;
}
// first position of the Morton-sorted code array whose code is >= key
__device__ inline int64_t pct_code_lower_bound(const unsigned long long* __restrict__ codes, int64_t n, unsigned long long key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (codes[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// The same through the bucket table of the tree build: bucket[p] = first position whose code >= p << kTreeBucketShift
// (2^18 + 2 entries).  A search is a chain of dependent loads, ~20 over a million codes; the table leaves the two or
// three inside one bucket (the refinement of a segment next to a much denser region is a chain of such searches).
constexpr int kTreeBucketBits = 18, kTreeBucketShift = 63 - kTreeBucketBits;
__device__ inline int64_t pct_code_lower_bound(const unsigned long long* __restrict__ codes, const int* __restrict__ bucket,
                                               unsigned long long key) {
    const unsigned p = (unsigned)(key >> kTreeBucketShift);
    int64_t lo = bucket[p], hi = bucket[p + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (codes[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#define PCT_HIP(ctx, call)                                                        \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return pct_fail((ctx), e_ == hipErrorOutOfMemory ? PCT_ERR_OOM : PCT_ERR_HIP, \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define PCT_TRY(expr)            \
    do {                         \
        int s_ = (expr);         \
        if (s_ != PCT_OK) return s_; \
    } while (0)

// grid build (pct_grid.hip)
int pct_pack_points(pct_ctx* ctx, float* bbox6);
int pct_pack_points_f64(pct_ctx* ctx, const double* d_xyz64);
int pct_build_grid(pct_ctx* ctx, int32_t k, double eps);
// neighbour sweeps (pct_knn.hip)
int pct_launch_knn_grid(pct_ctx* ctx, int32_t k, double eps, bool exact_only, int phase = 0);
int pct_launch_knn_brute(pct_ctx* ctx, int32_t k, double eps);
// census of the work items of the cell list in place: out4 = {queries, queries whose stencil overflows the staging
// area, queries whose stencil holds fewer than 2.5 (k+1) points (too few to vouch for k+1 within one cell edge), sum over the other queries of the non-empty cells in their
// 27-cell stencil} -- what decides between the plain and the density-adaptive sweep before anything is swept
int pct_item_census(pct_ctx* ctx, int32_t k, unsigned long long out4[4]);
int pct_knn_levels(pct_ctx* ctx, int32_t k, double eps);     // pct_levels.hip
int pct_build_tree(pct_ctx* ctx, int32_t k, double eps, bool* usable);     // pct_tree.hip (usable = false: not a cloud for it, nothing was built)
int pct_launch_knn_tree(pct_ctx* ctx, int32_t k, double eps);
int pct_launch_export_neighbors(pct_ctx* ctx, int64_t begin, int64_t end,
                                int32_t* d_idx, float* d_dist, int32_t* d_cnt);
// fit (pct_fit.hip)
int pct_launch_fit_table(pct_ctx* ctx);
int pct_launch_fit_pass(pct_ctx* ctx, int64_t rows);
int pct_launch_fit_rows(pct_ctx* ctx, const int32_t* d_idx, const int32_t* d_cnt,
                        const int64_t* d_query, int64_t rows, int32_t k, int32_t pitch,
                        float* d_coefs, float* d_K, float* d_H, float* d_H2, bool sorted_space);
int pct_launch_prefix_rows(pct_ctx* ctx, const int* d_sample_pos, int64_t n_samples, int n_lo, int n_hi, int* d_table,
                           int pitch, int* d_cnt, int64_t* d_row_query);
int pct_launch_gather_fit(pct_ctx* ctx, int64_t first, int64_t rows, float* d_coefs, float* d_K, float* d_H, float* d_H2);
int pct_launch_curvatures(pct_ctx* ctx, const float* d_coefs, int64_t rows, float* d_K, float* d_H, float* d_H2);
int pct_launch_plane_rotate(pct_ctx* ctx, const void* d_nbrs, bool f64, int64_t batch, int32_t m, double* d_out);
int pct_launch_quadric_rows(pct_ctx* ctx, const float* d_pts, int64_t batch, int32_t m, float* d_coefs);
int pct_launch_selftest(pct_ctx* ctx, int* d_fails);
int pct_ensure_plain_records(pct_ctx* ctx);
int pct_launch_fit_rows_f64(pct_ctx* ctx, const int32_t* d_idx, const int32_t* d_cnt, const int64_t* d_query, int64_t rows,
                            int32_t k, int32_t pitch, double* d_coefs, double* d_K, double* d_H);
int pct_launch_query_points(pct_ctx* ctx, const double* d_q, int64_t m, int32_t k, double eps, int32_t* d_idx, double* d_dist);
int pct_launch_gather_int(pct_ctx* ctx, const int* d_map, int* d_inout, int64_t n);
int pct_launch_export_rows(pct_ctx* ctx, const int64_t* d_rows, int64_t n_rows, int32_t* d_idx, float* d_dist, int32_t* d_cnt);
int pct_launch_mesh_energies(pct_ctx* ctx, const double* d_v, const int* d_tri, int64_t n_tri, const void* d_K, const void* d_H,
                             bool f64, double* d_partial, int nblk, double* d_out);
int pct_voxel_downsample_device(pct_ctx* ctx, const void* d_xyz, bool f64, int64_t n, double voxel, int64_t* d_out, int64_t* count);
int pct_launch_surface_variation(pct_ctx* ctx, float* d_out);
