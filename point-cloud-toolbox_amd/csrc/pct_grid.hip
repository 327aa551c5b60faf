// Uniform cell list build for the neighbour sweep (the search structure that
// stands where the reference builds scipy's cKDTree, pointCloudToolbox.py:74).
//
//   pack      xyz (n,3) f32 -> float4 {x,y,z,index}; finite check; bbox partials
//   hist      cell id per point + per-cell counts of owned / other points (integer atomics)
//   occupancy mean points-per-cell as seen by a point (drives the cell size)
//   scan      exclusive scan of the counts -> cell starts, ordered occupied list
//   scatter   counting sort of the float4 records into cell order, owned points first in each cell
//
// All kernels are HBM/L2 streaming passes over 16 B records with 64-wide waves.
#include "pct_internal.h"

#include <math.h>
#include <stdlib.h>

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// red layout: [0..5] = min xyz, max xyz as ordered ints ; [6] = non-finite flag
__device__ __forceinline__ int float_order(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float order_float(int i) {
    int j = i >= 0 ? i : i ^ 0x7fffffff;
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(j);
#else
    float f;
    memcpy(&f, &j, 4);
    return f;
#endif
}

__global__ __launch_bounds__(kBlock) void k_pack(const float* __restrict__ xyz, int64_t n,
                                                 float4* __restrict__ pts4, int* __restrict__ red) {
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        bad |= !(isfinite(x) && isfinite(y) && isfinite(z));
        pts4[i] = make_float4(x, y, z, __int_as_float((int)i));
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
    }
    __shared__ float s_mn[kBlock / 64][3], s_mx[kBlock / 64][3];
    __shared__ int s_bad[kBlock / 64];
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_min(mn[a]);
        mx[a] = wave_max(mx[a]);
    }
    bad = __any(bad);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; }
        s_bad[w] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 3) {      // one atomic per block and component (same-address atomics serialise)
        const int a = threadIdx.x;
        float lo = s_mn[0][a], hi = s_mx[0][a];
        for (int i = 1; i < kBlock / 64; ++i) { lo = fminf(lo, s_mn[i][a]); hi = fmaxf(hi, s_mx[i][a]); }
        atomicMin(&red[a], float_order(lo));
        atomicMax(&red[3 + a], float_order(hi));
    }
    if (threadIdx.x == 3) {
        int b = 0;
        for (int i = 0; i < kBlock / 64; ++i) b |= s_bad[i];
        if (b) atomicOr(&red[6], 1);
    }
}

// double4 variant: native float64 coordinates ride along (w = index).
__global__ __launch_bounds__(kBlock) void k_pack_f64(const double* __restrict__ xyz, int64_t n,
                                                     double4* __restrict__ pts4d, float* __restrict__ xyz32) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        pts4d[i] = make_double4(x, y, z, (double)i);
        xyz32[3 * i + 0] = (float)x;   // tree coordinates are float32-rounded (pct:74)
        xyz32[3 * i + 1] = (float)y;
        xyz32[3 * i + 2] = (float)z;
    }
}

__device__ __forceinline__ int cell_coord(double x, double o, double inv, int n) {
    int c = (int)floor((x - o) * inv);
    return min(max(c, 0), n - 1);
}

// Cell id per point and per-cell populations.  Points this handle owns (public
// index in [q_begin, q_end), the multi-GPU shard) and the others are counted
// separately: inside a cell the owned points are stored first, so that work
// items and neighbour-table rows exist for owned queries only.  The value an
// integer atomic returns is the point's arrival rank in its class -- the
// scatter then needs no second atomic pass.
__global__ __launch_bounds__(kBlock) void k_hist(const float4* __restrict__ pts4, int64_t n, pct_grid g, int q_begin, int q_end,
                                                 int* __restrict__ cell_of, int* __restrict__ rank_of,
                                                 int* __restrict__ cell_own, int* __restrict__ cell_oth) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = pts4[i];
    int cx = cell_coord((double)p.x, g.ox, g.inv_cell, g.nx);
    int cy = cell_coord((double)p.y, g.oy, g.inv_cell, g.ny);
    int cz = cell_coord((double)p.z, g.oz, g.inv_cell, g.nz);
    int c = (cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    if (i >= q_begin && i < q_end)
        rank_of[i] = atomicAdd(&cell_own[c], 1);
    else
        rank_of[i] = atomicAdd(&cell_oth[c], 1) | (int)0x80000000;
}

// ---- triple exclusive scan over the cells ---------------------------------
//   x: all points      -> cell_start   (position of the cell in the sorted cloud)
//   y: work items      -> item rank    (ceil(owned / items_q) per cell)
//   z: owned points    -> own_start    (first neighbour-table row of the cell)
constexpr int kScanItems = 8;                    // per thread
constexpr int kScanTile = kBlock * kScanItems;   // 2048 cells per block

__device__ __forceinline__ int4 add3(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, 0); }

__device__ __forceinline__ int4 cell_counts(const int* __restrict__ own, const int* __restrict__ oth, int64_t c, int64_t ncell, int items_q) {
    if (c >= ncell) return make_int4(0, 0, 0, 0);
    const int o = own[c];
    const int t = o + (oth ? oth[c] : 0);
    return make_int4(t, (o + items_q - 1) / items_q, o, 0);
}

// first pass: per-tile sums; also accumulates sum_c count_c^2 (= sum over points of the population of
// their own cell), the statistic the cell-size loop steers on
__global__ __launch_bounds__(kBlock) void k_scan_sums(const int* __restrict__ own, const int* __restrict__ oth, int64_t ncell, int items_q,
                                                      int4* __restrict__ tmp, unsigned long long* __restrict__ sq_part) {
    __shared__ int4 sh[kBlock / 64];
    __shared__ unsigned long long shq[kBlock / 64];
    unsigned long long sq = 0;
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int4 v = make_int4(0, 0, 0, 0);
    for (int j = 0; j < kScanItems; ++j) {
        const int4 x = cell_counts(own, oth, base + j, ncell, items_q);
        v = add3(v, x);
        sq += (unsigned long long)x.x * (unsigned)x.x;
    }
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o);
        sq += __shfl_xor(sq, o);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = v;
        shq[threadIdx.x >> 6] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int4 t = make_int4(0, 0, 0, 0);
        unsigned long long q = 0;
        for (int i = 0; i < kBlock / 64; ++i) { t = add3(t, sh[i]); q += shq[i]; }
        tmp[blockIdx.x] = t;
        sq_part[blockIdx.x] = q;        // per-tile partial: no same-address atomics (they serialise at ~88/us)
    }
}

// single block: exclusive scan of the per-tile sums; totals to tmp[nblk]
__global__ __launch_bounds__(1024) void k_scan_tiles(int4* __restrict__ tmp, int nblk, const unsigned long long* __restrict__ sq_part,
                                                      unsigned long long* __restrict__ sumsq) {
    __shared__ int4 sh[1024];
    __shared__ int4 carry;
    __shared__ unsigned long long shq[16];
    {   // sum of the per-tile sum-of-squares partials
        unsigned long long q = 0;
        for (int i = threadIdx.x; i < nblk; i += 1024) q += sq_part[i];
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if ((threadIdx.x & 63) == 0) shq[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long t = 0; for (int i = 0; i < 16; ++i) t += shq[i]; *sumsq = t; }
    }
    if (threadIdx.x == 0) carry = make_int4(0, 0, 0, 0);
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        int i = base + threadIdx.x;
        int4 v = i < nblk ? tmp[i] : make_int4(0, 0, 0, 0);
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int4 a = make_int4(0, 0, 0, 0);
            if ((int)threadIdx.x >= o) a = sh[threadIdx.x - o];
            __syncthreads();
            sh[threadIdx.x] = add3(sh[threadIdx.x], a);
            __syncthreads();
        }
        int4 incl = sh[threadIdx.x];
        int4 c = carry;
        if (i < nblk) tmp[i] = make_int4(c.x + incl.x - v.x, c.y + incl.y - v.y, c.z + incl.z - v.z, 0);
        __syncthreads();
        if (threadIdx.x == 1023) carry = add3(c, incl);
        __syncthreads();
    }
    if (threadIdx.x == 0) tmp[nblk] = carry;
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(const int* __restrict__ own, const int* __restrict__ oth, int64_t ncell, int items_q,
                                                       const int4* __restrict__ tmp, int* __restrict__ cell_start,
                                                       int* __restrict__ own_start, int2* __restrict__ items) {
    __shared__ int4 sh[kBlock];
    int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int4 x[kScanItems];
    int4 v = make_int4(0, 0, 0, 0);
    for (int j = 0; j < kScanItems; ++j) {
        x[j] = cell_counts(own, oth, base + j, ncell, items_q);
        v = add3(v, x[j]);
    }
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < kBlock; o <<= 1) {
        int4 a = make_int4(0, 0, 0, 0);
        if ((int)threadIdx.x >= o) a = sh[threadIdx.x - o];
        __syncthreads();
        sh[threadIdx.x] = add3(sh[threadIdx.x], a);
        __syncthreads();
    }
    const int4 off = tmp[blockIdx.x];
    int s = off.x + sh[threadIdx.x].x - v.x;
    int r = off.y + sh[threadIdx.x].y - v.y;
    int w = off.z + sh[threadIdx.x].z - v.z;
    for (int j = 0; j < kScanItems; ++j) {
        int64_t c = base + j;
        if (c < ncell) {
            cell_start[c] = s;
            own_start[c] = w;
            for (int ch = 0; ch < x[j].y; ++ch) items[r++] = make_int2((int)c, ch);
            s += x[j].x;
            w += x[j].z;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) { cell_start[ncell] = s; own_start[ncell] = w; }
}

// counting-sort scatter, owned points first inside every cell; also records for every public index its
// neighbour-table row (-1 if not owned) and for every row its sorted position
__global__ __launch_bounds__(kBlock) void k_scatter(const float4* __restrict__ pts4, const int* __restrict__ cell_of,
                                                    const int* __restrict__ cell_start, const int* __restrict__ cell_own,
                                                    const int* __restrict__ own_start, const int* __restrict__ rank_of,
                                                    int64_t n, float4* __restrict__ sorted4, int* __restrict__ row_of,
                                                    int* __restrict__ owned_pos,
                                                    const double4* __restrict__ pts4d, double4* __restrict__ sorted4d) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    const int r = rank_of[i];
    int pos, row = -1;
    if (r >= 0) {
        pos = cell_start[c] + r;
        row = own_start[c] + r;
        owned_pos[row] = pos;
    } else {
        pos = cell_start[c] + cell_own[c] + (r & 0x7fffffff);
    }
    sorted4[pos] = pts4[i];
    row_of[i] = row;
    if (pts4d) sorted4d[pos] = pts4d[i];
}

__global__ __launch_bounds__(256) void k_gather_int(const int* __restrict__ map, int* __restrict__ io, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) io[i] = map[io[i]];
}

int grid_1d(int64_t n, int per_block, int cap) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (cap > 0 && b > cap) b = cap;
    return (int)b;
}

}  // namespace

// ---------------------------------------------------------------------------
int pct_pack_points(pct_ctx* ctx, float* bbox /*6*/) {
    const int64_t n = ctx->n;
    PCT_TRY(pct_reserve(ctx, &ctx->pts4, (size_t)n * sizeof(float4)));
    PCT_TRY(pct_reserve(ctx, &ctx->red, 64));
    int init[8] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0, 0};
    PCT_HIP(ctx, hipMemcpyAsync(ctx->red.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pack, dim3(grid_1d(n, kBlock * 4, 512)), dim3(kBlock), 0, ctx->stream,
                       (const float*)ctx->xyz.p, n, (float4*)ctx->pts4.p, (int*)ctx->red.p);
    PCT_HIP(ctx, hipGetLastError());
    int out[8];
    PCT_HIP(ctx, hipMemcpyAsync(out, ctx->red.p, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (out[6]) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points");
    for (int a = 0; a < 6; ++a) bbox[a] = order_float(out[a]);
    ctx->pts4_valid = true;
    return PCT_OK;
}

int pct_pack_points_f64(pct_ctx* ctx, const double* d_xyz64) {
    const int64_t n = ctx->n;
    PCT_TRY(pct_reserve(ctx, &ctx->pts4d, (size_t)n * sizeof(double4)));
    PCT_TRY(pct_reserve(ctx, &ctx->xyz, (size_t)n * 3 * sizeof(float)));
    hipLaunchKernelGGL(k_pack_f64, dim3(grid_1d(n, kBlock, 2048)), dim3(kBlock), 0, ctx->stream,
                       d_xyz64, n, (double4*)ctx->pts4d.p, (float*)ctx->xyz.p);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

static void set_dims(pct_grid* g, const float* bbox, double a) {
    g->ox = bbox[0]; g->oy = bbox[1]; g->oz = bbox[2];
    g->cell = a;
    g->inv_cell = 1.0 / a;
    double ex = (double)bbox[3] - bbox[0], ey = (double)bbox[4] - bbox[1], ez = (double)bbox[5] - bbox[2];
    g->nx = (int32_t)floor(ex * g->inv_cell) + 1;
    g->ny = (int32_t)floor(ey * g->inv_cell) + 1;
    g->nz = (int32_t)floor(ez * g->inv_cell) + 1;
    g->ncell = (int64_t)g->nx * g->ny * g->nz;
}

// Chooses the cell edge so that a point shares its cell with about
// factor*(k+1) points, then counting-sorts the cloud.  With that occupancy the
// 27-cell stencil (guaranteed radius = one cell edge) contains the k+1 nearest
// points for nearly every query of a surface-like cloud; the sweep kernel
// widens ring by ring for the rest.
int pct_build_grid(pct_ctx* ctx, int32_t k, double eps) {
    const int64_t n = ctx->n;
    float bbox[6];
    PCT_TRY(pct_pack_points(ctx, bbox));

    // measured optima on surface clouds (tools/tune_factor.py): larger cells cost candidates, smaller ones cost
    // trips to the exact sweep; the LDS staging capacity caps the large side
    const double factor = ctx->occupancy_factor > 0 ? ctx->occupancy_factor : (k + 1 <= 32 ? 0.65 : k + 1 <= 64 ? 0.55 : 0.45);
    const double target = factor * (k + 1);
    const int64_t cell_cap = (int64_t)1 << 27;

    double ex = (double)bbox[3] - bbox[0], ey = (double)bbox[4] - bbox[1], ez = (double)bbox[5] - bbox[2];
    double emax = fmax(ex, fmax(ey, ez));
    if (!(emax > 0)) emax = 1.0;
    // first guess: the cloud is a surface whose area is about the bbox's half-surface * 1.2
    double area = 1.2 * (ex * ey + ey * ez + ex * ez);
    if (!(area > 0)) area = emax * emax;
    double a = sqrt(target * area / (double)n);
    if (!(a > 0) || !isfinite(a)) a = emax;
    a = fmin(a, emax * 1.0001 + 1e-30);

    PCT_TRY(pct_reserve(ctx, &ctx->cell_of, (size_t)n * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->cell_fill, (size_t)n * sizeof(int)));   // in-cell arrival ranks
    const bool sharded = ctx->q_begin > 0 || ctx->q_end < n;             // some points are candidates only
    int items_q = ctx->items_q > 0 && ctx->items_q <= 64 ? ctx->items_q : 12;
    if (const char* e = getenv("PCT_ITEMS_Q")) { const int v = atoi(e); if (v >= 1 && v <= 64) items_q = v; }   // tuning aid
    ctx->items_q = items_q;
    const int64_t n_owned = ctx->q_end - ctx->q_begin;
    PCT_TRY(pct_reserve(ctx, &ctx->sorted4, (size_t)n * sizeof(float4)));
    PCT_TRY(pct_reserve(ctx, &ctx->row_of, (size_t)n * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->owned_pos, (size_t)n_owned * sizeof(int)));
    if (ctx->has_f64) PCT_TRY(pct_reserve(ctx, &ctx->sorted4d, (size_t)n * sizeof(double4)));
    int nblk = 0;
    pct_grid g = {};
    double a_prev = 0, m_prev = 0;
    int iters = 0;
    const int max_iter = 6;
    int4 tot = make_int4(0, 0, 0, 0);
    // Every pass runs the whole build (histogram, scan, scatter) and only then reads back the occupancy statistic
    // together with the scan totals: the first cell size is accepted in the common case, which then costs ONE host
    // synchronisation instead of two; a rejected size costs a speculative scatter.
    for (int it = 0; it < max_iter; ++it) {
        if (eps > 0 && a > eps * 1.000001) a = eps * 1.000001;   // one ring already covers the eps ball
        set_dims(&g, bbox, a);
        while (g.ncell > cell_cap) {
            a *= cbrt((double)g.ncell / (double)cell_cap) * 1.01;
            set_dims(&g, bbox, a);
        }
        PCT_TRY(pct_reserve(ctx, &ctx->cell_own, (size_t)g.ncell * sizeof(int)));
        PCT_HIP(ctx, hipMemsetAsync(ctx->cell_own.p, 0, (size_t)g.ncell * sizeof(int), ctx->stream));
        if (sharded) {
            PCT_TRY(pct_reserve(ctx, &ctx->cell_oth, (size_t)g.ncell * sizeof(int)));
            PCT_HIP(ctx, hipMemsetAsync(ctx->cell_oth.p, 0, (size_t)g.ncell * sizeof(int), ctx->stream));
        }
        PCT_HIP(ctx, hipMemsetAsync(ctx->red.p, 0, 16, ctx->stream));
        hipLaunchKernelGGL(k_hist, dim3(grid_1d(n, kBlock, 0)), dim3(kBlock), 0, ctx->stream,
                           (const float4*)ctx->pts4.p, n, g, (int)ctx->q_begin, (int)ctx->q_end, (int*)ctx->cell_of.p,
                           (int*)ctx->cell_fill.p, (int*)ctx->cell_own.p, sharded ? (int*)ctx->cell_oth.p : nullptr);
        nblk = (int)((g.ncell + kScanTile - 1) / kScanTile);
        PCT_TRY(pct_reserve(ctx, &ctx->scan_tmp, (size_t)(nblk + 1) * sizeof(int4) + (size_t)nblk * sizeof(unsigned long long)));
        unsigned long long* sq_part = (unsigned long long*)((int4*)ctx->scan_tmp.p + nblk + 1);
        PCT_TRY(pct_reserve(ctx, &ctx->cell_cnt, (size_t)(g.ncell + 1) * sizeof(int)));
        PCT_TRY(pct_reserve(ctx, &ctx->own_start, (size_t)(g.ncell + 1) * sizeof(int)));
        PCT_TRY(pct_reserve(ctx, &ctx->occ, ((size_t)(n_owned < g.ncell ? n_owned : g.ncell) + (size_t)n_owned / items_q + 16) * sizeof(int2)));
        hipLaunchKernelGGL(k_scan_sums, dim3(nblk), dim3(kBlock), 0, ctx->stream,
                           (const int*)ctx->cell_own.p, sharded ? (const int*)ctx->cell_oth.p : nullptr, g.ncell, items_q,
                           (int4*)ctx->scan_tmp.p, sq_part);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, (int4*)ctx->scan_tmp.p, nblk,
                           (const unsigned long long*)sq_part, (unsigned long long*)ctx->red.p);
        hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(kBlock), 0, ctx->stream,
                           (const int*)ctx->cell_own.p, sharded ? (const int*)ctx->cell_oth.p : nullptr, g.ncell, items_q,
                           (const int4*)ctx->scan_tmp.p, (int*)ctx->cell_cnt.p, (int*)ctx->own_start.p, (int2*)ctx->occ.p);
        hipLaunchKernelGGL(k_scatter, dim3(grid_1d(n, kBlock, 0)), dim3(kBlock), 0, ctx->stream,
                           (const float4*)ctx->pts4.p, (const int*)ctx->cell_of.p, (const int*)ctx->cell_cnt.p,
                           (const int*)ctx->cell_own.p, (const int*)ctx->own_start.p, (const int*)ctx->cell_fill.p, n,
                           (float4*)ctx->sorted4.p, (int*)ctx->row_of.p, (int*)ctx->owned_pos.p,
                           ctx->has_f64 ? (const double4*)ctx->pts4d.p : nullptr,
                           ctx->has_f64 ? (double4*)ctx->sorted4d.p : nullptr);
        PCT_HIP(ctx, hipGetLastError());
        unsigned long long s2 = 0;
        PCT_HIP(ctx, hipMemcpyAsync(&s2, ctx->red.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        PCT_HIP(ctx, hipMemcpyAsync(&tot, (int4*)ctx->scan_tmp.p + nblk, sizeof(int4), hipMemcpyDeviceToHost, ctx->stream));
        PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ++iters;
        double m = (double)s2 / (double)n;
        bool eps_bound = eps > 0 && a >= eps;            // cannot grow past eps
        bool capped = g.ncell * 2 > cell_cap && m < target;
        if ((m >= 0.8 * target && m <= 1.3 * target) || it == max_iter - 1 || (eps_bound && m < target) ||
            capped || (a >= emax && m < target))
            break;
        double d = 2.0;
        if (a_prev > 0 && m != m_prev && a != a_prev) {
            d = log(m / m_prev) / log(a / a_prev);
            if (!(d >= 1.0)) d = 1.0;
            if (d > 3.0) d = 3.0;
        }
        a_prev = a;
        m_prev = m;
        double f = pow(target / m, 1.0 / d);
        f = fmin(fmax(f, 0.25), 4.0);
        a = fmin(a * f, emax * 1.0001 + 1e-30);
    }
    ctx->grid = g;
    ctx->tm.grid_iters = iters;
    ctx->tm.cells = g.ncell;
    ctx->tm.cell_size = g.cell;
    if (tot.x != n || tot.z != n_owned)
        return pct_fail(ctx, PCT_ERR_INVALID, "cell scan totals %d/%d != %lld/%lld", tot.x, tot.z, (long long)n, (long long)n_owned);
    ctx->n_items = tot.y;
    ctx->n_occ = tot.y;
    ctx->tm.occupied_cells = tot.y;
    ctx->grid_valid = true;
    return PCT_OK;
}

int pct_launch_gather_int(pct_ctx* ctx, const int* d_map, int* d_inout, int64_t n) {
    hipLaunchKernelGGL(k_gather_int, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_map, d_inout, n);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
