// Neighbour sweep kernels: what PointCloud.plant_kdtree computes with one
// cKDTree.query(point, k+1) per point (pointCloudToolbox.py:81-85).
//
// Semantics reproduced: candidates are the float32-rounded coordinates
// (pct:74); squared distances are accumulated in fp64 as ((dx*dx + dy*dy) +
// dz*dz) with no FMA contraction, which is bit-for-bit what SciPy evaluates
// for 3-D data; the k+1 smallest are taken, result 0 is dropped (pct:84-85),
// and sqrt(d2) is rounded to float32 (pct:78).  Exact-distance ties are ordered
// by public index so the result does not depend on the cell order.
//
// Mapping to CDNA4: one 64-lane wave owns one occupied grid cell.  It stages the
// 27-cell stencil once into LDS with coalesced 16 B/lane loads, then serves
// every query of the cell from LDS: 64 candidates per step, one fp64 distance
// per lane, a ballot against the running (k+1)-th distance, LDS compaction of
// the survivors, and a wave-wide bitonic sort/merge (cross-lane shuffles, no
// LDS traffic) whenever 64*R survivors are pending.  R = 1 holds k+1 <= 64,
// R = 2 holds k+1 <= 128.  Queries whose (k+1)-th distance exceeds the stencil's
// guaranteed radius widen the search shell by shell from global memory.
#include "pct_internal.h"

#include <math.h>

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kStageCap = 512;   // LDS-staged stencil candidates per wave

struct KnnArgs {
    const float4* pts;        // candidate records {x,y,z,public index}; cell-sorted (grid) or public order (brute)
    const double4* ptsd;      // native fp64 query coordinates in the same order (nullable)
    const int* cell_start;    // grid only
    const int* occ;           // grid only
    int64_t n_occ;
    int64_t n;
    pct_grid g;
    int k;
    double eps2;              // +inf when no bound
    int q_begin, q_end;       // owned public index range
    int* nbr_pos;
    float* nbr_dist;
    int* nbr_cnt;             // nullable
    unsigned long long* counters;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// LDS hand-off between lanes of ONE wave: no hardware barrier is needed (the
// wave issues its DS operations in order) but the compiler must not move LDS
// accesses across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int pub_index(const float4* pts, int pos) { return __float_as_int(pts[pos].w); }

// strict total order on (d2, public index); padding = (+inf, INT_MAX)
__device__ __forceinline__ bool key_less(double da, int pa, double db, int pb, const float4* pts) {
    if (da < db) return true;
    if (da > db || pa == pb) return false;
    if (pa == INT_MAX) return false;
    if (pb == INT_MAX) return true;
    return pub_index(pts, pa) < pub_index(pts, pb);
}

template <int R>
struct TopK {
    double d[R];
    int p[R];
};

// One compare-exchange level of the bitonic network over 64*R elements, element
// index i = lane + 64*slot.
template <int R>
__device__ __forceinline__ void bitonic_level(TopK<R>& t, int size, int stride, bool descending, const float4* pts) {
    const int lane = lane_id();
    if (stride >= 64) {
        const int ds = stride >> 6;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r & ds) == 0) {
                const int r2 = r | ds;
                if (r2 < R) {
                    const int i = lane + 64 * r;
                    const bool asc = ((i & size) == 0) != descending;
                    const bool hi_less = key_less(t.d[r2], t.p[r2], t.d[r], t.p[r], pts);
                    if (hi_less == asc) {
                        double td = t.d[r]; t.d[r] = t.d[r2]; t.d[r2] = td;
                        int tp = t.p[r]; t.p[r] = t.p[r2]; t.p[r2] = tp;
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double pd = __shfl_xor(t.d[r], stride);
            const int pp = __shfl_xor(t.p[r], stride);
            const int i = lane + 64 * r;
            const bool asc = ((i & size) == 0) != descending;
            const bool keep_min = ((lane & stride) == 0) == asc;
            const bool partner_less = key_less(pd, pp, t.d[r], t.p[r], pts);
            if (keep_min == partner_less) {
                t.d[r] = pd;
                t.p[r] = pp;
            }
        }
    }
}

template <int R>
__device__ __forceinline__ void bitonic_sort(TopK<R>& t, bool descending, const float4* pts) {
#pragma unroll
    for (int size = 2; size <= 64 * R; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) bitonic_level<R>(t, size, stride, descending, pts);
    }
}

// best (ascending) <- smallest 64*R of best U batch; batch must be descending
template <int R>
__device__ __forceinline__ void bitonic_merge_min(TopK<R>& best, const TopK<R>& batch, const float4* pts) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (key_less(batch.d[r], batch.p[r], best.d[r], best.p[r], pts)) {
            best.d[r] = batch.d[r];
            best.p[r] = batch.p[r];
        }
    }
#pragma unroll
    for (int stride = 32 * R; stride > 0; stride >>= 1) bitonic_level<R>(best, 64 * R, stride, false, pts);
}

// Per-wave running state for one query.
template <int R>
struct Sweep {
    TopK<R> best;
    double tau_d;     // current (k+1)-th smallest squared distance (+inf until known)
    int tau_p;
    int npend;
    bool empty;       // best holds no real element yet
    double qx, qy, qz;
    double eps2;
    int k;
    const float4* pts;
    double* pend_d;   // LDS, 64*R + 64 entries
    int* pend_p;

    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            best.d[r] = INFINITY;
            best.p[r] = INT_MAX;
        }
        tau_d = INFINITY;
        tau_p = INT_MAX;
        npend = 0;
        empty = true;
    }

    __device__ __forceinline__ void refresh_tau() {
        const int slot = k >> 6, src = k & 63;
        double d = best.d[0];
        int p = best.p[0];
#pragma unroll
        for (int r = 1; r < R; ++r)
            if (slot == r) { d = best.d[r]; p = best.p[r]; }
        tau_d = __shfl(d, src);
        tau_p = __shfl(p, src);
    }

    // take up to 64*R pending survivors into the running list
    __device__ __forceinline__ void flush() {
        const int lane = lane_id();
        TopK<R> b;
        wave_lds_sync();
        const int take = npend < 64 * R ? npend : 64 * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            const bool v = i < take;
            b.d[r] = v ? pend_d[i] : INFINITY;
            b.p[r] = v ? pend_p[i] : INT_MAX;
        }
        const int rest = npend - take;    // < 64
        double md = 0;
        int mp = 0;
        if (lane < rest) { md = pend_d[take + lane]; mp = pend_p[take + lane]; }
        wave_lds_sync();
        if (lane < rest) { pend_d[lane] = md; pend_p[lane] = mp; }
        wave_lds_sync();
        npend = rest;
        if (empty) {
            bitonic_sort<R>(b, false, pts);
            best = b;
            empty = false;
        } else {
            bitonic_sort<R>(b, true, pts);
            bitonic_merge_min<R>(best, b, pts);
        }
        refresh_tau();
    }

    // one candidate per lane
    __device__ __forceinline__ void consider(float4 c, int pos, bool valid) {
        const double dx = (double)c.x - qx, dy = (double)c.y - qy, dz = (double)c.z - qz;
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        const bool pass = valid && d2 < eps2 && key_less(d2, pos, tau_d, tau_p, pts);
        const unsigned long long m = __ballot(pass);
        if (m == 0) return;
        if (pass) {
            const int slot = npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            pend_d[slot] = d2;
            pend_p[slot] = pos;
        }
        npend += __popcll(m);
        if (npend >= 64 * R) flush();
    }

    __device__ __forceinline__ void scan_global(int s, int e) {
        const int lane = lane_id();
        for (int base = s; base < e; base += 64) {
            const int pos = base + lane;
            const bool valid = pos < e;
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) c = pts[pos];
            consider(c, pos, valid);
        }
    }

    __device__ __forceinline__ void finish() {
        if (npend > 0 || empty) flush();
    }

    // rows: element i (1..k) -> output column i-1
    __device__ __forceinline__ void store(int64_t row, int* nbr_pos, float* nbr_dist, int* nbr_cnt) {
        const int lane = lane_id();
        int found = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            const bool real = best.p[r] != INT_MAX;
            if (i >= 1 && i <= k) {
                nbr_pos[row * k + (i - 1)] = real ? best.p[r] : -1;
                nbr_dist[row * k + (i - 1)] = real ? (float)sqrt(best.d[r]) : INFINITY;
                found += real;
            }
        }
        if (nbr_cnt) {
            for (int o = 32; o > 0; o >>= 1) found += __shfl_xor(found, o);
            if (lane == 0) nbr_cnt[row] = found;
        }
    }
};

// ---------------------------------------------------------------------------
// Grid sweep: wave = occupied cell
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_knn_grid(KnnArgs a) {
    __shared__ float4 s_cand[kWavesPerBlock][kStageCap];
    __shared__ int s_pos[kWavesPerBlock][kStageCap];
    __shared__ double s_pend_d[kWavesPerBlock][64 * R + 64];
    __shared__ int s_pend_p[kWavesPerBlock][64 * R + 64];

    const int w = threadIdx.x >> 6;
    const int lane = lane_id();
    const int64_t cell_slot = (int64_t)blockIdx.x * kWavesPerBlock + w;
    if (cell_slot >= a.n_occ) return;

    const pct_grid g = a.g;
    const int cell = a.occ[cell_slot];
    const int cx = cell % g.nx;
    const int cy = (cell / g.nx) % g.ny;
    const int cz = cell / (g.nx * g.ny);
    const int qs = a.cell_start[cell], qe = a.cell_start[cell + 1];

    // ---- stage the 27-cell stencil (9 x-runs of <= 3 consecutive cells) ----
    float4* cand = s_cand[w];
    int* cpos = s_pos[w];
    int m = 0;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    for (int dz = -1; dz <= 1; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= g.nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = cy + dy;
            if (y < 0 || y >= g.ny) continue;
            const int row = (z * g.ny + y) * g.nx;
            const int s = a.cell_start[row + x0], e = a.cell_start[row + x1 + 1];
            for (int base = s; base < e; base += 64) {
                const int pos = base + lane;
                const int slot = m + (pos - s);
                if (pos < e && slot < kStageCap) {
                    cand[slot] = a.pts[pos];
                    cpos[slot] = pos;
                }
            }
            m += e - s;
        }
    }
    wave_lds_sync();
    const bool staged = m <= kStageCap;
    if (!staged && lane == 0) atomicAdd(&a.counters[1], 1ull);

    Sweep<R> sw;
    sw.k = a.k;
    sw.eps2 = a.eps2;
    sw.pts = a.pts;
    sw.pend_d = s_pend_d[w];
    sw.pend_p = s_pend_p[w];

    const double cell_len = g.cell * (1.0 - 1e-6);
    unsigned long long fallbacks = 0;

    for (int q = qs; q < qe; ++q) {
        const float4 qp = a.pts[q];
        const int pub = __float_as_int(qp.w);
        if (pub < a.q_begin || pub >= a.q_end) continue;
        if (a.ptsd) {
            const double4 qd = a.ptsd[q];
            sw.qx = qd.x; sw.qy = qd.y; sw.qz = qd.z;
        } else {
            sw.qx = (double)qp.x; sw.qy = (double)qp.y; sw.qz = (double)qp.z;
        }
        sw.reset();

        if (staged) {
            for (int base = 0; base < m; base += 64) {
                const int slot = base + lane;
                const bool valid = slot < m;
                float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                int pos = 0;
                if (valid) { c = cand[slot]; pos = cpos[slot]; }
                sw.consider(c, pos, valid);
            }
        } else {
            for (int dz = -1; dz <= 1; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z >= g.nz) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.ny) continue;
                    const int row = (z * g.ny + y) * g.nx;
                    sw.scan_global(a.cell_start[row + x0], a.cell_start[row + x1 + 1]);
                }
            }
        }
        sw.finish();

        // ---- widen shell by shell until the guarantee radius covers the answer
        int ring = 1;
        while (true) {
            const double need = fmin(sw.tau_d, sw.eps2);
            const double rr = ring * cell_len;
            if (need <= rr * rr) break;
            const bool covered = cx - ring <= 0 && cx + ring >= g.nx - 1 && cy - ring <= 0 && cy + ring >= g.ny - 1 &&
                                 cz - ring <= 0 && cz + ring >= g.nz - 1;
            if (covered) break;
            ++ring;
            if (ring == 2) ++fallbacks;
            const int xa = cx - ring, xb = cx + ring;
            const int xa_c = max(xa, 0), xb_c = min(xb, g.nx - 1);
            for (int dz = -ring; dz <= ring; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z >= g.nz) continue;
                for (int dy = -ring; dy <= ring; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y >= g.ny) continue;
                    const int row = (z * g.ny + y) * g.nx;
                    if (dz == -ring || dz == ring || dy == -ring || dy == ring) {
                        sw.scan_global(a.cell_start[row + xa_c], a.cell_start[row + xb_c + 1]);
                    } else {
                        if (xa >= 0) sw.scan_global(a.cell_start[row + xa], a.cell_start[row + xa + 1]);
                        if (xb < g.nx) sw.scan_global(a.cell_start[row + xb], a.cell_start[row + xb + 1]);
                    }
                }
            }
            sw.finish();
        }
        sw.store(q, a.nbr_pos, a.nbr_dist, a.nbr_cnt);
    }
    if (fallbacks && lane == 0) atomicAdd(&a.counters[0], fallbacks);
}

// ---------------------------------------------------------------------------
// Exhaustive sweep: wave = query, candidates streamed from global in public
// order (coalesced 1 KiB per wave instruction).  Exact at any N; used for small
// clouds and as the on-device cross-check of the grid sweep.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_knn_brute(KnnArgs a) {
    __shared__ double s_pend_d[kWavesPerBlock][64 * R + 64];
    __shared__ int s_pend_p[kWavesPerBlock][64 * R + 64];
    const int w = threadIdx.x >> 6;
    const int64_t q = (int64_t)a.q_begin + (int64_t)blockIdx.x * kWavesPerBlock + w;
    if (q >= a.q_end) return;

    Sweep<R> sw;
    sw.k = a.k;
    sw.eps2 = a.eps2;
    sw.pts = a.pts;
    sw.pend_d = s_pend_d[w];
    sw.pend_p = s_pend_p[w];
    if (a.ptsd) {
        const double4 qd = a.ptsd[q];
        sw.qx = qd.x; sw.qy = qd.y; sw.qz = qd.z;
    } else {
        const float4 qp = a.pts[q];
        sw.qx = (double)qp.x; sw.qy = (double)qp.y; sw.qz = (double)qp.z;
    }
    sw.reset();
    sw.scan_global(0, (int)a.n);
    sw.finish();
    sw.store(q, a.nbr_pos, a.nbr_dist, a.nbr_cnt);
}

// sorted-space table -> public (rows,k) arrays for public rows [begin,end)
__global__ __launch_bounds__(256) void k_export(const float4* __restrict__ pts, const int* __restrict__ nbr_pos,
                                                const float* __restrict__ nbr_dist, const int* __restrict__ nbr_cnt,
                                                int64_t n, int k, int64_t begin, int64_t end,
                                                int* __restrict__ idx_out, float* __restrict__ dist_out,
                                                int* __restrict__ cnt_out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / k;
    const int j = (int)(t - row * k);
    if (row >= n) return;
    const int pub = __float_as_int(pts[row].w);
    if (pub < begin || pub >= end) return;
    const int64_t o = (int64_t)(pub - begin) * k + j;
    const int pos = nbr_pos[row * k + j];
    if (idx_out) idx_out[o] = pos < 0 ? (int)n : __float_as_int(pts[pos].w);
    if (dist_out) dist_out[o] = nbr_dist[row * k + j];
    if (cnt_out && j == 0) cnt_out[pub - begin] = nbr_cnt ? nbr_cnt[row] : k;
}

KnnArgs make_args(pct_ctx* ctx, int32_t k, double eps, bool grid) {
    KnnArgs a = {};
    a.pts = (const float4*)(grid ? ctx->sorted4.p : ctx->pts4.p);
    a.ptsd = ctx->has_f64 ? (const double4*)(grid ? ctx->sorted4d.p : ctx->pts4d.p) : nullptr;
    a.cell_start = (const int*)ctx->cell_cnt.p;
    a.occ = (const int*)ctx->occ.p;
    a.n_occ = ctx->n_occ;
    a.n = ctx->n;
    a.g = ctx->grid;
    a.k = k;
    a.eps2 = eps > 0 ? eps * eps : INFINITY;
    a.q_begin = (int)ctx->q_begin;
    a.q_end = (int)ctx->q_end;
    a.nbr_pos = (int*)ctx->nbr_pos.p;
    a.nbr_dist = (float*)ctx->nbr_dist.p;
    a.nbr_cnt = eps > 0 ? (int*)ctx->nbr_cnt.p : nullptr;
    a.counters = (unsigned long long*)ctx->counters.p;
    return a;
}

int reserve_table(pct_ctx* ctx, int32_t k, double eps) {
    PCT_TRY(pct_reserve(ctx, &ctx->nbr_pos, (size_t)ctx->n * k * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->nbr_dist, (size_t)ctx->n * k * sizeof(float)));
    if (eps > 0) PCT_TRY(pct_reserve(ctx, &ctx->nbr_cnt, (size_t)ctx->n * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->counters, 64));
    PCT_HIP(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
    return PCT_OK;
}

}  // namespace

int pct_launch_knn_grid(pct_ctx* ctx, int32_t k, double eps) {
    PCT_TRY(reserve_table(ctx, k, eps));
    KnnArgs a = make_args(ctx, k, eps, true);
    const int blocks = (int)((ctx->n_occ + kWavesPerBlock - 1) / kWavesPerBlock);
    if (blocks > 0) {
        if (k + 1 <= 64)
            hipLaunchKernelGGL(k_knn_grid<1>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(k_knn_grid<2>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a);
        PCT_HIP(ctx, hipGetLastError());
    }
    ctx->knn_sorted_space = true;
    return PCT_OK;
}

int pct_launch_knn_brute(pct_ctx* ctx, int32_t k, double eps) {
    PCT_TRY(reserve_table(ctx, k, eps));
    KnnArgs a = make_args(ctx, k, eps, false);
    const int64_t nq = ctx->q_end - ctx->q_begin;
    const int blocks = (int)((nq + kWavesPerBlock - 1) / kWavesPerBlock);
    if (blocks > 0) {
        if (k + 1 <= 64)
            hipLaunchKernelGGL(k_knn_brute<1>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(k_knn_brute<2>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, ctx->stream, a);
        PCT_HIP(ctx, hipGetLastError());
    }
    ctx->knn_sorted_space = false;
    return PCT_OK;
}

int pct_launch_export_neighbors(pct_ctx* ctx, int64_t begin, int64_t end, int32_t* d_idx, float* d_dist,
                                int32_t* d_cnt) {
    const float4* pts = (const float4*)(ctx->knn_sorted_space ? ctx->sorted4.p : ctx->pts4.p);
    const int64_t total = ctx->n * ctx->k;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(k_export, dim3(blocks), dim3(256), 0, ctx->stream, pts, (const int*)ctx->nbr_pos.p,
                       (const float*)ctx->nbr_dist.p, ctx->eps > 0 ? (const int*)ctx->nbr_cnt.p : nullptr, ctx->n,
                       ctx->k, begin, end, d_idx, d_dist, d_cnt);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
