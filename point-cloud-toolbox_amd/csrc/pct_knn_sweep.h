// Shared by pct_knn.hip and pct_knn_wide.hip: the wave-per-query sweeps (exact sweep over the cell list, exhaustive
// sweep) and what they are built from -- the running list of 64 R (fp64 d2, sorted position) elements, its bitonic
// network, the shell iterator over the cell list, the radius a searched cube vouches for.  R = 1, 2 are instantiated
// in pct_knn.hip; R = 4, 8 (rows of up to 511 neighbours: cKDTree.query takes any k, pct:83) in pct_knn_wide.hip, a
// translation unit of its own so that the hot kernels' file does not pay their compile time.
#pragma once
#include "pct_internal.h"

#include <math.h>
#include <type_traits>

struct KnnArgs {
    const float4* pts;        // candidate records {x,y,z,public index}; cell-sorted (grid) or public order (brute)
    const double4* ptsd;      // native fp64 query coordinates in the same order (nullable)
    const int* cell_start;    // grid only
    const int* cell_own;      // grid only: owned points per cell (they come first inside the cell)
    const int* own_start;     // grid only: first table row of every cell
    const int* owned_pos;     // grid only: sorted position of every table row
    int64_t n_owned;          // rows of the neighbour table
    const int* occ;           // grid only
    int64_t n_occ;
    int64_t n;
    pct_grid g;
    int k;
    int pitch;                // row pitch of the neighbour table in elements: k rounded up to a multiple of 4
    double eps2;              // +inf when no bound
    int q_begin, q_end;       // owned public index range
    int* nbr_pos;
    float* nbr_dist;
    int* nbr_cnt;             // nullable
    int* row_done;            // nullable: set to 1 for every row a kernel has answered (level passes)
    int* redo_m;              // nullable (level passes), one slot per table row: why << 29 | candidates the stencil of the row's item held
    // tree sweep (pct_tree.hip): the cloud in Morton order; a work item is a run of queries of one cell of the
    // octree level that suits them, its stencil the 27 cells of that level -- 27 contiguous ranges of the cloud
    const int4* tree_seg;     // per segment {level, cx, cy, cz}
    const int2* tree_runs;    // per segment 27 x {first position, points}, centre cell first
    int tree_bits;            // levels below the root (cells per axis at level l: 2^(tree_bits - l)); g = the finest level's grid
    const unsigned long long* tree_codes;   // Morton code of every position (exact sweep on the tree)
    const unsigned char* tree_lvl;          // level every position is served at
    const int* tree_bucket;                 // first position of every 18-bit code prefix
    int stats;                // collect the counters below (off by default)
    unsigned long long* counters;   // [0] ring fallbacks [1] LDS overflows [2] flushes [3] candidate steps [4] redone queries [5] queries beyond the culling limits (always counted)
};

// rows of 128 .. PCT_K_MAX neighbours (pct_knn_wide.hip): list registers R = 4 | 8
int pct_launch_knn_exact_wide(pct_ctx* ctx, const KnnArgs& a, int blocks, const int* list, const int* list_count);
int pct_launch_knn_brute_wide(pct_ctx* ctx, const KnnArgs& a, int blocks);

namespace {

constexpr int kWavesPerBlock = 4;
#ifndef PCT_STAGE_CAP
#define PCT_STAGE_CAP 512
#endif
constexpr int kStageCap = PCT_STAGE_CAP;   // LDS-staged stencil candidates per wave and per 64 list slots (12 B each, SoA)
#ifndef PCT_STAGE_CAP2
#define PCT_STAGE_CAP2 768
#endif
constexpr int PCT_STAGE_CAP2_HOST = PCT_STAGE_CAP2;
#ifndef PCT_TREE_CAP
#define PCT_TREE_CAP 768                   // staged candidates of a work item of the hierarchical cell list (A/B: 512 | 768 | 1024)
#endif
#ifndef PCT_TREE_CAP2
#define PCT_TREE_CAP2 1024                 // ... for k + 1 > 64 (two list registers): the proofs want ~2.6 (k+1) stencil points
#endif


__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// LDS hand-off between lanes of ONE wave: no hardware barrier is needed (the
// wave issues its DS operations in order) but the compiler must not move LDS
// accesses across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of lane (lane ^ S) -- S is a compile-time power of two
template <int S>
__device__ __forceinline__ int lane_xor(int v) {
    if constexpr (S == 1) {
        return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);             // quad_perm [1,0,3,2]
    } else if constexpr (S == 2) {
        return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);             // quad_perm [2,3,0,1]
    } else if constexpr (S == 4) {
        int t = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xA, false);    // row_ror:4  -> banks 1,3 (lane bit2 set)
        return __builtin_amdgcn_update_dpp(t, v, 0x12C, 0xF, 0x5, false);     // row_ror:12 -> banks 0,2
    } else if constexpr (S == 8) {
        return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);            // row_ror:8
    } else if constexpr (S == 16) {
        // v_permlane16_swap: odd rows of the first operand <-> even rows of the second (VALU, no LDS round trip)
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((threadIdx.x & 16) ? r[0] : r[1]);
    } else {
        // v_permlane32_swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)((threadIdx.x & 32) ? r[0] : r[1]);
    }
}

__device__ __forceinline__ int pub_index(const float4* pts, int pos) { return __float_as_int(pts[pos].w); }

// strict total order on (d2, public index); padding = (+inf, INT_MAX).
// The public-index lookup only runs when some lane sees an exact fp64 tie.
__device__ __forceinline__ bool key_less(double da, int pa, double db, int pb, const float4* pts) {
    const bool lt = da < db;
    bool tie = (da == db) && (pa != pb) && (pa != INT_MAX) && (pb != INT_MAX);
    bool tb = false;
    if (__builtin_expect(__ballot(tie) != 0ull, 0)) {
        if (tie) tb = pub_index(pts, pa) < pub_index(pts, pb);
    }
    // finite value vs padding with equal d2 cannot happen (padding is +inf); inf-vs-inf real elements do not exist
    return lt || (tie && tb);
}

template <int R>
struct TopK {
    double d[R];
    int p[R];
};

// One compare-exchange level over 64*R elements, element index i = lane + 64*slot.
template <int R, int STRIDE>
__device__ __forceinline__ void bitonic_level(TopK<R>& t, int size, bool descending, const float4* pts) {
    const int lane = lane_id();
    if constexpr (STRIDE >= 64) {
        constexpr int ds = STRIDE >> 6;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r & ds) == 0 && (r | ds) < R) {
                const int r2 = r | ds;
                const int i = lane + 64 * r;
                const bool asc = ((i & size) == 0) != descending;
                const bool hi_less = key_less(t.d[r2], t.p[r2], t.d[r], t.p[r], pts);
                if (hi_less == asc) {
                    double td = t.d[r]; t.d[r] = t.d[r2]; t.d[r2] = td;
                    int tp = t.p[r]; t.p[r] = t.p[r2]; t.p[r2] = tp;
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int lo = lane_xor<STRIDE>(__double2loint(t.d[r]));
            const int hi = lane_xor<STRIDE>(__double2hiint(t.d[r]));
            const int pp = lane_xor<STRIDE>(t.p[r]);
            const double pd = __hiloint2double(hi, lo);
            const int i = lane + 64 * r;
            const bool asc = ((i & size) == 0) != descending;
            const bool keep_min = ((lane & STRIDE) == 0) == asc;
            const bool partner_less = key_less(pd, pp, t.d[r], t.p[r], pts);
            if (keep_min == partner_less) {
                t.d[r] = pd;
                t.p[r] = pp;
            }
        }
    }
}

template <int R, int STRIDE>
__device__ __forceinline__ void bitonic_strides(TopK<R>& t, int size, bool descending, const float4* pts) {
    bitonic_level<R, STRIDE>(t, size, descending, pts);
    if constexpr (STRIDE > 1) bitonic_strides<R, STRIDE / 2>(t, size, descending, pts);
}

template <int R, int SIZE>
__device__ __forceinline__ void bitonic_sort_from(TopK<R>& t, bool descending, const float4* pts) {
    bitonic_strides<R, SIZE / 2>(t, SIZE, descending, pts);
    if constexpr (SIZE < 64 * R) bitonic_sort_from<R, SIZE * 2>(t, descending, pts);
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
    bitonic_strides<R, 32 * R>(best, 64 * R, false, pts);
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
        // first batch: ascending and adopted as-is; later batches: descending, then merged
        bitonic_sort_from<R, 2>(b, !empty, pts);
        if (empty) {
            best = b;
            empty = false;
        } else {
            bitonic_merge_min<R>(best, b, pts);
        }
        refresh_tau();
    }

    // one candidate per lane; survivors are compacted into the pending buffer
    __device__ __forceinline__ void consider(float4 c, int pos, bool valid) {
        const double dx = (double)c.x - qx, dy = (double)c.y - qy, dz = (double)c.z - qz;
        const double d2 = (dx * dx + dy * dy) + dz * dz;
        // Every lane takes part (wave ballot inside) -- but a lane without a candidate must not take part in the
        // tie-break: its registers hold the coordinates of an EARLIER batch (or zeros), whose d2 can equal the running
        // (k+1)-th distance exactly (that earlier candidate may BE the (k+1)-th), and the tie-break reads the public
        // index at `pos` -- up to 63 records past the end of the cloud for the last batch of the last run.  On clouds
        // below ~380 points that is beyond the head-room of the allocation: the GPU memory fault that aborted a test
        // run once in a while for two rounds.  The padding position never enters a tie-break.
        const bool closer = key_less(d2, valid ? pos : INT_MAX, tau_d, tau_p, pts);
        const bool pass = valid && d2 < eps2 && closer;
        const unsigned long long m = __ballot(pass);
        if (pass) {
            const int slot = npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            pend_d[slot] = d2;
            pend_p[slot] = pos;
        }
        npend += __popcll(m);
    }

    // rows: element i (1..k) -> output column i-1
    __device__ __forceinline__ void store(int64_t row, int pitch, int* nbr_pos, float* nbr_dist, int* nbr_cnt) {
        const int lane = lane_id();
        int found = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane + 64 * r;
            const bool real = best.p[r] != INT_MAX;
            if (i >= 1 && i <= k) {
                nbr_pos[row * pitch + (i - 1)] = real ? best.p[r] : -1;
                if (nbr_dist) nbr_dist[row * pitch + (i - 1)] = real ? (float)sqrt(best.d[r]) : INFINITY;
                found += real;
            }
        }
        if (nbr_cnt) {
            for (int o = 32; o > 0; o >>= 1) found += __shfl_xor(found, o);
            if (lane == 0) nbr_cnt[row] = found;
        }
    }
};

// (dz, dy) of the nine stencil rows, nearest first: the first batches then hold
// the closest points and the (k+1)-th distance tightens early.
__constant__ signed char kRowOrder[9][2] = {{0, 0}, {0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};

// Iterator over the x-runs of the cells at Chebyshev distance d from cell (cx,cy,cz) with r_lo < d <= r_hi
// (r_lo = -1: the whole cube of radius r_hi), clipped to the grid.  Only the (dz,dy) rows that exist in the grid
// are enumerated -- a query clamped into a corner of a long thin grid would otherwise walk (2r+1)^2 row slots per
// ring, almost all of them outside.  The run bounds of up to 64 rows are fetched by the 64 lanes in parallel (one
// memory round trip per 64 rows instead of two dependent scalar loads per row), rows without points are skipped
// with a ballot, and the runs are handed out one 64-candidate step at a time, wave-uniformly.
struct ShellIter {
    int r_lo, ring, zlo, ylo, wy, nrows, row, part, pos, end, chunk;
    unsigned long long live;        // rows of the current chunk that hold at least one point
    int b_s0, b_e0, b_s1, b_e1;     // per lane: bounds of the (up to two) runs of row chunk + lane
    // Pruning: once the list is full, a cell whose box lies farther from the query than the current (k+1)-th distance
    // cannot change it (a candidate replaces an entry only if it is nearer, or as near with a smaller index: a point
    // strictly farther never does).  prune = that squared distance in cell units, shrunk bounds (1 - 1e-6, as in
    // guaranteed_r2) on the cell side; +inf = keep everything.  Rows wholly beyond it are dropped, the others keep the
    // cells of their x-range that can reach.  Clamped points (outside the grid box, filed in a boundary cell) lie
    // beyond that cell's outer face: never nearer than the box says.
    // (slack: cell coordinates are fl(fl(x - o) inv) -- two roundings of a number of up to n cells, or of the distance of
    // a clamped query from the box: an absolute error of that many ulps, taken off every gap)
    double prune, gx, gy, gz, slack;
    __device__ __forceinline__ void start(const pct_grid& g, int cy, int cz, int lo, int hi, double prune_d2 = INFINITY,
                                          double qgx = 0.0, double qgy = 0.0, double qgz = 0.0) {
        const double ce = g.cell * (1.0 - 1e-6);
        prune = prune_d2 / (ce * ce);
        gx = qgx; gy = qgy; gz = qgz;
        slack = 1e-15 * ((double)g.nx + (double)g.ny + (double)g.nz + fabs(qgx) + fabs(qgy) + fabs(qgz));
        r_lo = lo; ring = hi;
        zlo = max(-hi, -cz);
        ylo = max(-hi, -cy);
        const int zhi = min(hi, g.nz - 1 - cz), yhi = min(hi, g.ny - 1 - cy);
        wy = yhi - ylo + 1;
        nrows = (zhi - zlo + 1) * wy;
        row = 0; part = 0; pos = 0; end = 0; chunk = -1; live = 0ull;
    }
    __device__ __forceinline__ void fetch(const pct_grid& g, const int* __restrict__ cs, int cx, int cy, int cz) {
        const int ri = chunk + lane_id();
        b_s0 = b_e0 = b_s1 = b_e1 = 0;
        if (ri < nrows) {
            const int dz = zlo + ri / wy, dy = ylo + ri % wy;
            const int base = ((cz + dz) * g.ny + (cy + dy)) * g.nx;
            const bool inner = max(abs(dz), abs(dy)) <= r_lo;      // the row crosses the cube already searched
            // x-range of the row the pruning bound leaves: offsets [x_lo, x_hi] around cx (empty: x_lo > x_hi)
            int x_lo = -ring, x_hi = ring;
            if (prune < INFINITY) {
                const double ty = dy > 0 ? fmax(dy - gy - slack, 0.0) : dy < 0 ? fmax(gy - (dy + 1) - slack, 0.0) : 0.0;
                const double tz = dz > 0 ? fmax(dz - gz - slack, 0.0) : dz < 0 ? fmax(gz - (dz + 1) - slack, 0.0) : 0.0;
                const double left = prune - (ty * ty + tz * tz);
                if (left < 0.0) { x_lo = 1; x_hi = 0; }
                else {
                    const double rx = sqrt(left) * (1.0 + 1e-9) + 1e-9 + slack;
                    // offset dx > 0 reaches if dx - gx <= rx; dx < 0 if gx - (dx + 1) <= rx
                    x_hi = (int)fmin((double)ring, floor(rx + gx));
                    x_lo = -(int)fmin((double)ring, floor(rx + 1.0 - gx));
                    x_hi = max(x_hi, 0);                         // (the query's own column always stays)
                    x_lo = min(x_lo, 0);
                }
            }
            if (x_lo <= x_hi) {
                if (!inner) {
                    b_s0 = cs[base + max(cx + x_lo, 0)];
                    b_e0 = cs[base + min(cx + x_hi, g.nx - 1) + 1];
                } else {
                    const int a0 = max(cx + x_lo, 0), a1 = cx - r_lo - 1;        // left of the searched cube
                    const int c0 = cx + r_lo + 1, c1 = min(cx + x_hi, g.nx - 1); // right of it
                    if (a1 >= a0) { b_s0 = cs[base + a0]; b_e0 = cs[base + a1 + 1]; }
                    if (c1 >= c0) { b_s1 = cs[base + c0]; b_e1 = cs[base + c1 + 1]; }
                }
            }
        }
        live = __builtin_amdgcn_ballot_w64(b_e0 > b_s0 || b_e1 > b_s1);
    }
    // next 64-candidate step: returns false when the shell is exhausted
    __device__ __forceinline__ bool next(const pct_grid& g, const int* __restrict__ cs, int cx, int cy, int cz, int& base, int& lim) {
        while (pos >= end) {
            if (row >= nrows) return false;
            if (chunk < 0 || row - chunk >= 64) { chunk = row; fetch(g, cs, cx, cy, cz); }
            int l = row - chunk;
            if (part == 0) {
                const unsigned long long rest = live >> l;      // rows of this chunk from l on that hold points
                if (rest == 0ull) { row = chunk + 64; continue; }
                const int skip = (int)__builtin_ctzll(rest);
                row += skip;
                l += skip;
                pos = __builtin_amdgcn_readlane(b_s0, l);
                end = __builtin_amdgcn_readlane(b_e0, l);
                part = 1;
            } else {
                pos = __builtin_amdgcn_readlane(b_s1, l);
                end = __builtin_amdgcn_readlane(b_e1, l);
                part = 0;
                ++row;
            }
        }
        base = pos;
        lim = end;
        pos += 64;
        return true;
    }
};

// squared radius (cell units folded in) inside which the cube of radius `ring`
// around the query's cell is known to contain every point; +inf once the cube
// covers the grid.  gx,gy,gz = position of the query inside its cell in cell units.
__device__ __forceinline__ double guaranteed_r2(const pct_grid& g, int cx, int cy, int cz, double gx, double gy, double gz, int ring) {
    const double inf = INFINITY;
    double gmin = inf;
    gmin = fmin(gmin, cx - ring <= 0 ? inf : gx + ring);
    gmin = fmin(gmin, cx + ring >= g.nx - 1 ? inf : (1.0 - gx) + ring);
    gmin = fmin(gmin, cy - ring <= 0 ? inf : gy + ring);
    gmin = fmin(gmin, cy + ring >= g.ny - 1 ? inf : (1.0 - gy) + ring);
    gmin = fmin(gmin, cz - ring <= 0 ? inf : gz + ring);
    gmin = fmin(gmin, cz + ring >= g.nz - 1 ? inf : (1.0 - gz) + ring);
    const double rr = gmin * g.cell * (1.0 - 1e-6);
    return rr * rr;
}

// Squared distance from the query to the nearest face beyond which points were left out of the grid
// (pct_grid::lim_*): nothing the kept points say is proven past it.  +inf for a grid that holds the whole cloud.
__device__ __forceinline__ double limit_r2(const pct_grid& g, int cx, int cy, int cz, double gx, double gy, double gz) {
    const double px = cx + gx, py = cy + gy, pz = cz + gz;
    double t = fmin(px - g.lim_lo[0], g.lim_hi[0] - px);
    t = fmin(t, fmin(py - g.lim_lo[1], g.lim_hi[1] - py));
    t = fmin(t, fmin(pz - g.lim_lo[2], g.lim_hi[2] - pz));
    const double rr = fmax(t, 0.0) * g.cell * (1.0 - 1e-6);
    return rr * rr;
}

__device__ __forceinline__ int cell_coord_d(double x, double o, double inv, int n) {
    int c = (int)floor((x - o) * inv);
    return min(max(c, 0), n - 1);
}

// ---------------------------------------------------------------------------
// Exact sweep, wave = query, candidates from global memory cube by cube.
// Runs (a) the queries the fast kernel flagged as ambiguous (list != null) and
// (b) every owned query when the exact path is requested for testing.
// Comparisons are on (fp64 d2, public index), so ties are resolved exactly.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_knn_exact(KnnArgs a, const int* __restrict__ list,
                                                                   const int* __restrict__ list_count) {
    __shared__ double s_pend_d[kWavesPerBlock][64 * R + 64];
    __shared__ int s_pend_p[kWavesPerBlock][64 * R + 64];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const pct_grid g = a.g;
    const int* __restrict__ cs = a.cell_start;
    const int64_t total = list ? (int64_t)*list_count : a.n_owned;
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

    Sweep<R> sw;
    sw.k = a.k;
    sw.eps2 = a.eps2;
    sw.pts = a.pts;
    sw.pend_d = s_pend_d[w];
    sw.pend_p = s_pend_p[w];

    for (int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + w; item < total; item += nwaves) {
        const int row = list ? list[item] : (int)item;          // neighbour-table row (owned queries only)
        const int q = a.owned_pos[row];
        const float4 qp = a.pts[q];
        // the query's cell comes from its float32 (tree) coordinates, as in the build
        const int cx = __builtin_amdgcn_readfirstlane(cell_coord_d((double)qp.x, g.ox, g.inv_cell, g.nx));
        const int cy = __builtin_amdgcn_readfirstlane(cell_coord_d((double)qp.y, g.oy, g.inv_cell, g.ny));
        const int cz = __builtin_amdgcn_readfirstlane(cell_coord_d((double)qp.z, g.oz, g.inv_cell, g.nz));
        if (a.ptsd) {
            const double4 qd = a.ptsd[q];
            sw.qx = qd.x; sw.qy = qd.y; sw.qz = qd.z;
        } else {
            sw.qx = (double)qp.x; sw.qy = (double)qp.y; sw.qz = (double)qp.z;
        }
        sw.reset();
        const double gx = (sw.qx - g.ox) * g.inv_cell - cx;
        const double gy = (sw.qy - g.oy) * g.inv_cell - cy;
        const double gz = (sw.qz - g.oz) * g.inv_cell - cz;
        ShellIter it;
        it.start(g, cy, cz, -1, 1);
        // candidate loads run one step ahead of their use (a shell is many short runs, each a dependent load)
        int nbase = 0, nlim = 0;
        bool have_next = it.next(g, cs, cx, cy, cz, nbase, nlim);
        float4 c_next = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have_next && nbase + lane < nlim) c_next = a.pts[nbase + lane];
        for (;;) {
            const bool have = have_next;
            if (have) {
                const int pos = nbase + lane;
                const bool valid = pos < nlim;
                const float4 c = c_next;
                have_next = it.next(g, cs, cx, cy, cz, nbase, nlim);
                if (have_next && nbase + lane < nlim) c_next = a.pts[nbase + lane];
                sw.consider(c, pos, valid);
                if (sw.npend < 64 * R) continue;
            }
            if (sw.npend > 0 || sw.empty) sw.flush();
            if (have) continue;
            if (fmin(sw.tau_d, sw.eps2) <= guaranteed_r2(g, cx, cy, cz, gx, gy, gz, it.ring)) {
                // final among the kept points; a handle that left points out must also be inside its limits
                if (lane == 0 && fmin(sw.tau_d, sw.eps2) > limit_r2(g, cx, cy, cz, gx, gy, gz)) atomicAdd(&a.counters[5], 1ull);
                break;
            }
            // widen: one ring at a time near the query, then by half the radius (a query clamped into a corner of a
            // large empty grid must not pay one round per ring; the guarantee is that of the outer radius)
            it.start(g, cy, cz, it.ring, it.ring < 4 ? it.ring + 1 : it.ring + (it.ring + 1) / 2, fmin(sw.tau_d, sw.eps2), gx, gy, gz);
            have_next = it.next(g, cs, cx, cy, cz, nbase, nlim);
            if (have_next && nbase + lane < nlim) c_next = a.pts[nbase + lane];
        }
        if (a.stats && lane == 0 && it.ring > 1) atomicAdd(&a.counters[0], 1ull);
        sw.store(row, a.pitch, a.nbr_pos, a.nbr_dist, a.nbr_cnt);
        if (a.row_done && lane == 0) a.row_done[row] = 1;
    }
}

// Exact sweep on the hierarchical cell list (pct_tree.hip), wave = query of the redo list: the 27 cells of the query's
// octree level are 27 ranges of the Morton-ordered cloud (found by binary search, one cell per lane); a result the
// cube cannot vouch for is started over one level up, where the cube is twice as wide -- the root vouches for
// everything.  Same (fp64 d2, public index) order and the same running list as k_knn_exact.
template <int R>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_knn_exact_tree(KnnArgs a, const int* __restrict__ list,
                                                                        const int* __restrict__ list_count) {
    __shared__ double s_pend_d[kWavesPerBlock][64 * R + 64];
    __shared__ int s_pend_p[kWavesPerBlock][64 * R + 64];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int64_t total = list ? (int64_t)*list_count : a.n_owned;
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

    Sweep<R> sw;
    sw.k = a.k;
    sw.eps2 = a.eps2;
    sw.pts = a.pts;
    sw.pend_d = s_pend_d[w];
    sw.pend_p = s_pend_p[w];

    for (int64_t item = (int64_t)blockIdx.x * kWavesPerBlock + w; item < total; item += nwaves) {
        const int row = list ? list[item] : (int)item;          // table row = Morton position of the query
        if (a.ptsd) {                               // float64 clouds: native query, float32-rounded candidates (pct:74, 83)
            const double4 qd = a.ptsd[row];
            sw.qx = qd.x; sw.qy = qd.y; sw.qz = qd.z;
        } else {
            const float4 qp = a.pts[row];
            sw.qx = (double)qp.x; sw.qy = (double)qp.y; sw.qz = (double)qp.z;
        }
        const unsigned long long code = a.tree_codes[row];
        const int fx = (int)pct_compact3(code), fy = (int)pct_compact3(code >> 1), fz = (int)pct_compact3(code >> 2);
        int level = __builtin_amdgcn_readfirstlane((int)a.tree_lvl[row]);
        int rounds = 0;
        unsigned long long steps = 0;
        for (;;) {
            pct_grid g = a.g;
            g.cell = __builtin_ldexp(a.g.cell, level);
            g.inv_cell = __builtin_ldexp(a.g.inv_cell, -level);
            g.nx = g.ny = g.nz = 1 << (a.tree_bits - level);
            const int cx = __builtin_amdgcn_readfirstlane(fx >> level), cy = __builtin_amdgcn_readfirstlane(fy >> level),
                      cz = __builtin_amdgcn_readfirstlane(fz >> level);
            int run_s = 0, run_len = 0;
            if (lane < 27) {
                int dx, dy, dz;
                pct_stencil_cell(lane, &dx, &dy, &dz);
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                    const unsigned long long prefix = pct_spread3((unsigned)x) | pct_spread3((unsigned)y) << 1 | pct_spread3((unsigned)z) << 2;
                    const int64_t lo = pct_code_lower_bound(a.tree_codes, a.tree_bucket, prefix << (3 * level));
                    const int64_t hi = pct_code_lower_bound(a.tree_codes, a.tree_bucket, (prefix + 1) << (3 * level));
                    run_s = (int)lo;
                    run_len = (int)(hi - lo);
                }
            }
            sw.reset();
            for (int t = 0; t < 27; ++t) {
                const int s0 = __builtin_amdgcn_readlane(run_s, t), s1 = s0 + __builtin_amdgcn_readlane(run_len, t);
                for (int base = s0; base < s1; base += 64) {
                    const int pos = base + lane;
                    const bool valid = pos < s1;
                    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid) c = a.pts[pos];
                    sw.consider(c, pos, valid);
                    if (sw.npend >= 64 * R) sw.flush();
                    ++steps;
                }
            }
            if (sw.npend > 0 || sw.empty) sw.flush();
            const double gx = (sw.qx - g.ox) * g.inv_cell - cx;
            const double gy = (sw.qy - g.oy) * g.inv_cell - cy;
            const double gz = (sw.qz - g.oz) * g.inv_cell - cz;
            if (fmin(sw.tau_d, sw.eps2) <= guaranteed_r2(g, cx, cy, cz, gx, gy, gz, 1) || level >= a.tree_bits) break;
            ++level;
            ++rounds;
        }
        if (a.stats && lane == 0) {
            if (rounds > 0) atomicAdd(&a.counters[0], 1ull);
            atomicAdd(&a.counters[3], steps);
            atomicMax(&a.counters[6], (steps << 32) | (unsigned)row);      // the costliest query: 64-candidate steps, row
        }
        sw.store(row, a.pitch, a.nbr_pos, a.nbr_dist, a.nbr_cnt);
    }
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
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
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
    const int n = (int)a.n;
    for (int base = 0;; base += 64) {
        const bool have = base < n;
        if (have) {
            const int pos = base + lane;
            const bool valid = pos < n;
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) c = a.pts[pos];
            sw.consider(c, pos, valid);
            if (sw.npend < 64 * R) continue;
        }
        if (sw.npend > 0 || sw.empty) sw.flush();
        if (!have) break;
    }
    sw.store(q - a.q_begin, a.pitch, a.nbr_pos, a.nbr_dist, a.nbr_cnt);
}

}  // namespace
