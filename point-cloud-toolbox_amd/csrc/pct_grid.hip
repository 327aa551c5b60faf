// Uniform cell list build for the neighbour sweep (the search structure that
// stands where the reference builds scipy's cKDTree, pointCloudToolbox.py:74).
//
//   pack      xyz (n,3) f32 -> float4 {x,y,z,index}; finite check; bbox and moments per block, folded by a one-block
//             kernel (sharded handles: k_cull_pack keeps the owned rows and the points near them)
//   hist      cell id per point + per-cell counts of owned / other points (integer atomics)
//   occupancy mean points-per-cell as seen by a point (drives the cell size)
//   scan      exclusive scan of the counts -> cell starts, ordered occupied list
//   scatter   counting sort of the float4 records into cell order, owned points first in each cell
//
// All kernels are HBM/L2 streaming passes over 16 B records with 64-wide waves.  Host side (pct_build_grid): outlier-
// trimmed grid box, cell size by occupancy (warm-started, speculative box from the previous similar cloud), one
// host synchronisation per accepted build; small results come back through pinned memory the kernels write.
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

// Reduction record of the packing passes (device memory, ctx->red).
struct PackRed {
    int bb[6];                 // min xyz, max xyz as ordered ints
    int bad;                   // a non-finite coordinate was seen
    int g_begin;               // compacted position of the first owned point (cull pass)
    double s[3], ss[3];        // sums / sums of squares of (coordinate - shift): drives the outlier-trimmed grid box
    unsigned long long cnt;    // points accumulated
};

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

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Per-thread accumulator of the packing passes; commit() folds a block into its own PackRed record.
struct PackAcc {
    float mn[3], mx[3];
    double s[3], ss[3];
    int bad;
    unsigned cnt;
    __device__ void init() {
        for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; s[a] = 0; ss[a] = 0; }
        bad = 0;
        cnt = 0;
    }
    __device__ void add(float x, float y, float z, const float* sh) {
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        const double dx = (double)x - sh[0], dy = (double)y - sh[1], dz = (double)z - sh[2];
        s[0] += dx; ss[0] += dx * dx;
        s[1] += dy; ss[1] += dy * dy;
        s[2] += dz; ss[2] += dz * dz;
        ++cnt;
    }
    __device__ void add_box(float x, float y, float z) {     // extent only (no moments)
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        ++cnt;
    }
    template <int NW = kBlock / 64>
    __device__ void commit(PackRed* part) {      // this block's record (plain stores); NW = waves per block
        __shared__ float s_mn[NW][3], s_mx[NW][3];
        __shared__ double s_s[NW][3], s_ss[NW][3];
        __shared__ int s_bad[NW];
        __shared__ unsigned s_cnt[NW];
        for (int a = 0; a < 3; ++a) {
            mn[a] = wave_min(mn[a]);
            mx[a] = wave_max(mx[a]);
            s[a] = wave_sum(s[a]);
            ss[a] = wave_sum(ss[a]);
        }
        bad = __any(bad);
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            for (int a = 0; a < 3; ++a) { s_mn[w][a] = mn[a]; s_mx[w][a] = mx[a]; s_s[w][a] = s[a]; s_ss[w][a] = ss[a]; }
            s_bad[w] = bad;
            s_cnt[w] = cnt;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int a = threadIdx.x;
            float lo = s_mn[0][a], hi = s_mx[0][a];
            double t = s_s[0][a], tt = s_ss[0][a];
            for (int i = 1; i < NW; ++i) {
                lo = fminf(lo, s_mn[i][a]); hi = fmaxf(hi, s_mx[i][a]);
                t += s_s[i][a]; tt += s_ss[i][a];
            }
            part->bb[a] = float_order(lo);
            part->bb[3 + a] = float_order(hi);
            part->s[a] = t;
            part->ss[a] = tt;
        }
        if (threadIdx.x == 3) {
            int b = 0;
            unsigned c = 0;
            for (int i = 0; i < NW; ++i) { b |= s_bad[i]; c += s_cnt[i]; }
            part->bad = b;
            part->cnt = c;
        }
    }
};

// second stage of every packing pass: folds the per-block records into the handle's PackRed.  (Atomics from
// every block onto one cache line cost more than the pass itself.)
// carry = 0: nothing of the pass wrote the record directly (k_pack), so its previous content is not read and the
// 128-byte memset in front of the pass -- a launch of its own -- is not needed.
__global__ __launch_bounds__(kBlock) void k_pack_final(const PackRed* __restrict__ parts, int n_parts, PackRed* __restrict__ red,
                                                       PackRed* __restrict__ host_copy, int carry) {
    __shared__ PackRed sh[kBlock / 64];
    int bb[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
    double s[3] = {0, 0, 0}, ss[3] = {0, 0, 0};
    int bad = 0;
    unsigned long long cnt = 0;
    for (int i = threadIdx.x; i < n_parts; i += kBlock) {
        const PackRed p = parts[i];
        for (int a = 0; a < 3; ++a) {
            bb[a] = min(bb[a], p.bb[a]);
            bb[3 + a] = max(bb[3 + a], p.bb[3 + a]);
            s[a] += p.s[a];
            ss[a] += p.ss[a];
        }
        bad |= p.bad;
        cnt += p.cnt;
    }
    for (int o = 32; o > 0; o >>= 1) {
        for (int a = 0; a < 3; ++a) {
            bb[a] = min(bb[a], __shfl_xor(bb[a], o));
            bb[3 + a] = max(bb[3 + a], __shfl_xor(bb[3 + a], o));
            s[a] += __shfl_xor(s[a], o);
            ss[a] += __shfl_xor(ss[a], o);
        }
        bad |= __shfl_xor(bad, o);
        cnt += __shfl_xor(cnt, o);
    }
    if ((threadIdx.x & 63) == 0) {
        PackRed& w = sh[threadIdx.x >> 6];
        for (int a = 0; a < 6; ++a) w.bb[a] = bb[a];
        for (int a = 0; a < 3; ++a) { w.s[a] = s[a]; w.ss[a] = ss[a]; }
        w.bad = bad;
        w.cnt = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PackRed out = {};            // red was zeroed before the pass; bad / g_begin may have been set directly
        for (int a = 0; a < 3; ++a) { out.bb[a] = INT32_MAX; out.bb[3 + a] = INT32_MIN; }
        out.bad = carry ? red->bad : 0;
        out.g_begin = carry ? red->g_begin : 0;
        for (int i = 0; i < kBlock / 64; ++i) {
            for (int a = 0; a < 3; ++a) {
                out.bb[a] = min(out.bb[a], sh[i].bb[a]);
                out.bb[3 + a] = max(out.bb[3 + a], sh[i].bb[3 + a]);
                out.s[a] += sh[i].s[a];
                out.ss[a] += sh[i].ss[a];
            }
            out.bad |= sh[i].bad;
            out.cnt += sh[i].cnt;
        }
        *red = out;
        *host_copy = out;
    }
}

// xyz (n,3) -> float4 {x,y,z,index}; finite check; bbox and moments (fp64, about the origin: the reference
// shifts every cloud by its per-axis maximum, pct:56-57, so coordinates are of the order of the extent).
// four points = twelve consecutive floats = three 16-byte loads per lane, neighbouring lanes adjacent (the scalar
// x / y / z loads of a 12-byte record reach a third of every line each).  VEC needs a 16-byte aligned array.
template <bool VEC>
__device__ __forceinline__ int load_four_points(const float* __restrict__ xyz, int64_t g, int64_t n, float (&v)[12]) {
    const int64_t i0 = g * 4;
    const int have = (int)min((int64_t)4, n - i0);
    if (VEC && have == 4) {
        const float4* q = (const float4*)(xyz + 3 * i0);
        const float4 a = q[0], b = q[1], c = q[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    } else {
        for (int j = 0; j < 12; ++j) v[j] = j < 3 * have ? xyz[3 * i0 + j] : 0.f;
    }
    return have;
}

// (one point per lane and trip: the float4 stores of a wave are 1 KB contiguous; the four-points-per-lane form of the
// cull / slab passes reads better but scatters these stores over 64 lines per instruction -- 16.7 against 12.7 us)
__global__ __launch_bounds__(kBlock) void k_pack(const float* __restrict__ xyz, int64_t n,
                                                 float4* __restrict__ pts4, PackRed* __restrict__ parts) {
    const float sh[3] = {0.f, 0.f, 0.f};
    PackAcc acc;
    acc.init();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const bool ok = isfinite(x) && isfinite(y) && isfinite(z);
        acc.bad |= !ok;
        pts4[i] = make_float4(x, y, z, __int_as_float((int)i));
        if (ok) acc.add(x, y, z, sh);
    }
    acc.commit(parts + blockIdx.x);
}

// bbox / finite check of the rows [begin, end) only (the owned range of a sharded handle)
__global__ __launch_bounds__(kBlock) void k_range_box(const float* __restrict__ xyz, int64_t begin, int64_t end,
                                                      PackRed* __restrict__ parts) {
    const float sh[3] = {0.f, 0.f, 0.f};
    PackAcc acc;
    acc.init();
    for (int64_t i = begin + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < end; i += (int64_t)gridDim.x * kBlock) {
        float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const bool ok = isfinite(x) && isfinite(y) && isfinite(z);
        acc.bad |= !ok;
        if (ok) acc.add(x, y, z, sh);
    }
    acc.commit(parts + blockIdx.x);
}

// moments of the packed points inside a box (outlier trimming of the grid box)
struct Box3 { float lo[3], hi[3]; };
__device__ __forceinline__ bool in_box(const Box3& b, float x, float y, float z) {
    return x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1] && z >= b.lo[2] && z <= b.hi[2];
}
__global__ __launch_bounds__(kBlock) void k_box_stats(const float4* __restrict__ pts4, int64_t n, Box3 box, Box3 shift,
                                                      PackRed* __restrict__ parts) {
    PackAcc acc;
    acc.init();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float4 p = pts4[i];
        if (in_box(box, p.x, p.y, p.z)) acc.add(p.x, p.y, p.z, shift.lo);
    }
    acc.commit(parts + blockIdx.x);
}

// ---- sharded handles: keep only the points near the owned range ------------------------------------------------
// The candidate set of a handle that owns rows [q_begin, q_end) is the whole cloud, but only the points inside
// the owned rows' bounding box plus a margin can be neighbours (the sweep verifies that per query, see
// pct_grid::lim_lo).  One pass: the owned rows go to the front of the packed array in their own order, the kept
// others behind them in whatever order the blocks arrive (one counter increment per 4096-row chunk).
constexpr int kCullBlock = 512;          // (fewer chunks = fewer ticket atomics on one address, the pass's bound)
constexpr int kCullChunk = kCullBlock * 16;

// Ownership by slab (pct_set_query_slab): bin of a coordinate along the cut axis.  Every rank evaluates this very
// expression on the same bytes (the library is built with -ffp-contract=off), in the histogram and in the pack alike.
constexpr int kSlabBins = 4096;
struct SlabCut { int axis, bin_lo, bin_hi; float x0, inv; };
__device__ __forceinline__ int slab_bin(float x, float x0, float inv) {
    const int b = (int)((x - x0) * inv);
    return min(max(b, 0), kSlabBins - 1);
}

// bounding box / finite check of the whole cloud, for the slab cut (count and moments are not needed)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void k_cloud_box(const float* __restrict__ xyz, int64_t n, PackRed* __restrict__ parts) {
    PackAcc acc;
    acc.init();
    const int64_t groups = (n + 3) / 4;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
        float v[12];
        const int have = load_four_points<VEC>(xyz, g, n, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < have) {
                const bool ok = isfinite(v[3 * j]) && isfinite(v[3 * j + 1]) && isfinite(v[3 * j + 2]);
                acc.bad |= !ok;
                if (ok) acc.add_box(v[3 * j], v[3 * j + 1], v[3 * j + 2]);
            }
        }
    }
    acc.commit(parts + blockIdx.x);
}

// (few, large blocks: every block ends with one global atomic per non-empty bin)
constexpr int kSlabHistBlock = 1024;
template <bool VEC>
__global__ __launch_bounds__(kSlabHistBlock) void k_slab_hist(const float* __restrict__ xyz, int64_t n, int axis, float x0, float inv,
                                                              unsigned* __restrict__ hist) {
    constexpr int kBlock = kSlabHistBlock;
    __shared__ unsigned s_h[kSlabBins];
    for (int i = threadIdx.x; i < kSlabBins; i += kBlock) s_h[i] = 0;
    __syncthreads();
    const int64_t groups = (n + 3) / 4;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
        float v[12];
        const int have = load_four_points<VEC>(xyz, g, n, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < have) atomicAdd(&s_h[slab_bin(axis == 0 ? v[3 * j] : axis == 1 ? v[3 * j + 1] : v[3 * j + 2], x0, inv)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSlabBins; i += kBlock)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// cuts: part p owns the bins [cut[p], cut[p+1]) where cut[p] = the first bin with floor(p n / parts) points below it
__global__ __launch_bounds__(kBlock) void k_slab_cut(const unsigned* __restrict__ hist, long long n, int parts, int* __restrict__ cut,
                                                     long long* __restrict__ counts) {
    __shared__ long long s_cum[kSlabBins + 1];
    __shared__ long long s_part[kBlock];
    __shared__ int s_cut[PCT_SLAB_PARTS_MAX + 1];
    constexpr int per = kSlabBins / kBlock;
    long long mine = 0;
    for (int j = 0; j < per; ++j) mine += hist[threadIdx.x * per + j];
    s_part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int t = 0; t < kBlock; ++t) { const long long c = s_part[t]; s_part[t] = run; run += c; }
    }
    __syncthreads();
    long long run = s_part[threadIdx.x];
    for (int j = 0; j < per; ++j) { s_cum[threadIdx.x * per + j] = run; run += hist[threadIdx.x * per + j]; }
    if (threadIdx.x == kBlock - 1) s_cum[kSlabBins] = run;
    __syncthreads();
    if ((int)threadIdx.x <= parts) {
        const int p = threadIdx.x;
        int b = kSlabBins;
        if (p < parts) {
            const long long want = (long long)p * n / parts;
            int lo = 0, hi = kSlabBins;                // smallest b with cum[b] >= want (cum is non-decreasing)
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cum[mid] >= want) hi = mid; else lo = mid + 1; }
            b = lo;
        }
        s_cut[p] = b;
        cut[p] = b;
    }
    __syncthreads();
    if ((int)threadIdx.x < parts) counts[threadIdx.x] = s_cum[s_cut[threadIdx.x + 1]] - s_cum[s_cut[threadIdx.x]];
}

// SLAB: the owned points are those of the slab (compacted to the front through a second counter, in whatever order the
// blocks arrive: the public index rides in w); otherwise the rows [q_begin, q_end), in their own order.
template <bool SLAB, bool VEC>
__global__ __launch_bounds__(kCullBlock) void k_cull_pack(const float* __restrict__ xyz, int64_t n, Box3 box, int64_t q_begin,
                                                      int64_t q_end, SlabCut cut, float4* __restrict__ pts4, unsigned* __restrict__ kept_others,
                                                      PackRed* __restrict__ red, PackRed* __restrict__ parts) {
    constexpr int kBlock = kCullBlock, NW = kCullBlock / 64;
    __shared__ int s_cnt[16][NW];
    __shared__ int s_own[16][NW];
    __shared__ int s_base, s_own_base;
    const float sh[3] = {0.f, 0.f, 0.f};
    PackAcc acc;
    acc.init();
    const int64_t base = (int64_t)blockIdx.x * kCullChunk;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n_owned = q_end - q_begin;
    float px[16], py[16], pz[16];
    unsigned keep_bits = 0, own_bits = 0;
    int bad = 0;
    // slot r of a thread = point 4 (chunk's first group + (r / 4) kBlock + thread) + r % 4: four consecutive points per
    // 48-byte load (load_four_points)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        float v[12];
        load_four_points<VEC>(xyz, base / 4 + rr * kBlock + threadIdx.x, n, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) { px[4 * rr + j] = v[3 * j]; py[4 * rr + j] = v[3 * j + 1]; pz[4 * rr + j] = v[3 * j + 2]; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t i = base + 4 * ((r >> 2) * kBlock + threadIdx.x) + (r & 3);
        bool keep = false, own = false;
        if (i < n) {
            bad |= !(isfinite(px[r]) && isfinite(py[r]) && isfinite(pz[r]));
            if (SLAB) {
                const int b = slab_bin(cut.axis == 0 ? px[r] : cut.axis == 1 ? py[r] : pz[r], cut.x0, cut.inv);
                own = b >= cut.bin_lo && b < cut.bin_hi;
                keep = !own && in_box(box, px[r], py[r], pz[r]);
            } else if (i >= q_begin && i < q_end) {
                pts4[i - q_begin] = make_float4(px[r], py[r], pz[r], __int_as_float((int)i));
                acc.add(px[r], py[r], pz[r], sh);
            } else {
                keep = in_box(box, px[r], py[r], pz[r]);
            }
        }
        keep_bits |= keep ? 1u << r : 0u;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_cnt[r][w] = (int)__popcll(m);
        if (SLAB) {
            own_bits |= own ? 1u << r : 0u;
            const unsigned long long mo = __ballot(own);
            if (lane == 0) s_own[r][w] = (int)__popcll(mo);
        }
    }
    __syncthreads();
    // exclusive prefix over (round, wave) by one wave per list, one counter increment per chunk
    if (w == 0 || (SLAB && w == 1)) {
        int* list = w == 0 ? &s_cnt[0][0] : &s_own[0][0];
        constexpr int per = 16 * NW / 64;
        int c[per], mine = 0;
#pragma unroll
        for (int j = 0; j < per; ++j) { c[j] = list[lane * per + j]; mine += c[j]; }
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        int run = incl - mine;
#pragma unroll
        for (int j = 0; j < per; ++j) { list[lane * per + j] = run; run += c[j]; }
        const int total = __shfl(incl, 63);
        if (lane == 0) {
            if (w == 0) s_base = total ? (int)atomicAdd(kept_others, (unsigned)total) : 0;
            else s_own_base = total ? (int)atomicAdd(kept_others + 32, (unsigned)total) : 0;       // (its own 128-byte line)
        }
    }
    __syncthreads();
    const int64_t out0 = n_owned + s_base;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool keep = (keep_bits >> r) & 1u;
        const unsigned long long m = __ballot(keep);
        const int64_t i = base + 4 * ((r >> 2) * kBlock + threadIdx.x) + (r & 3);
        if (keep) {
            const int64_t at = out0 + s_cnt[r][w] + (int)__popcll(m & ((1ull << lane) - 1ull));
            if (!SLAB || at < n) pts4[at] = make_float4(px[r], py[r], pz[r], __int_as_float((int)i));
            acc.add(px[r], py[r], pz[r], sh);
        }
        if (SLAB) {
            const bool own = (own_bits >> r) & 1u;
            const unsigned long long mo = __ballot(own);
            if (own) {
                const int64_t at = (int64_t)s_own_base + s_own[r][w] + (int)__popcll(mo & ((1ull << lane) - 1ull));
                if (at < n_owned) pts4[at] = make_float4(px[r], py[r], pz[r], __int_as_float((int)i));   // (guard: a cut that no longer fits the buffer is reported by the host)
                acc.add(px[r], py[r], pz[r], sh);
            }
        }
    }
    if (__any(bad) && lane == 0) atomicOr(&red->bad, 1);
    acc.commit<NW>(parts + blockIdx.x);
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
                                                 const float* __restrict__ own_want, float own_lo, float own_hi, int skip_outside,
                                                 int* __restrict__ cell_of, int* __restrict__ rank_of,
                                                 int* __restrict__ cell_own, int* __restrict__ cell_oth) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = pts4[i];
    if (skip_outside) {
        // fast level pass over a sub-box: a point outside the box would only be clamped into a boundary cell that no
        // owned stencil touches -- and a million of them into the same few cells serialise the atomics.  Leave it out.
        const double fx = floor(((double)p.x - g.ox) * g.inv_cell), fy = floor(((double)p.y - g.oy) * g.inv_cell),
                     fz = floor(((double)p.z - g.oz) * g.inv_cell);
        if (fx < 0 || fx >= g.nx || fy < 0 || fy >= g.ny || fz < 0 || fz >= g.nz) {
            cell_of[i] = -1;
            return;
        }
    }
    int cx = cell_coord((double)p.x, g.ox, g.inv_cell, g.nx);
    int cy = cell_coord((double)p.y, g.oy, g.inv_cell, g.ny);
    int cz = cell_coord((double)p.z, g.oz, g.inv_cell, g.nz);
    int c = (cz * g.ny + cy) * g.nx + cx;
    cell_of[i] = c;
    // level passes (pct_levels.hip) own the still unanswered points whose wanted cell edge (log2) lies in a band
    const float want = own_want ? own_want[__float_as_int(p.w)] : 0.f;
    const bool owned = own_want ? (want >= own_lo && want < own_hi) : (i >= q_begin && i < q_end);
    if (owned) {
        rank_of[i] = atomicAdd(&cell_own[c], 1);
    } else {
        // fast level pass: a cell that already holds far more candidates than the sweep can stage overflows every
        // stencil it is part of (those queries go to another pass whatever happens), so further candidates in it are
        // dead weight -- and a hundred thousand of them on one counter serialise (~88 atomics/us per address)
        if (skip_outside && __builtin_nontemporal_load(&cell_oth[c]) > 2048) {
            cell_of[i] = -1;
            return;
        }
        rank_of[i] = atomicAdd(&cell_oth[c], 1) | (int)0x80000000;
    }
}

// k_pack folded in (a handle fed a stream of similar clouds builds on the previous cloud's box without waiting for the new
// one: no float4 copy of the cloud is needed before the histogram): reads the caller's xyz rows, bins them, and leaves the
// bounding box / moments / finite check of the cloud as per-block records for k_pack_final.  Whole-cloud handles only
// (every point owned or not by its index; no level pass, no sub-box).
__global__ __launch_bounds__(kBlock) void k_hist_raw(const float* __restrict__ xyz, int64_t n, pct_grid g, int q_begin, int q_end,
                                                     int* __restrict__ cell_of, int* __restrict__ rank_of,
                                                     int* __restrict__ cell_own, int* __restrict__ cell_oth, PackRed* __restrict__ parts) {
    const float sh[3] = {0.f, 0.f, 0.f};
    PackAcc acc;
    acc.init();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const bool ok = isfinite(x) && isfinite(y) && isfinite(z);
        acc.bad |= !ok;
        if (ok) acc.add(x, y, z, sh);
        // (a non-finite row makes the call fail at its next read-back; it is binned at the origin meanwhile)
        const int cx = cell_coord(ok ? (double)x : g.ox, g.ox, g.inv_cell, g.nx);
        const int cy = cell_coord(ok ? (double)y : g.oy, g.oy, g.inv_cell, g.ny);
        const int cz = cell_coord(ok ? (double)z : g.oz, g.oz, g.inv_cell, g.nz);
        const int c = (cz * g.ny + cy) * g.nx + cx;
        cell_of[i] = c;
        if (i >= q_begin && i < q_end) rank_of[i] = atomicAdd(&cell_own[c], 1);
        else rank_of[i] = atomicAdd(&cell_oth[c], 1) | (int)0x80000000;
    }
    if (parts) acc.commit(parts + blockIdx.x);
}

// The same for the later passes of the density-adaptive sweep (pct_levels.hip), whose cells are sized for ONE band of
// densities while every point is still a candidate: the denser part of the cloud then piles into a handful of cells
// and its atomics serialise at the memory side (~88 per microsecond and address: a 1/r^2 scan spent 1-2 ms per pass
// there).  These passes read the points in the cell order of the first pass, so the lanes of a wave mostly share
// their cell: one atomic per distinct (cell, class) of the wave, ranks handed out by bit counts.
__global__ __launch_bounds__(kBlock) void k_hist_agg(const float4* __restrict__ pts4, int64_t n, pct_grid g,
                                                     const float* __restrict__ own_want, float own_lo, float own_hi, int skip_outside,
                                                     int* __restrict__ cell_of, int* __restrict__ rank_of,
                                                     int* __restrict__ cell_own, int* __restrict__ cell_oth) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int key = -1;                       // 2 * cell + (owned ? 1 : 0); -1 = not counted
    if (i < n) {
        const float4 p = pts4[i];
        bool inside = true;
        if (skip_outside) {
            const double fx = floor(((double)p.x - g.ox) * g.inv_cell), fy = floor(((double)p.y - g.oy) * g.inv_cell),
                         fz = floor(((double)p.z - g.oz) * g.inv_cell);
            inside = !(fx < 0 || fx >= g.nx || fy < 0 || fy >= g.ny || fz < 0 || fz >= g.nz);
        }
        if (inside) {
            const int cx = cell_coord((double)p.x, g.ox, g.inv_cell, g.nx);
            const int cy = cell_coord((double)p.y, g.oy, g.inv_cell, g.ny);
            const int cz = cell_coord((double)p.z, g.oz, g.inv_cell, g.nz);
            const int c = (cz * g.ny + cy) * g.nx + cx;
            const float w = own_want[__float_as_int(p.w)];
            const bool owned = w >= own_lo && w < own_hi;
            // a cell that already holds far more candidates than the sweep can stage overflows every stencil it is
            // part of: further candidates in it are dead weight (its count stays above the staging capacity)
            if (owned || !skip_outside || __builtin_nontemporal_load(&cell_oth[c]) <= 2048) key = 2 * c + (owned ? 1 : 0);
        }
    }
    int rank = 0;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(key >= 0);
    while (todo) {
        const int leader = (int)__builtin_ctzll(todo);
        const int k0 = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(key == k0);
        int base = 0;
        if (lane == leader) base = atomicAdd((k0 & 1) ? &cell_own[k0 >> 1] : &cell_oth[k0 >> 1], (int)__popcll(same));
        base = __builtin_amdgcn_readlane(base, leader);
        if (key == k0) rank = base + (int)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (i < n) {
        cell_of[i] = key < 0 ? -1 : key >> 1;
        if (key >= 0) rank_of[i] = (key & 1) ? rank : (rank | (int)0x80000000);
    }
}

// ---- triple exclusive scan over the cells ---------------------------------
//   x: all points      -> cell_start   (position of the cell in the sorted cloud)
//   y: work items      -> item rank    (ceil(owned / items_q) per cell)
//   z: owned points    -> own_start    (first neighbour-table row of the cell)
constexpr int kScanItems = 8;                    // per thread
constexpr int kScanTile = kBlock * kScanItems;   // 2048 cells per block

// (x, y, z are scanned; w just counts the non-empty cells along)
__device__ __forceinline__ int4 add3(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ int4 cell_counts(const int* __restrict__ own, const int* __restrict__ oth, int64_t c, int64_t ncell, int items_q) {
    if (c >= ncell) return make_int4(0, 0, 0, 0);
    const int o = own[c];
    const int t = o + (oth ? oth[c] : 0);
    return make_int4(t, (o + items_q - 1) / items_q, o, t > 0 ? 1 : 0);
}

// first pass: per-tile sums; also accumulates sum_c owned_c * count_c (= sum over the owned points of the
// population of their own cell), the statistic the cell-size loop steers on
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
        sq += (unsigned long long)x.z * (unsigned)x.x;       // owned points x population of their cell
    }
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_xor(v.x, o); v.y += __shfl_xor(v.y, o); v.z += __shfl_xor(v.z, o); v.w += __shfl_xor(v.w, o);
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
struct ScanTotals { unsigned long long sumsq; int4 tot; };

// (also clears the sweep's eight statistics words and its redo count, which saves the sweep a 64-byte memset launch)
__global__ __launch_bounds__(1024) void k_scan_tiles(int4* __restrict__ tmp, int nblk, const unsigned long long* __restrict__ sq_part,
                                                      ScanTotals* __restrict__ host_copy, unsigned long long* __restrict__ zero8) {
    __shared__ int4 sh[1024];
    if (zero8 && threadIdx.x < 8) zero8[threadIdx.x] = 0ull;
    __shared__ int4 carry;
    __shared__ unsigned long long shq[16];
    {   // sum of the per-tile sum-of-squares partials
        unsigned long long q = 0;
        for (int i = threadIdx.x; i < nblk; i += 1024) q += sq_part[i];
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if ((threadIdx.x & 63) == 0) shq[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long t = 0; for (int i = 0; i < 16; ++i) t += shq[i]; host_copy->sumsq = t; }
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
    if (threadIdx.x == 0) { tmp[nblk] = carry; host_copy->tot = carry; }
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

// counting-sort scatter, owned points first inside every cell; also records for every owned point its
// neighbour-table row and for every row its sorted position
__global__ __launch_bounds__(kBlock) void k_scatter(const float4* __restrict__ pts4, const int* __restrict__ cell_of,
                                                    const int* __restrict__ cell_start, const int* __restrict__ cell_own,
                                                    const int* __restrict__ own_start, const int* __restrict__ rank_of,
                                                    int64_t n, int g_begin, float4* __restrict__ sorted4,
                                                    int* __restrict__ row_of, int* __restrict__ owned_pos,
                                                    const double4* __restrict__ pts4d, double4* __restrict__ sorted4d) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    if (c < 0) return;                  // left out of this pass's cell list
    const int r = rank_of[i];
    int pos, row;
    if (r >= 0) {
        pos = cell_start[c] + r;
        row = own_start[c] + r;
        owned_pos[row] = pos;
        row_of[i - g_begin] = row;      // owned points are contiguous in the packed array
    } else {
        pos = cell_start[c] + cell_own[c] + (r & 0x7fffffff);
    }
    const float4 p = pts4[i];
    sorted4[pos] = p;
    if (pts4d) sorted4d[pos] = pts4d[__float_as_int(p.w)];      // float64 records stay in public order (never culled)
}

// ... from the caller's xyz rows (k_hist_raw's counterpart)
__global__ __launch_bounds__(kBlock) void k_scatter_raw(const float* __restrict__ xyz, const int* __restrict__ cell_of,
                                                        const int* __restrict__ cell_start, const int* __restrict__ cell_own,
                                                        const int* __restrict__ own_start, const int* __restrict__ rank_of,
                                                        int64_t n, int g_begin, float4* __restrict__ sorted4,
                                                        int* __restrict__ row_of, int* __restrict__ owned_pos,
                                                        const double4* __restrict__ pts4d, double4* __restrict__ sorted4d) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    const int r = rank_of[i];
    int pos;
    if (r >= 0) {
        pos = cell_start[c] + r;
        const int row = own_start[c] + r;
        owned_pos[row] = pos;
        row_of[i - g_begin] = row;
    } else {
        pos = cell_start[c] + cell_own[c] + (r & 0x7fffffff);
    }
    sorted4[pos] = make_float4(xyz[3 * i + 0], xyz[3 * i + 1], xyz[3 * i + 2], __int_as_float((int)i));
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
static PackRed* red_parts(pct_ctx* ctx) { return (PackRed*)((char*)ctx->red.p + 128); }

static int red_reset(pct_ctx* ctx, int n_parts, bool zero = true) {
    PCT_TRY(pct_reserve(ctx, &ctx->red, 128 + (size_t)n_parts * sizeof(PackRed)));
    if (zero) PCT_HIP(ctx, hipMemsetAsync(ctx->red.p, 0, 128, ctx->stream));
    return PCT_OK;
}

// folds the n_parts block records of the pass just launched (the result also lands in pinned host memory)
static int red_fold(pct_ctx* ctx, int n_parts, bool carry = true) {
    PCT_LAUNCH(k_pack_final, dim3(1), dim3(kBlock), 0, ctx->stream, (const PackRed*)red_parts(ctx), n_parts,
                       (PackRed*)ctx->red.p, (PackRed*)ctx->pin, carry ? 1 : 0);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

// ... and reads the result back
static int red_read(pct_ctx* ctx, int n_parts, PackRed* out, float* bbox, bool carry = true) {
    PCT_TRY(red_fold(ctx, n_parts, carry));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->pin, sizeof(*out));
    for (int a = 0; a < 6; ++a) bbox[a] = order_float(out->bb[a]);
    return PCT_OK;
}

// every point, public order
// defer: do not wait for the result (the caller builds on the previous call's box and checks at its next
// synchronisation, pct_build_grid)
static int pack_all(pct_ctx* ctx, float* bbox, PackRed* red, bool defer = false) {
    const int64_t n = ctx->n;
    PCT_TRY(pct_reserve(ctx, &ctx->pts4, (size_t)n * sizeof(float4)));
    const int nb = grid_1d(n, kBlock * 4, 512);
    PCT_TRY(red_reset(ctx, nb, false));
    PCT_LAUNCH(k_pack, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->xyz_view, n, (float4*)ctx->pts4.p, red_parts(ctx));
    PCT_HIP(ctx, hipGetLastError());
    if (defer) {
        PCT_TRY(red_fold(ctx, nb, false));
    } else {
        PCT_TRY(red_read(ctx, nb, red, bbox, false));
        if (red->bad) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points");
    }
    ctx->pts4_valid = true;
    ctx->n_grid = n;
    ctx->g_begin = ctx->q_begin;
    ctx->culled = false;
    return PCT_OK;
}

int pct_pack_points(pct_ctx* ctx, float* bbox /*6*/) {
    PackRed red;
    return pack_all(ctx, bbox, &red);
}

// Sharded handle: pack only the points inside the owned rows' bounding box grown by `margin_cells` first-guess
// cell edges.  *kept_box receives the box that was applied (its faces become pct_grid::lim_*).
// Ownership by slab: bounding box of the whole cloud, histogram along its longest axis, the cuts of all parts (on the
// device; the cuts and the populations come back through pinned memory).  Leaves q_begin = 0, q_end = population of
// this handle's slab: the owned points will lead the packed array.
static int slab_split(pct_ctx* ctx) {
    const int64_t n = ctx->n;
    const int parts = ctx->slab_parts;
    const int nb = grid_1d(n, kBlock * 4, 2048);
    const bool vec = ((uintptr_t)ctx->xyz_view & 15) == 0;
    PackRed red;
    float ob[6];
    PCT_TRY(red_reset(ctx, nb));
    if (vec) PCT_LAUNCH(k_cloud_box<true>, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->xyz_view, n, red_parts(ctx));
    else PCT_LAUNCH(k_cloud_box<false>, dim3(nb), dim3(kBlock), 0, ctx->stream, ctx->xyz_view, n, red_parts(ctx));
    PCT_HIP(ctx, hipGetLastError());
    PCT_TRY(red_read(ctx, nb, &red, ob));
    if (red.bad) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points");
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (ob[3 + a] - ob[a] > ob[3 + axis] - ob[axis]) axis = a;
    const double ext = (double)ob[3 + axis] - ob[axis];
    ctx->slab_axis = axis;
    ctx->slab_x0 = ob[axis];
    ctx->slab_inv = ext > 0 && isfinite((float)(kSlabBins / ext)) ? (float)(kSlabBins / ext) : 0.f;
    for (int a = 0; a < 6; ++a) ctx->slab_bbox[a] = ob[a];
    PCT_TRY(pct_reserve(ctx, &ctx->scan_tmp, 256 + kSlabBins * sizeof(unsigned)));
    unsigned* hist = (unsigned*)((char*)ctx->scan_tmp.p + 256);
    PCT_HIP(ctx, hipMemsetAsync(hist, 0, kSlabBins * sizeof(unsigned), ctx->stream));
    const int nh = grid_1d(n, kSlabHistBlock * 4, 512);
    if (vec) PCT_LAUNCH(k_slab_hist<true>, dim3(nh), dim3(kSlabHistBlock), 0, ctx->stream, ctx->xyz_view, n, axis, ctx->slab_x0, ctx->slab_inv, hist);
    else PCT_LAUNCH(k_slab_hist<false>, dim3(nh), dim3(kSlabHistBlock), 0, ctx->stream, ctx->xyz_view, n, axis, ctx->slab_x0, ctx->slab_inv, hist);
    int* h_cut = (int*)(ctx->pin + 2560);                       // mapped pinned memory: the kernel writes the host's copy
    long long* h_counts = (long long*)(ctx->pin + 3072);
    PCT_LAUNCH(k_slab_cut, dim3(1), dim3(kBlock), 0, ctx->stream, (const unsigned*)hist, (long long)n, parts, h_cut, h_counts);
    PCT_HIP(ctx, hipGetLastError());
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    long long total = 0;
    for (int p = 0; p < parts; ++p) { ctx->slab_counts[p] = h_counts[p]; total += h_counts[p]; }
    if (total != n || h_cut[0] != 0 || h_cut[parts] != kSlabBins)
        return pct_fail(ctx, PCT_ERR_INVALID, "slab cut: %lld of %lld points in %d parts", total, (long long)n, parts);
    ctx->slab_bin_lo = h_cut[ctx->slab_part];
    ctx->slab_bin_hi = h_cut[ctx->slab_part + 1];
    ctx->q_begin = 0;
    ctx->q_end = ctx->slab_counts[ctx->slab_part];
    ctx->slab_split_valid = true;
    return PCT_OK;
}

static int pack_near_owned(pct_ctx* ctx, double target, float* bbox, PackRed* red, Box3* kept_box) {
    const int64_t n = ctx->n, n_owned = ctx->q_end - ctx->q_begin;
    const bool slab = ctx->slab_parts >= 1;
    Box3 box;
    if (slab) {
        // the slab and 4 first-guess cell edges either side of it along the cut axis, everything across; the outer
        // faces of the first and the last slab are open.  After a limit retry (no_cull): every point.
        const float* ob = ctx->slab_bbox;
        const double ex = (double)ob[3] - ob[0], ey = (double)ob[4] - ob[1], ez = (double)ob[5] - ob[2];
        const double emax = fmax(ex, fmax(ey, ez));
        double area = 1.2 * (ex * ey + ey * ez + ex * ez);
        if (!(area > 0)) area = emax * emax;
        double a0 = sqrt(target * area / (double)n);
        if (!(a0 > 0) || !isfinite(a0)) a0 = emax;
        double margin = 4.0 * a0;
        if (const char* e = pct_getenv("PCT_SLAB_MARGIN")) margin = atof(e) * a0;      // test aid: a thin margin forces the limit retry
        for (int a = 0; a < 3; ++a) { box.lo[a] = -INFINITY; box.hi[a] = INFINITY; }
        if (!ctx->no_cull && ctx->slab_inv > 0) {
            const int ax = ctx->slab_axis;
            if (ctx->slab_bin_lo > 0) box.lo[ax] = (float)((double)ctx->slab_x0 + (double)ctx->slab_bin_lo / ctx->slab_inv - margin);
            if (ctx->slab_bin_hi < kSlabBins) box.hi[ax] = (float)((double)ctx->slab_x0 + (double)ctx->slab_bin_hi / ctx->slab_inv + margin);
        }
    } else {
    // A handle that is fed a stream of similar clouds (same size, same owned range) reuses the box of the last call
    // instead of measuring the owned rows first (one pass and one host synchronisation less).  A stale box is safe:
    // the owned rows are packed whatever it says, and a query that reaches past a face makes the sweep repeat
    // with every point (pct_grid::lim_*) -- after which the box is measured again.
    const bool reuse = ctx->cull_box_valid && ctx->cull_box_n == n && ctx->cull_box_q0 == ctx->q_begin &&
                       ctx->cull_box_q1 == ctx->q_end;
    if (reuse) {
        for (int a = 0; a < 3; ++a) { box.lo[a] = ctx->cull_box[a]; box.hi[a] = ctx->cull_box[3 + a]; }
    } else {
        const int nb = grid_1d(n_owned, kBlock * 4, 512);
        PCT_TRY(red_reset(ctx, nb));
        PCT_LAUNCH(k_range_box, dim3(nb), dim3(kBlock), 0, ctx->stream,
                           ctx->xyz_view, ctx->q_begin, ctx->q_end, red_parts(ctx));
        PCT_HIP(ctx, hipGetLastError());
        float ob[6];
        PCT_TRY(red_read(ctx, nb, red, ob));
        if (red->bad) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points");
        // margin: 4 cell edges of the cell size the owned points alone would get (an over-estimate of the final
        // edge whenever other handles' points share the region); the sweep checks the assumption per query
        const double ex = (double)ob[3] - ob[0], ey = (double)ob[4] - ob[1], ez = (double)ob[5] - ob[2];
        const double emax = fmax(ex, fmax(ey, ez));
        double area = 1.2 * (ex * ey + ey * ez + ex * ez);
        if (!(area > 0)) area = emax * emax;
        double a0 = sqrt(target * area / (double)n_owned);
        if (!(a0 > 0) || !isfinite(a0)) a0 = emax;
        const float margin = (float)(4.0 * a0);
        for (int a = 0; a < 3; ++a) { box.lo[a] = ob[a] - margin; box.hi[a] = ob[3 + a] + margin; }
        for (int a = 0; a < 3; ++a) { ctx->cull_box[a] = box.lo[a]; ctx->cull_box[3 + a] = box.hi[a]; }
        ctx->cull_box_valid = true;
        ctx->cull_box_n = n;
        ctx->cull_box_q0 = ctx->q_begin;
        ctx->cull_box_q1 = ctx->q_end;
    }
    }
    *kept_box = box;

    const int nchunk = (int)((n + kCullChunk - 1) / kCullChunk);
    PCT_TRY(pct_reserve(ctx, &ctx->pts4, (size_t)n * sizeof(float4)));           // worst case: everything is kept
    PCT_TRY(pct_reserve(ctx, &ctx->scan_tmp, 256));
    PCT_TRY(red_reset(ctx, nchunk));
    PCT_HIP(ctx, hipMemsetAsync(ctx->scan_tmp.p, 0, 33 * sizeof(unsigned), ctx->stream));
    const SlabCut cut = {ctx->slab_axis, ctx->slab_bin_lo, ctx->slab_bin_hi, ctx->slab_x0, ctx->slab_inv};
    const bool vec = ((uintptr_t)ctx->xyz_view & 15) == 0;
#define PCT_CULL(S, V) PCT_LAUNCH((k_cull_pack<S, V>), dim3(nchunk), dim3(kCullBlock), 0, ctx->stream, ctx->xyz_view, n, box, ctx->q_begin, \
                                  ctx->q_end, cut, (float4*)ctx->pts4.p, (unsigned*)ctx->scan_tmp.p, (PackRed*)ctx->red.p, red_parts(ctx))
    if (slab) { if (vec) PCT_CULL(true, true); else PCT_CULL(true, false); }
    else { if (vec) PCT_CULL(false, true); else PCT_CULL(false, false); }
#undef PCT_CULL
    PCT_HIP(ctx, hipGetLastError());
    PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 160, ctx->scan_tmp.p, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 164, (const unsigned*)ctx->scan_tmp.p + 32, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    PCT_TRY(red_read(ctx, nchunk, red, bbox));
    if (red->bad) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points");
    if (slab && (int64_t)((const unsigned*)(ctx->pin + 160))[1] != n_owned)
        return pct_fail(ctx, PCT_ERR_INVALID, "slab pack found %u owned points, the cut said %lld (was the cloud's buffer written meanwhile?)",
                        ((const unsigned*)(ctx->pin + 160))[1], (long long)n_owned);
    const int64_t kept = n_owned + (int64_t)*(const unsigned*)(ctx->pin + 160);
    if ((int64_t)red->cnt != kept || kept > n)
        return pct_fail(ctx, PCT_ERR_INVALID, "cull pass kept %lld / counted %lld points", (long long)red->cnt, (long long)kept);
    ctx->pts4_valid = false;                 // pts4 is not the full public-order pack
    ctx->n_grid = kept;
    ctx->g_begin = 0;                        // the owned rows lead the packed array
    ctx->culled = kept < n;
    return PCT_OK;
}

// Outlier-trimmed grid box.  A few far outliers would stretch the bounding box until the whole cloud falls into
// a handful of cells.  Cell coordinates are clamped, and a clamped point is never nearer than its boundary cell
// suggests, so the grid may cover any sub-box: take mean +- 6 sigma per axis (re-estimated inside the box until
// it settles) and let the points outside share the boundary cells.  Clouds without outliers keep their bbox:
// a uniform or surface-like distribution spans about +-1.7 sigma.
static int trim_box(pct_ctx* ctx, const PackRed& first, float* bbox) {
    PackRed r = first;
    float sh[3] = {0.f, 0.f, 0.f};           // the moments in `first` are about the origin
    for (int pass = 0; pass < 5; ++pass) {
        if (r.cnt < 2) break;
        bool shrunk = false;
        float nb[6];
        for (int a = 0; a < 3; ++a) {
            const double mean = r.s[a] / (double)r.cnt, var = fmax(r.ss[a] / (double)r.cnt - mean * mean, 0.0);
            const double sd = sqrt(var), c = (double)sh[a] + mean;
            const double lo = fmax((double)bbox[a], c - 6.0 * sd), hi = fmin((double)bbox[3 + a], c + 6.0 * sd);
            const double cut = fmax(lo - bbox[a], 0.0) + fmax((double)bbox[3 + a] - hi, 0.0);
            nb[a] = bbox[a];
            nb[3 + a] = bbox[3 + a];
            if (cut > 0.25 * (hi - lo) && hi > lo) {           // worth it only if a good part of the extent goes
                nb[a] = (float)lo;
                nb[3 + a] = (float)hi;
                shrunk = true;
            }
        }
        if (!shrunk) break;
        for (int a = 0; a < 6; ++a) bbox[a] = nb[a];
        Box3 box, shift;
        for (int a = 0; a < 3; ++a) { box.lo[a] = bbox[a]; box.hi[a] = bbox[3 + a]; shift.lo[a] = shift.hi[a] = 0.5f * (bbox[a] + bbox[3 + a]); }
        for (int a = 0; a < 3; ++a) sh[a] = shift.lo[a];
        const int nblk = grid_1d(ctx->n_grid, kBlock * 4, 512);
        PCT_TRY(red_reset(ctx, nblk));
        PCT_LAUNCH(k_box_stats, dim3(nblk), dim3(kBlock), 0, ctx->stream,
                           (const float4*)ctx->pts4.p, ctx->n_grid, box, shift, red_parts(ctx));
        PCT_HIP(ctx, hipGetLastError());
        float inner[6];
        PCT_TRY(red_read(ctx, nblk, &r, inner));
        ++ctx->tm.grid_iters;     // counted with the cell-size passes
    }
    return PCT_OK;
}

int pct_pack_points_f64(pct_ctx* ctx, const double* d_xyz64) {
    const int64_t n = ctx->n;
    PCT_TRY(pct_reserve(ctx, &ctx->pts4d, (size_t)n * sizeof(double4)));
    PCT_TRY(pct_reserve(ctx, &ctx->xyz, (size_t)n * 3 * sizeof(float)));
    PCT_LAUNCH(k_pack_f64, dim3(grid_1d(n, kBlock, 2048)), dim3(kBlock), 0, ctx->stream,
                       d_xyz64, n, (double4*)ctx->pts4d.p, (float*)ctx->xyz.p);
    ctx->xyz_view = (const float*)ctx->xyz.p;
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

static void set_dims(pct_grid* g, const float* bbox, double a) {
    g->ox = bbox[0]; g->oy = bbox[1]; g->oz = bbox[2];
    g->cell = a;
    g->inv_cell = 1.0 / a;
    double ex = (double)bbox[3] - bbox[0], ey = (double)bbox[4] - bbox[1], ez = (double)bbox[5] - bbox[2];
    if (!(ex >= 0)) ex = 0;          // an empty or non-finite box is one cell, not INT_MIN cells
    if (!(ey >= 0)) ey = 0;
    if (!(ez >= 0)) ez = 0;
    // counted in double first: an absurdly small edge (eps = 1e-30) must read as "far too many cells" for the caller's
    // budget loop, not overflow the int32 dimensions
    const double dx = floor(ex * g->inv_cell) + 1, dy = floor(ey * g->inv_cell) + 1, dz = floor(ez * g->inv_cell) + 1;
    if (!(dx * dy * dz < 4.0e18) || !(dx < 2.0e9) || !(dy < 2.0e9) || !(dz < 2.0e9)) {
        g->nx = g->ny = g->nz = 1 << 30;
        g->ncell = (int64_t)1 << 62;
        return;
    }
    g->nx = (int32_t)dx;
    g->ny = (int32_t)dy;
    g->nz = (int32_t)dz;
    g->ncell = (int64_t)g->nx * g->ny * g->nz;
}

// Chooses the cell edge so that a point shares its cell with about
// factor*(k+1) points, then counting-sorts the cloud.  With that occupancy the
// 27-cell stencil (guaranteed radius = one cell edge) contains the k+1 nearest
// points for nearly every query of a surface-like cloud; the sweep kernel
// widens ring by ring for the rest.
int pct_build_grid(pct_ctx* ctx, int32_t k, double eps) {
    // measured optima on surface clouds (tools/tune_factor.py): larger cells cost candidates, smaller ones cost
    // trips to the exact sweep; the LDS staging capacity caps the large side
    const double factor = ctx->occupancy_factor > 0 ? ctx->occupancy_factor : pct_default_factor(k);
    const double target = factor * (k + 1);
    // (the first pass of a chained sweep only has to be roughly right: the later passes adapt)
    // cell budget: 2^27 cells (1.6 GB of counters and starts) serve surfaces of up to a few million points; beyond,
    // 32 cells per point up to 2^30 (a 60 M-point torus wants 3.3e8 cells: capped at 2^27 its cells held 51 points
    // instead of 29.5, the staging area overflowed and the sweep took 235 ms instead of ~40)
    int64_t cell_cap = (int64_t)1 << 27;
    if (ctx->level_mode) cell_cap = (int64_t)1 << 24;
    else if (32 * ctx->n > cell_cap) cell_cap = 32 * ctx->n < ((int64_t)1 << 30) ? 32 * ctx->n : (int64_t)1 << 30;
    const float* own_flag = (const float*)ctx->own_flag;   // level passes: ownership by wanted-edge band
    const bool slab = ctx->slab_parts >= 1 && !own_flag;
    if (slab) PCT_TRY(slab_split(ctx));                    // (sets the owned count: q_begin = 0, q_end = this slab's population)
    const int64_t n_owned = own_flag ? ctx->own_count : ctx->q_end - ctx->q_begin;
    const bool sharded = own_flag || ctx->q_begin > 0 || ctx->q_end < ctx->n || slab;   // some points are candidates only

    float bbox[6];
    PackRed red;
    Box3 kept_box = {};
    // (a slab's owned points are found by the pack itself: it always runs, with an open box after a limit retry)
    const bool try_cull = slab ? n_owned > 0
                               : !own_flag && sharded && n_owned > 0 && !ctx->has_f64 && !ctx->no_cull && !pct_getenv("PCT_NO_CULL");
    ctx->tm.grid_iters = 0;
    // Speculation: a handle fed a stream of similar clouds builds the cell list over the (trimmed) box of the previous
    // call without waiting for the new bounding box -- any box is a valid grid box, points outside are clamped into
    // the boundary cells -- and checks the new box at the synchronisation that ends the first pass.  One host round
    // trip less per build.
    bool spec = !try_cull && !own_flag && !ctx->level_mode && ctx->hint_edge > 0 && ctx->spec_valid && ctx->spec_n == ctx->n &&
                !pct_getenv("PCT_NO_SPEC");
    // later passes of the density-adaptive sweep: the points in the cell order of its first pass, and that pass's box
    const bool from_base = own_flag && ctx->lvl_src_valid;
    if (from_base) {
        for (int a = 0; a < 6; ++a) bbox[a] = ctx->lvl_bbox[a];
        red = PackRed{};
        ctx->n_grid = ctx->n;
        ctx->g_begin = 0;
        ctx->culled = false;
    } else
    if (try_cull) {
        PCT_TRY(pack_near_owned(ctx, target, bbox, &red, &kept_box));
        if (ctx->n_grid < (int64_t)k + 1) {
            // The kept part cannot even fill a row (a few owned rows and a culling box -- possibly a stale one of the
            // previous cloud -- that holds nobody else): the table would carry "missing" entries that the fused fit,
            // launched before the limits are checked, must never see without a count array.  Take every point.
            ctx->no_cull = true;
            ctx->cull_box_valid = false;
            return pct_build_grid(ctx, k, eps);
        }
        PCT_TRY(trim_box(ctx, red, bbox));
    } else if (spec) {
        // no pack pass at all: the histogram reads the caller's rows and takes the cloud's box along (k_hist_raw)
        PCT_TRY(red_reset(ctx, grid_1d(ctx->n, kBlock, 0), false));
        ctx->pts4_valid = false;
        ctx->n_grid = ctx->n;
        ctx->g_begin = ctx->q_begin;
        ctx->culled = false;
        for (int a = 0; a < 6; ++a) bbox[a] = ctx->spec_bbox[a];
    } else {
        PCT_TRY(pack_all(ctx, bbox, &red));
        PCT_TRY(trim_box(ctx, red, bbox));
    }
    const bool raw = spec;                                                // (spec is cleared once the box has been checked)
    const bool sub_box = ctx->level_edge > 0 && ctx->level_box_valid;
    const int64_t n = ctx->n_grid;                                        // points the grid holds
    const int g_begin = own_flag ? 0 : (int)ctx->g_begin, g_end = own_flag ? 0 : (int)(ctx->g_begin + n_owned);

    double ex = (double)bbox[3] - bbox[0], ey = (double)bbox[4] - bbox[1], ez = (double)bbox[5] - bbox[2];
    double emax = fmax(ex, fmax(ey, ez));
    if (!(emax > 0)) emax = 1.0;
    // first guess: the cloud is a surface whose area is about the bbox's half-surface * 1.2
    double area = 1.2 * (ex * ey + ey * ez + ex * ez);
    if (!(area > 0)) area = emax * emax;
    double a = sqrt(target * area / (double)n);
    if (!(a > 0) || !isfinite(a)) a = emax;
    // Warm start: a handle that sees a stream of similar clouds (same scanner, same shard of the same job) reuses
    // the edge the last build converged to, rescaled by the first-guess ratio, and so normally needs one pass.
    const double first_guess_raw = a;
    bool hinted = false;                                  // the first edge comes from the previous cloud on this handle
    const bool level_pass = ctx->level_edge > 0;          // a later level of the density-adaptive sweep: edge given
    if (level_pass) {
        a = ctx->level_edge;
    } else if (ctx->hint_edge > 0 && ctx->hint_guess > 0) {
        const double r = first_guess_raw / ctx->hint_guess * sqrt(ctx->hint_target / target);   // guess ~ sqrt(target)
        if (r > 0.5 && r < 2.0) { a = ctx->hint_edge * r * sqrt(target / ctx->hint_target); hinted = true; }
    }
    a = fmin(a, emax * 1.0001 + 1e-30);

    PCT_TRY(pct_reserve(ctx, &ctx->cell_of, (size_t)n * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->cell_fill, (size_t)n * sizeof(int)));   // in-cell arrival ranks
    // queries per work item: 16 - 20 measured best for k_knn_pair and k_knn_duo (one staged stencil serves more queries; a
    // cell of ~28 points is one or two items; 1 M torus, k = 50: sweep 0.376 | 0.366 | 0.363 | 0.364 | 0.370 ms at 12 | 14 | 18 |
    // 20 | 24; the reference's lattice torus 0.510 | 0.489 | 0.500 at 12 | 16 | 18; k = 80: 0.597 | 0.574 | 0.583 --
    // tools/items_q_probe.py, tools/lattice_probe.py)
    int items_q = 16;
    if (const char* e = pct_getenv("PCT_ITEMS_Q")) { const int v = atoi(e); if (v >= 1 && v <= 64) items_q = v; }   // tuning aid
    ctx->items_q = items_q;
    PCT_TRY(pct_reserve(ctx, &ctx->sorted4, (size_t)n * sizeof(float4)));
    PCT_TRY(pct_reserve(ctx, &ctx->row_of, (size_t)((own_flag ? n : n_owned) + 1) * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->owned_pos, (size_t)n_owned * sizeof(int)));
    if (ctx->has_f64) PCT_TRY(pct_reserve(ctx, &ctx->sorted4d, (size_t)n * sizeof(double4)));
    int nblk = 0;
    pct_grid g = {};
    double a_prev = 0, m_prev = 0;
    int iters = 0;
    // a level pass takes the edge it is given: its owned set is a mix of densities, and the mean population this
    // loop steers on would be pulled to the dense minority (pct_levels.hip sizes by the geometric mean instead)
    const int max_iter = level_pass ? 1 : ctx->level_mode ? 2 : 8;
    const double win_lo = 0.88, win_hi = 1.12;
    int4 tot = make_int4(0, 0, 0, 0);
    double m_last = 0, d_last = 2.0;
    // Every pass runs the whole build (histogram, scan, scatter) and only then reads back the occupancy statistic
    // together with the scan totals: the first cell size is accepted in the common case, which then costs ONE host
    // synchronisation instead of two; a rejected size costs a speculative scatter.
    for (int it = 0; it < max_iter; ++it) {
        if (eps > 0 && a > eps * 1.000001) a = eps * 1.000001;   // one ring already covers the eps ball
        // sub-box of a fast level pass: 2.5 edges of the FINAL cell size around the owned points (their stencils must
        // end inside the box: the points outside it are left out of the cell list)
        const auto set_box = [&]() {
            if (!sub_box) return;
            for (int ax = 0; ax < 3; ++ax) {
                bbox[ax] = ctx->level_box[ax] - (float)(2.5 * a);
                bbox[3 + ax] = ctx->level_box[3 + ax] + (float)(2.5 * a);
            }
        };
        set_box();
        set_dims(&g, bbox, a);
        // PCT_KNN_AUTO: a surface wants about a third of a cell per point; an edge that asks for 16 cells per point was
        // steered there by a dense minority (the size-biased occupancy) -- a 1/r^2 scan wants 130 per point, 1.6 ms of
        // counters and scans that the hierarchical list does not need.  Nothing was built: the caller goes there.
        if (!own_flag && !level_pass && g.ncell > 16 * n && g.ncell > ((int64_t)1 << 20)) {
            if (it == 0 && hinted) {
                // ... unless the edge is the PREVIOUS cloud's (same size, similar box, another density -- a torus after
                // a 1/r^2 scan inherited its 130 cells per point): this cloud's own first guess, then
                hinted = false;
                a = fmin(first_guess_raw, emax * 1.0001 + 1e-30);
                set_box();
                set_dims(&g, bbox, a);
            } else if (ctx->auto_probe) {
                ctx->grid_skewed = true;
                ctx->grid_valid = false;
                return PCT_OK;
            }
        }
        bool hit_cap = false;            // the cell budget, not the occupancy target, set this edge
        while (g.ncell > cell_cap) {
            a *= cbrt((double)g.ncell / (double)cell_cap) * 1.01;
            set_box();
            set_dims(&g, bbox, a);
            hit_cap = true;
        }
        PCT_TRY(pct_reserve(ctx, &ctx->cell_own, (size_t)g.ncell * sizeof(int)));
        PCT_HIP(ctx, hipMemsetAsync(ctx->cell_own.p, 0, (size_t)g.ncell * sizeof(int), ctx->stream));
        if (sharded) {
            PCT_TRY(pct_reserve(ctx, &ctx->cell_oth, (size_t)g.ncell * sizeof(int)));
            PCT_HIP(ctx, hipMemsetAsync(ctx->cell_oth.p, 0, (size_t)g.ncell * sizeof(int), ctx->stream));
        }
        const float4* src = from_base ? (const float4*)ctx->lvl_src.p : (const float4*)ctx->pts4.p;
        if (raw) {
            const int nhb = grid_1d(n, kBlock, 0);
            PCT_LAUNCH(k_hist_raw, dim3(nhb), dim3(kBlock), 0, ctx->stream, ctx->xyz_view, n, g, g_begin, g_end, (int*)ctx->cell_of.p,
                       (int*)ctx->cell_fill.p, (int*)ctx->cell_own.p, sharded ? (int*)ctx->cell_oth.p : nullptr, spec ? red_parts(ctx) : nullptr);
            if (spec) PCT_TRY(red_fold(ctx, nhb, false));
        } else
        if (from_base)
            PCT_LAUNCH(k_hist_agg, dim3(grid_1d(n, kBlock, 0)), dim3(kBlock), 0, ctx->stream, src, n, g, own_flag, ctx->own_lo,
                               ctx->own_hi, sub_box ? 1 : 0, (int*)ctx->cell_of.p, (int*)ctx->cell_fill.p, (int*)ctx->cell_own.p,
                               (int*)ctx->cell_oth.p);
        else
        PCT_LAUNCH(k_hist, dim3(grid_1d(n, kBlock, 0)), dim3(kBlock), 0, ctx->stream,
                           src, n, g, g_begin, g_end, own_flag, ctx->own_lo, ctx->own_hi, sub_box ? 1 : 0, (int*)ctx->cell_of.p,
                           (int*)ctx->cell_fill.p, (int*)ctx->cell_own.p, sharded ? (int*)ctx->cell_oth.p : nullptr);
        nblk = (int)((g.ncell + kScanTile - 1) / kScanTile);
        if (pct_getenv("PCT_GRID_DEBUG"))
            fprintf(stderr, "[grid] pass %d: n %lld owned %lld edge %g dims %d x %d x %d = %lld cells (%d scan tiles), box [%g %g %g]-[%g %g %g]\n", it,
                    (long long)n, (long long)n_owned, a, g.nx, g.ny, g.nz, (long long)g.ncell, nblk, bbox[0], bbox[1], bbox[2], bbox[3], bbox[4], bbox[5]);
        PCT_TRY(pct_reserve(ctx, &ctx->scan_tmp, (size_t)(nblk + 1) * sizeof(int4) + (size_t)nblk * sizeof(unsigned long long)));
        unsigned long long* sq_part = (unsigned long long*)((int4*)ctx->scan_tmp.p + nblk + 1);
        PCT_TRY(pct_reserve(ctx, &ctx->cell_cnt, (size_t)(g.ncell + 1) * sizeof(int)));
        PCT_TRY(pct_reserve(ctx, &ctx->own_start, (size_t)(g.ncell + 1) * sizeof(int)));
        PCT_TRY(pct_reserve(ctx, &ctx->occ, ((size_t)(n_owned < g.ncell ? n_owned : g.ncell) + (size_t)n_owned / items_q + 16) * sizeof(int2)));
        PCT_LAUNCH(k_scan_sums, dim3(nblk), dim3(kBlock), 0, ctx->stream,
                           (const int*)ctx->cell_own.p, sharded ? (const int*)ctx->cell_oth.p : nullptr, g.ncell, items_q,
                           (int4*)ctx->scan_tmp.p, sq_part);
        PCT_TRY(pct_reserve(ctx, &ctx->counters, 64 * sizeof(unsigned long long)));
        PCT_LAUNCH(k_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, (int4*)ctx->scan_tmp.p, nblk,
                           (const unsigned long long*)sq_part, (ScanTotals*)(ctx->pin + 128), (unsigned long long*)ctx->counters.p);
        ctx->counters_clean = true;
        // the totals this pass is judged by (and, with a deferred pack, the cloud's box) are in pinned memory from here on:
        // the host waits for THIS point, not for the end of the stream -- while it wakes up, decides and enqueues the
        // sweep, the device applies the scan and scatters the records (50 us at 1 M points; the read-back used to be ~25
        // us of an idle device).  A pass that is rejected has scattered for nothing, as before.
        PCT_HIP(ctx, hipEventRecord(ctx->ev[8], ctx->stream));
        PCT_LAUNCH(k_scan_apply, dim3(nblk), dim3(kBlock), 0, ctx->stream,
                           (const int*)ctx->cell_own.p, sharded ? (const int*)ctx->cell_oth.p : nullptr, g.ncell, items_q,
                           (const int4*)ctx->scan_tmp.p, (int*)ctx->cell_cnt.p, (int*)ctx->own_start.p, (int2*)ctx->occ.p);
        if (raw)
            PCT_LAUNCH(k_scatter_raw, dim3(grid_1d(n, kBlock, 0)), dim3(kBlock), 0, ctx->stream, ctx->xyz_view, (const int*)ctx->cell_of.p,
                       (const int*)ctx->cell_cnt.p, (const int*)ctx->cell_own.p, (const int*)ctx->own_start.p, (const int*)ctx->cell_fill.p, n,
                       g_begin, (float4*)ctx->sorted4.p, (int*)ctx->row_of.p, (int*)ctx->owned_pos.p,
                       ctx->has_f64 ? (const double4*)ctx->pts4d.p : nullptr, ctx->has_f64 ? (double4*)ctx->sorted4d.p : nullptr);
        else
        PCT_LAUNCH(k_scatter, dim3(grid_1d(n, kBlock, 0)), dim3(kBlock), 0, ctx->stream,
                           src, (const int*)ctx->cell_of.p, (const int*)ctx->cell_cnt.p,
                           (const int*)ctx->cell_own.p, (const int*)ctx->own_start.p, (const int*)ctx->cell_fill.p, n,
                           g_begin, (float4*)ctx->sorted4.p, (int*)ctx->row_of.p, (int*)ctx->owned_pos.p,
                           ctx->has_f64 ? (const double4*)ctx->pts4d.p : nullptr,
                           ctx->has_f64 ? (double4*)ctx->sorted4d.p : nullptr);
        PCT_HIP(ctx, hipGetLastError());
        PCT_HIP(ctx, hipEventSynchronize(ctx->ev[8]));
        if (spec) {                      // the deferred pack result is in: was the old box still right?
            spec = false;
            memcpy(&red, ctx->pin, sizeof(red));
            if (red.bad) return pct_fail(ctx, PCT_ERR_NONFINITE, "Non-finite values in input points");
            bool same = true;
            for (int ax = 0; ax < 3; ++ax) {
                const float lo = order_float(red.bb[ax]), hi = order_float(red.bb[3 + ax]);
                const float tol = 0.02f * (ctx->spec_raw[3 + ax] - ctx->spec_raw[ax]) + 1e-30f;
                same = same && fabsf(lo - ctx->spec_raw[ax]) <= tol && fabsf(hi - ctx->spec_raw[3 + ax]) <= tol;
            }
            if (!same) {                 // a different cloud: start over the regular way
                ctx->spec_valid = false;
                return pct_build_grid(ctx, k, eps);
            }
        }
        const unsigned long long s2 = ((const ScanTotals*)(ctx->pin + 128))->sumsq;
        tot = ((const ScanTotals*)(ctx->pin + 128))->tot;
        ++iters;
        double m = (double)s2 / (double)(n_owned > 0 ? n_owned : 1);
        m_last = m;
        bool eps_bound = eps > 0 && a >= eps;            // cannot grow past eps
        bool capped = g.ncell * 2 > cell_cap && m < target;
        if ((m >= win_lo * target && m <= win_hi * target) || it == max_iter - 1 || (eps_bound && m < target) ||
            capped || (a >= emax && m < target) || (hit_cap && m > target))      // (cannot refine past the cell budget)
            break;
        double d = 2.0;
        if (a_prev > 0 && m != m_prev && a != a_prev) {
            d = log(m / m_prev) / log(a / a_prev);
            if (!(d >= 1.0)) d = 1.0;
            if (d > 3.0) d = 3.0;
        }
        a_prev = a;
        m_prev = m;
        d_last = d;
        double f = pow(target / m, 1.0 / d);
        f = fmin(fmax(f, 1.0 / 16.0), 16.0);
        a = fmin(a * f, emax * 1.0001 + 1e-30);
    }
    // faces of the culling box become limits of what the grid can vouch for (cell units from the origin)
    for (int ax = 0; ax < 3; ++ax) {
        const double o = ax == 0 ? g.ox : ax == 1 ? g.oy : g.oz;
        g.lim_lo[ax] = ctx->culled ? ((double)kept_box.lo[ax] - o) * g.inv_cell : -INFINITY;
        g.lim_hi[ax] = ctx->culled ? ((double)kept_box.hi[ax] - o) * g.inv_cell : INFINITY;
    }
    if (pct_getenv("PCT_GRID_DEBUG"))
        fprintf(stderr, "[grid] box [%g %g %g]-[%g %g %g] edge %g dims %d x %d x %d = %lld cells, %d passes, occupancy %.1f, items %d\n", bbox[0],
                bbox[1], bbox[2], bbox[3], bbox[4], bbox[5], g.cell, g.nx, g.ny, g.nz, (long long)g.ncell, iters, m_last, tot.y);
    if (ctx->level_mode && !own_flag)                    // first pass of a density-adaptive sweep: its (trimmed) box serves the later ones
        for (int a2 = 0; a2 < 6; ++a2) ctx->lvl_bbox[a2] = bbox[a2];
    ctx->grid = g;
    ctx->tm.grid_iters += iters;
    ctx->tm.cells = g.ncell;
    ctx->tm.cell_size = g.cell;
    ctx->tm.grid_points = n;
    ctx->tm.occupancy = m_last;
    if (!try_cull && !own_flag && !ctx->level_mode) {       // remember the box for the next similar cloud
        for (int a2 = 0; a2 < 6; ++a2) ctx->spec_bbox[a2] = bbox[a2];
        if (red.cnt > 0)                  // raw bounding box of this cloud (before trimming)
            for (int a2 = 0; a2 < 6; ++a2) ctx->spec_raw[a2] = order_float(red.bb[a2]);
        ctx->spec_n = ctx->n;
        ctx->spec_valid = true;
    }
    if (m_last > 0 && !level_pass) {        // what the heuristic first guess should have been for this cloud
        ctx->hint_edge = g.cell * pow(target / m_last, 1.0 / d_last);
        ctx->hint_guess = first_guess_raw;
        ctx->hint_target = target;
    }
    if ((tot.x != n && !sub_box) || tot.x > n || tot.z != n_owned)
        return pct_fail(ctx, PCT_ERR_INVALID, "cell scan totals %d/%d != %lld/%lld", tot.x, tot.z, (long long)n, (long long)n_owned);
    // A uniform cell list cannot resolve every cloud (tight clusters very far apart exhaust the cell budget):
    // refuse when the sweep would degenerate into an all-pairs scan of hours rather than run it.
    if (!level_pass && m_last > 64.0 * target && m_last * 27.0 * (double)n_owned > 1e12)
        return pct_fail(ctx, PCT_ERR_INVALID,
                        "the cell list cannot resolve this cloud: a point shares its cell with %.0f others on average at the "
                        "smallest usable cell edge %.3g (%lld cells); thin it out or split it into compact pieces",
                        m_last, g.cell, (long long)g.ncell);
    ctx->n_items = tot.y;
    ctx->nonempty_cells = tot.w;
    ctx->n_occ = tot.y;
    ctx->tm.occupied_cells = tot.y;
    ctx->grid_valid = true;
    return PCT_OK;
}

int pct_launch_gather_int(pct_ctx* ctx, const int* d_map, int* d_inout, int64_t n) {
    PCT_LAUNCH(k_gather_int, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_map, d_inout, n);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
