// Hierarchical cell list: the cloud in Morton order, every query swept at the octree level that suits ITS density.
//
// One cell size serves one density (pct_grid.hip); a chain of cell lists (pct_levels.hip) serves one octave of
// densities per pass and pays a build, a merge and two read-backs for each.  Here the points are sorted ONCE, by the
// Morton code of their position in a 2^21-cube over the bounding box.  A cell of ANY octree level is then a
// contiguous range of the sorted cloud, and so is every cell of its 27-cell stencil: the fast sweep stages 27 ranges
// instead of 9 x-runs and everything behind the staging -- pre-selection, keys, network, proofs -- is the code the
// uniform cell list runs (k_knn_fast<..., TREE>, pct_knn.hip).
//
//   level of a point   first guess: the finest level whose cell around it holds >= n_min points (n_min ~ 0.45 (k+1):
//                      on a surface the disc the 27-cell cube vouches for then holds about k+1).  Found without a
//                      tree: the cell of a point is a run of its neighbours in Morton order, and how far the run
//                      reaches at level l follows from the highest differing bit against each neighbour -- a merge
//                      of two monotone sequences, n_min loads (k_tree_level).
//   segment            a maximal run of consecutive points with the same level and the same cell of that level;
//   stencil            per segment 27 {first position, points}: two binary searches in the code array per cell
//                      (k_tree_stencil);
//   refinement         the own cell says nothing about the neighbours: a sparse cell next to dense ones (the rim of
//                      the cloud, a lone point -- whose own cell is "full" only at the root) gets a stencil far over
//                      the staging capacity.  Such segments are split octant by octant until every piece fits
//                      (k_tree_refine: one wave per segment walks its run depth first); stencil populations are
//                      monotone in the level, so this ends, at the latest at level 0;
//   work item          <= items_q consecutive queries of a segment, {first query | count, segment}.
//
// What the fast sweep cannot prove (too few points inside the radius its cube vouches for, more than 16 non-empty
// stencil cells, a level-0 pile over the capacity) goes to k_knn_exact_tree: the same 27 ranges from global memory,
// one level up whenever the cube cannot vouch for the list -- the root vouches for everything.
// Bit-identical to the exhaustive sweep (tests/test_gpu_parity.py: test_hierarchical_cell_list_on_awkward_clouds,
// test_chain_of_cell_lists_equals_exhaustive_sweep[tree], the fuzzers).  The sort and the two scans are rocPRIM's
// device primitives; everything else here is hand-written.
#include "pct_internal.h"

#include <cstring>
#include <math.h>
#include <time.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int kTreeBits = 21;                       // 3 x 21 = 63 code bits
typedef unsigned long long u64;

__device__ __forceinline__ u64 spread3(unsigned v) { return pct_spread3(v); }
__device__ __forceinline__ unsigned compact3(u64 x) { return pct_compact3(x); }

__global__ __launch_bounds__(256) void k_tree_codes(const float4* __restrict__ pts, int64_t n, double ox, double oy, double oz,
                                                    double inv_f, u64* __restrict__ codes, unsigned* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int top = (1 << kTreeBits) - 1;
    // the very expression the sweep locates a query inside its cell with (k_knn_fast: (x - o) * inv_cell - c)
    const int qx = min(max((int)floor(((double)p.x - ox) * inv_f), 0), top);
    const int qy = min(max((int)floor(((double)p.y - oy) * inv_f), 0), top);
    const int qz = min(max((int)floor(((double)p.z - oz) * inv_f), 0), top);
    codes[i] = spread3((unsigned)qx) | spread3((unsigned)qy) << 1 | spread3((unsigned)qz) << 2;
    vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void k_tree_gather(const float4* __restrict__ pts, const unsigned* __restrict__ vals, int64_t n,
                                                     float4* __restrict__ sorted4, int* __restrict__ owned_pos, int* __restrict__ row_of,
                                                     const double4* __restrict__ ptsd, double4* __restrict__ sorted4d) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const unsigned v = vals[j];
    sorted4[j] = pts[v];
    if (ptsd) sorted4d[j] = ptsd[v];    // float64 clouds: the native coordinates travel with the float32-rounded ones
    owned_pos[j] = (int)j;              // table row = Morton position
    row_of[v] = (int)j;                 // (pts is in public order: v is the public index)
}

// level of the smallest cell that holds both codes
__device__ __forceinline__ int join_level(u64 a, u64 b) {
    const u64 x = a ^ b;
    return x ? (66 - __clzll(x)) / 3 : 0;
}

// finest level whose cell around point j holds >= n_min points (capped: cells beyond max_level are never needed --
// one ring of them already covers the eps ball)
__global__ __launch_bounds__(256) void k_tree_level(const u64* __restrict__ codes, int64_t n, int n_min, int max_level,
                                                    unsigned char* __restrict__ lvl) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u64 c = codes[j];
    int64_t a = j - 1, b = j + 1;
    int la = a >= 0 ? join_level(codes[a], c) : 99, lb = b < n ? join_level(codes[b], c) : 99;
    int l = 0;
    for (int need = n_min - 1; need > 0; --need) {
        if (la <= lb) {
            if (la == 99) { l = kTreeBits; break; }          // the whole cloud is smaller than n_min
            l = la;
            --a;
            la = a >= 0 ? join_level(codes[a], c) : 99;
        } else {
            l = lb;
            ++b;
            lb = b < n ? join_level(codes[b], c) : 99;
        }
    }
    lvl[j] = (unsigned char)min(min(l, kTreeBits), max_level);
}

// bucket[p] = first position whose code is >= p << kTreeBucketShift (p = 0 .. 2^18 + 1)
__global__ __launch_bounds__(256) void k_tree_buckets(const u64* __restrict__ codes, int64_t n, int* __restrict__ bucket) {
    const unsigned p = blockIdx.x * 256 + threadIdx.x;
    if (p > (1u << kTreeBucketBits) + 1u) return;
    bucket[p] = p >= (1u << kTreeBucketBits) ? (int)n : (int)pct_code_lower_bound(codes, n, (u64)p << kTreeBucketShift);
}

// head[j] = j where a segment starts (0 elsewhere; position 0 starts one anyway): an inclusive max-scan turns it into
// the start of the segment of every point
__global__ __launch_bounds__(256) void k_tree_heads(const u64* __restrict__ codes, const unsigned char* __restrict__ lvl, int64_t n,
                                                    int* __restrict__ head) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    bool h = j == 0;
    if (!h) {
        const int l = lvl[j];
        h = lvl[j - 1] != l || (codes[j] >> (3 * l)) != (codes[j - 1] >> (3 * l));
    }
    head[j] = h ? (int)j : 0;
}

// marks: high word 1 at segment starts, low word 1 at item starts (every items_q-th query of a segment)
__global__ __launch_bounds__(256) void k_tree_marks(const int* __restrict__ seg_start, int64_t n, int items_q, u64* __restrict__ marks) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int s = seg_start[j];
    const bool h = s == (int)j;
    const bool it = ((int)j - s) % items_q == 0;
    marks[j] = (h ? 1ull << 32 : 0ull) | (it ? 1ull : 0ull);
}

__device__ __forceinline__ void tree_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A work item: {first query (26 bits: Morton position = table row) | (queries - 1) << 26, segment}; segment < 0 = dead.
__device__ __forceinline__ int2 make_item(int qs, int nq, int seg) { return make_int2((int)((unsigned)qs | (unsigned)(nq - 1) << 26), seg); }

// inclusive sums of the marks -> items, segment headers {level, cx, cy, cz}, segment runs {first position, points}
// and the totals (device words + pinned host words)
__global__ __launch_bounds__(256) void k_tree_items(const u64* __restrict__ codes, const unsigned char* __restrict__ lvl,
                                                    const int* __restrict__ seg_start, const u64* __restrict__ sums, int64_t n,
                                                    int items_q, int2* __restrict__ items, int4* __restrict__ seg_hdr,
                                                    int2* __restrict__ seg_range, int* __restrict__ counts, long long* __restrict__ totals) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u64 s = sums[j];
    const int seg = (int)(s >> 32) - 1, item = (int)(s & 0xffffffffull) - 1;
    const int ss = seg_start[j];
    const bool last = j == n - 1 || seg_start[j + 1] == (int)j + 1;          // last point of its segment
    const int off = (int)j - ss;
    // the item this point closes: full after items_q queries, or short at the end of the segment
    if (off % items_q == items_q - 1 || last) items[item] = make_item((int)j - off % items_q, off % items_q + 1, seg);
    if (ss == (int)j) {
        const int l = lvl[j];
        const u64 c = codes[j];
        seg_hdr[seg] = make_int4(l, (int)(compact3(c) >> l), (int)(compact3(c >> 1) >> l), (int)(compact3(c >> 2) >> l));
    }
    if (last) seg_range[seg] = make_int2(ss, off + 1);
    if (j == n - 1) {
        counts[0] = item + 1;
        counts[1] = seg + 1;
        counts[2] = counts[3] = 0;
        for (int l = 0; l < 22; ++l) counts[4 + l] = 0;       // points per level, over a sample of the segments (k_tree_stencil)
        totals[0] = item + 1;
        totals[1] = seg + 1;
    }
}

// the 27 ranges of the stencil of cell (cx, cy, cz) of level l: lane t < 27 returns range t (centre first)
__device__ __forceinline__ int2 stencil_range(const u64* __restrict__ codes, const int* __restrict__ bucket, int l, int cx, int cy, int cz, int t) {
    int2 r = make_int2(0, 0);
    if (t < 27) {
        const int dim = 1 << (kTreeBits - l);
        int dx, dy, dz;
        pct_stencil_cell(t, &dx, &dy, &dz);
        const int x = cx + dx, y = cy + dy, z = cz + dz;
        if (x >= 0 && x < dim && y >= 0 && y < dim && z >= 0 && z < dim) {
            const u64 prefix = spread3((unsigned)x) | spread3((unsigned)y) << 1 | spread3((unsigned)z) << 2;
            const int64_t lo = pct_code_lower_bound(codes, bucket, prefix << (3 * l));
            const int64_t hi = pct_code_lower_bound(codes, bucket, (prefix + 1) << (3 * l));      // l = 21: 1 << 63, above every code
            r = make_int2((int)lo, (int)(hi - lo));
        }
    }
    return r;
}

// 32 threads per segment: thread t < 27 finds the range of stencil cell t; the population of the stencil decides
// whether the segment is fine as it is (bad[0] / bad[1]: segments / points of the segments over the cap)
__global__ __launch_bounds__(256) void k_tree_stencil(const u64* __restrict__ codes, const int* __restrict__ bucket, const int4* __restrict__ seg_hdr,
                                                      const int2* __restrict__ seg_range, int64_t n_segs, int cap,
                                                      int2* __restrict__ runs, int* __restrict__ seg_pop, int* __restrict__ bad) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t seg = g >> 5;
    const int t = (int)(g & 31);
    if (seg >= n_segs) return;                        // (whole 32-thread groups leave together)
    const int4 hd = seg_hdr[seg];
    const int2 r = stencil_range(codes, bucket, hd.x, hd.y, hd.z, hd.w, t);
    if (t < 27) runs[seg * 27 + t] = r;
    int pop = r.y;
    for (int o = 16; o > 0; o >>= 1) pop += __shfl_xor(pop, o, 32);
    if (t == 0) {
        if ((seg & 63) == 0) atomicAdd(&bad[2 + min(hd.x, 21)], seg_range[seg].y);   // one segment in 64: how many octaves of density the cloud spans
        seg_pop[seg] = pop;
        if (pop > cap && hd.x > 0) {
            atomicAdd(&bad[0], 1);                     // (device words: an atomic on pinned host memory is a PCIe round trip)
            atomicAdd(&bad[1], seg_range[seg].y);
        }
    }
}

// Segments whose stencil holds more than `cap` points (a sparse cell next to dense ones: the own-cell count that chose
// the level says nothing about the neighbours) are split, octant by octant, until every piece fits: one wave per such
// segment walks its own run of points depth first.  Pieces become new segments and items at the end of the arrays,
// the segment's old items die (segment -1).
__global__ __launch_bounds__(256) void k_tree_refine(const u64* __restrict__ codes, const int* __restrict__ bucket, const u64* __restrict__ sums,
                                                     int64_t n_segs, int cap, int items_q, int4* __restrict__ seg_hdr,
                                                     const int2* __restrict__ seg_range, const int* __restrict__ seg_pop,
                                                     int2* __restrict__ runs, int2* __restrict__ items, unsigned char* __restrict__ lvl,
                                                     int* __restrict__ counts) {
    __shared__ int4 s_stack[4][160];
    __shared__ int2 s_runs[4][8][27];
    __shared__ int s_pop[4][8];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int4* stack = s_stack[w];
    for (int64_t seg = (int64_t)blockIdx.x * 4 + w; seg < n_segs; seg += (int64_t)gridDim.x * 4) {
        const int4 hd = seg_hdr[seg];
        if (seg_pop[seg] <= cap || hd.x <= 0) continue;
        const int2 own = seg_range[seg];
        const int first_item = (int)(sums[own.x] & 0xffffffffull) - 1, old_items = (own.y + items_q - 1) / items_q;
        for (int i = lane; i < old_items; i += 64) items[first_item + i].y = -1;
        int sp = 0;
        if (lane == 0) stack[0] = make_int4(hd.x, own.x, own.y, 0);
        sp = 1;
        while (sp > 0) {
            tree_lds_sync();
            const int4 e = stack[--sp];                       // {level, first position, points}: a run inside ONE cell of that level
            const int cl = e.x - 1;                           // split it by the octants of the level below
            const u64 base = (codes[e.y] >> (3 * e.x)) << 3;
            int bound = e.y + e.z;
            if (lane == 0) bound = e.y;
            else if (lane < 8) bound = e.y + (int)pct_code_lower_bound(codes + e.y, e.z, (base + (u64)lane) << (3 * cl));
            if (lane < 8) s_pop[w][lane] = 0;
            tree_lds_sync();
            // the 8 x 27 stencil ranges of the octants, 64 at a time (the binary searches are what this kernel waits for)
            for (int idx = lane; idx < 8 * 27; idx += 64) {
                const int c = idx / 27, t = idx - c * 27;
                const int cs = __shfl(bound, c), ce = __shfl(bound, c + 1);
                int2 r = make_int2(0, 0);
                if (ce > cs) {
                    const u64 code = codes[cs];
                    r = stencil_range(codes, bucket, cl, (int)(compact3(code) >> cl), (int)(compact3(code >> 1) >> cl), (int)(compact3(code >> 2) >> cl), t);
                    if (r.y > 0) atomicAdd(&s_pop[w][c], r.y);
                }
                s_runs[w][c][t] = r;
            }
            tree_lds_sync();
            for (int c = 0; c < 8; ++c) {
                const int cs = __builtin_amdgcn_readlane(bound, c), ce = __builtin_amdgcn_readlane(bound, c + 1);
                if (ce <= cs) continue;
                const int pop = s_pop[w][c];
                if (pop > cap && cl > 0 && sp < 159) {
                    if (lane == 0) stack[sp] = make_int4(cl, cs, ce - cs, 0);
                    ++sp;
                    continue;
                }
                const u64 code = codes[cs];
                const int cx = (int)(compact3(code) >> cl), cy = (int)(compact3(code >> 1) >> cl), cz = (int)(compact3(code >> 2) >> cl);
                const int len = ce - cs, n_new = (len + items_q - 1) / items_q;
                int seg_new = 0, item_new = 0;
                if (lane == 0) { seg_new = atomicAdd(&counts[1], 1); item_new = atomicAdd(&counts[0], n_new); }
                seg_new = __builtin_amdgcn_readfirstlane(seg_new);
                item_new = __builtin_amdgcn_readfirstlane(item_new);
                if (lane == 0) seg_hdr[seg_new] = make_int4(cl, cx, cy, cz);
                if (lane < 27) runs[(int64_t)seg_new * 27 + lane] = s_runs[w][c][lane];
                for (int i = lane; i < n_new; i += 64) items[item_new + i] = make_item(cs + i * items_q, min(items_q, len - i * items_q), seg_new);
                for (int i = lane; i < len; i += 64) lvl[cs + i] = (unsigned char)cl;
            }
        }
    }
}

struct MaxInt {
    __device__ __host__ int operator()(int a, int b) const { return a > b ? a : b; }
};

}  // namespace

// Morton order, levels, segments, items and stencil ranges of the whole cloud (public rows [0, n) all owned).
// Leaves: sorted4 (Morton order), owned_pos (identity: table row = Morton position), occ (items + sentinel),
// tree_seg / tree_runs, grid = the FINEST level's grid, n_items.
int pct_build_tree(pct_ctx* ctx, int32_t k, double eps, bool* usable) {
    const int64_t n = ctx->n;
    float bbox[6];
    *usable = false;
    PCT_TRY(pct_pack_points(ctx, bbox));                 // pts4 (public order) + bounding box; refuses non-finite input
    if (ctx->n_grid != n) return pct_fail(ctx, PCT_ERR_INVALID, "the tree sweep needs the whole cloud packed");
    if (ctx->tree_check_bbox) {          // PCT_KNN_AUTO came here on a remembered verdict: is this still that kind of cloud?
        bool same = true;
        for (int a = 0; a < 3; ++a) {
            const float tol = 0.02f * (ctx->auto_tree_bbox[3 + a] - ctx->auto_tree_bbox[a]) + 1e-30f;
            same = same && fabsf(bbox[a] - ctx->auto_tree_bbox[a]) <= tol && fabsf(bbox[3 + a] - ctx->auto_tree_bbox[3 + a]) <= tol;
        }
        if (!same) { ctx->tree_hint_mismatch = true; return PCT_OK; }
    }
    for (int a = 0; a < 6; ++a) ctx->tree_bbox[a] = bbox[a];
    double ext = 0;
    for (int a = 0; a < 3; ++a) ext = fmax(ext, (double)bbox[3 + a] - bbox[a]);
    if (!(ext > 0)) ext = 1.0;
    const double root = ext * (1.0 + 0x1p-18);           // every coordinate strictly inside the cube
    // The sweep pre-selects in float32: squared differences of up to three cell edges of ANY level, and the squared
    // eps radius, must neither overflow nor underflow there (the uniform list has a float64-keyed variant for such
    // clouds; here they are left to it).
    const double fine2 = ldexp(root, -kTreeBits) * ldexp(root, -kTreeBits);
    if (!(fine2 > 1e-30) || !(root * root < 1e30) || (eps > 0 && !(eps * eps > 1e-36))) return PCT_OK;
    *usable = true;
    pct_grid g = {};
    g.ox = bbox[0]; g.oy = bbox[1]; g.oz = bbox[2];
    g.cell = ldexp(root, -kTreeBits);
    g.inv_cell = 1.0 / g.cell;
    g.nx = g.ny = g.nz = 1 << kTreeBits;
    g.ncell = 0;
    for (int a = 0; a < 3; ++a) { g.lim_lo[a] = -INFINITY; g.lim_hi[a] = INFINITY; }
    int items_q = 12;
    if (const char* e = pct_getenv("PCT_ITEMS_Q")) { const int v = atoi(e); if (v >= 1 && v <= 64) items_q = v; }
    ctx->items_q = items_q;
    double f_min = 0.45;
    if (const char* e = pct_getenv("PCT_TREE_NMIN")) { const double v = atof(e); if (v > 0.05 && v < 8) f_min = v; }   // tuning aid
    int n_min = (int)lrint(f_min * (k + 1));
    n_min = n_min < 2 ? 2 : n_min;
    int max_level = kTreeBits;
    if (eps > 0) {                                       // one ring of cells of edge >= eps covers the eps ball
        const double need = log2(eps * 1.000001 / g.cell);
        max_level = need <= 0 ? 0 : need >= kTreeBits ? kTreeBits : (int)ceil(need);
    }

    const size_t nn = (size_t)n;
    PCT_TRY(pct_reserve(ctx, &ctx->tree_codes, 2 * nn * sizeof(u64)));
    PCT_TRY(pct_reserve(ctx, &ctx->tree_vals, 2 * nn * sizeof(unsigned)));
    PCT_TRY(pct_reserve(ctx, &ctx->tree_lvl, nn));
    PCT_TRY(pct_reserve(ctx, &ctx->tree_bucket, ((size_t)(1u << kTreeBucketBits) + 2) * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->tree_head, 2 * nn * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->tree_marks, 2 * nn * sizeof(u64)));
    PCT_TRY(pct_reserve(ctx, &ctx->sorted4, nn * sizeof(float4)));
    PCT_TRY(pct_reserve(ctx, &ctx->owned_pos, nn * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->row_of, (nn + 1) * sizeof(int)));
    if (ctx->has_f64) PCT_TRY(pct_reserve(ctx, &ctx->sorted4d, nn * sizeof(double4)));
    PCT_TRY(pct_reserve(ctx, &ctx->occ, (2 * nn + 16) * sizeof(int2)));          // items; refinement appends (<= one per point)
    PCT_TRY(pct_reserve(ctx, &ctx->tree_range, (nn + 1) * (sizeof(int2) + sizeof(int)) + 128));
    u64* codes_in = (u64*)ctx->tree_codes.p;
    u64* codes = codes_in + nn;
    unsigned* vals_in = (unsigned*)ctx->tree_vals.p;
    unsigned* vals = vals_in + nn;
    int* head = (int*)ctx->tree_head.p;
    int* seg_start = head + nn;
    u64* marks = (u64*)ctx->tree_marks.p;
    u64* sums = marks + nn;
    const dim3 grid1((unsigned)((n + 255) / 256)), block(256);
    const bool debug = pct_getenv("PCT_TREE_DEBUG") != nullptr;
    double t_mark[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const auto tick = [&](int i) {
        if (!debug) return;
        (void)hipStreamSynchronize(ctx->stream);
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        t_mark[i] = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    tick(0);

    PCT_LAUNCH(k_tree_codes, grid1, block, 0, ctx->stream, (const float4*)ctx->pts4.p, n, g.ox, g.oy, g.oz, g.inv_cell, codes_in, vals_in);
    PCT_HIP(ctx, hipGetLastError());
    size_t tmp_sort = 0, tmp_max = 0, tmp_sum = 0;
    PCT_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_sort, codes_in, codes, vals_in, vals, nn, 0, 3 * kTreeBits, ctx->stream));
    PCT_HIP(ctx, rocprim::inclusive_scan(nullptr, tmp_max, head, seg_start, nn, MaxInt(), ctx->stream));
    PCT_HIP(ctx, rocprim::inclusive_scan(nullptr, tmp_sum, marks, sums, nn, rocprim::plus<u64>(), ctx->stream));
    size_t tmp_bytes = tmp_sort > tmp_max ? tmp_sort : tmp_max;
    tmp_bytes = tmp_bytes > tmp_sum ? tmp_bytes : tmp_sum;
    PCT_TRY(pct_reserve(ctx, &ctx->tree_tmp, tmp_bytes + 256));
    PCT_HIP(ctx, rocprim::radix_sort_pairs(ctx->tree_tmp.p, tmp_sort, codes_in, codes, vals_in, vals, nn, 0, 3 * kTreeBits, ctx->stream));
    tick(1);
    PCT_LAUNCH(k_tree_gather, grid1, block, 0, ctx->stream, (const float4*)ctx->pts4.p, (const unsigned*)vals, n,
                       (float4*)ctx->sorted4.p, (int*)ctx->owned_pos.p, (int*)ctx->row_of.p,
                       ctx->has_f64 ? (const double4*)ctx->pts4d.p : nullptr, ctx->has_f64 ? (double4*)ctx->sorted4d.p : nullptr);
    PCT_LAUNCH(k_tree_buckets, dim3(((1u << kTreeBucketBits) + 2 + 255) / 256), block, 0, ctx->stream, (const u64*)codes, n, (int*)ctx->tree_bucket.p);
    PCT_LAUNCH(k_tree_level, grid1, block, 0, ctx->stream, (const u64*)codes, n, n_min, max_level, (unsigned char*)ctx->tree_lvl.p);
    PCT_LAUNCH(k_tree_heads, grid1, block, 0, ctx->stream, (const u64*)codes, (const unsigned char*)ctx->tree_lvl.p, n, head);
    PCT_HIP(ctx, hipGetLastError());
    PCT_HIP(ctx, rocprim::inclusive_scan(ctx->tree_tmp.p, tmp_max, head, seg_start, nn, MaxInt(), ctx->stream));
    PCT_LAUNCH(k_tree_marks, grid1, block, 0, ctx->stream, (const int*)seg_start, n, items_q, marks);
    PCT_HIP(ctx, hipGetLastError());
    PCT_HIP(ctx, rocprim::inclusive_scan(ctx->tree_tmp.p, tmp_sum, marks, sums, nn, rocprim::plus<u64>(), ctx->stream));
    // (segments <= items <= n; refinement appends at most one segment per point of a segment it splits)
    PCT_TRY(pct_reserve(ctx, &ctx->tree_seg, (2 * nn + 2) * sizeof(int4)));
    int2* seg_range = (int2*)ctx->tree_range.p;
    int* seg_pop = (int*)(seg_range + nn + 1);
    int* counts = seg_pop + nn + 1;                       // device: {items, segments, segments over the cap, their points, points per level [22]}
    long long* totals = (long long*)(ctx->pin + 2176);    // host: {items, segments}
    PCT_LAUNCH(k_tree_items, grid1, block, 0, ctx->stream, (const u64*)codes, (const unsigned char*)ctx->tree_lvl.p,
                       (const int*)seg_start, (const u64*)sums, n, items_q, (int2*)ctx->occ.p, (int4*)ctx->tree_seg.p, seg_range, counts, totals);
    PCT_HIP(ctx, hipGetLastError());
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    tick(2);
    int64_t n_items = totals[0], n_segs = totals[1];
    if (n_items <= 0 || n_items > n || n_segs <= 0 || n_segs > n_items)
        return pct_fail(ctx, PCT_ERR_INVALID, "tree build: %lld items in %lld segments for %lld points", (long long)n_items, (long long)n_segs, (long long)n);
    // staging capacity of the sweep that will run (k_knn_fast<..., TREE>)
#ifndef PCT_TREE_CAP
#define PCT_TREE_CAP 768
#endif
#ifndef PCT_TREE_CAP2
#define PCT_TREE_CAP2 1024
#endif
    int cap = k + 1 <= pct_fast_r1_max() ? PCT_TREE_CAP : PCT_TREE_CAP2;
    if (const char* e = pct_getenv("PCT_TREE_SPLIT")) { const int v = atoi(e); if (v >= 64 && v <= cap) cap = v; }       // tuning aid
    size_t room = (size_t)n_segs + nn / 8 + 64;           // segments the range table has room for
    PCT_TRY(pct_reserve(ctx, &ctx->tree_runs, room * 27 * sizeof(int2)));
    PCT_LAUNCH(k_tree_stencil, dim3((unsigned)((n_segs * 32 + 255) / 256)), block, 0, ctx->stream, (const u64*)codes, (const int*)ctx->tree_bucket.p,
                       (const int4*)ctx->tree_seg.p, (const int2*)seg_range, n_segs, cap, (int2*)ctx->tree_runs.p, seg_pop, counts + 2);
    PCT_HIP(ctx, hipGetLastError());
    PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 2208, counts, 26 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    tick(3);
    {   // levels between the 5th and the 95th percentile of the sampled points
        const int* hist = (const int*)(ctx->pin + 2208) + 4;
        long long tot = 0, acc = 0;
        for (int l = 0; l < 22; ++l) tot += hist[l];
        int lo = 0, hi = 21;
        for (int l = 0; l < 22; ++l) { acc += hist[l]; if (acc * 20 >= tot) { lo = l; break; } }
        acc = 0;
        for (int l = 21; l >= 0; --l) { acc += hist[l]; if (acc * 20 >= tot) { hi = l; break; } }
        ctx->tree_level_spread = tot > 0 && hi > lo ? hi - lo : 0;
        long long best2 = 0;                   // share of the points on the two most populated ADJACENT levels
        for (int l = 0; l + 1 < 22; ++l) best2 = best2 > (long long)hist[l] + hist[l + 1] ? best2 : (long long)hist[l] + hist[l + 1];
        ctx->tree_two_level_share = tot > 0 ? (double)best2 / (double)tot : 1.0;
    }
    const int64_t bad_segs = ((const int*)(ctx->pin + 2208))[2], bad_pts = ((const int*)(ctx->pin + 2208))[3];
    if (bad_segs > 0 && !pct_getenv("PCT_TREE_NO_REFINE")) {
        if ((size_t)(n_segs + bad_pts) > room) {          // rare: most of the cloud is being split -- a larger table, contents kept
            room = (size_t)(n_segs + bad_pts) + 64;
            pct_buf bigger;
            PCT_TRY(pct_reserve(ctx, &bigger, room * 27 * sizeof(int2)));
            PCT_HIP(ctx, hipMemcpyAsync(bigger.p, ctx->tree_runs.p, (size_t)n_segs * 27 * sizeof(int2), hipMemcpyDeviceToDevice, ctx->stream));
            PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            pct_release(&ctx->tree_runs);
            ctx->tree_runs = bigger;
        }
        const int64_t waves = n_segs;
        const int blocks = (int)((waves + 3) / 4 < 8192 ? (waves + 3) / 4 : 8192);
        PCT_LAUNCH(k_tree_refine, dim3(blocks), block, 0, ctx->stream, (const u64*)codes, (const int*)ctx->tree_bucket.p, (const u64*)sums, n_segs, cap, items_q,
                           (int4*)ctx->tree_seg.p, (const int2*)seg_range, (const int*)seg_pop, (int2*)ctx->tree_runs.p, (int2*)ctx->occ.p,
                           (unsigned char*)ctx->tree_lvl.p, counts);
        PCT_HIP(ctx, hipGetLastError());
        PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 2208, counts, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int* fin = (const int*)(ctx->pin + 2208);
        if (fin[0] < n_items || fin[0] > n_items + bad_pts || fin[1] < n_segs || (size_t)fin[1] > room)
            return pct_fail(ctx, PCT_ERR_INVALID, "tree refinement: %d items, %d segments (from %lld / %lld, %lld points split)", fin[0], fin[1],
                            (long long)n_items, (long long)n_segs, (long long)bad_pts);
        n_items = fin[0];
        n_segs = fin[1];
    }
    tick(4);
    if (debug)
        fprintf(stderr, "[tree] %lld points: %lld items in %lld segments (levels spread %d, %.0f %% of the points on two adjacent levels); %lld segments over %d stencil points (%lld points) split | codes+sort %.3f, "
                "gather..items %.3f, stencil %.3f, refine %.3f ms\n", (long long)n, (long long)n_items, (long long)n_segs, ctx->tree_level_spread, 100.0 * ctx->tree_two_level_share, (long long)bad_segs, cap,
                (long long)bad_pts, t_mark[1] - t_mark[0], t_mark[2] - t_mark[1], t_mark[3] - t_mark[2], t_mark[4] - t_mark[3]);

    ctx->grid = g;
    ctx->tree_bits = kTreeBits;
    ctx->n_items = n_items;
    ctx->n_occ = n_items;
    ctx->tree_segs = n_segs;
    ctx->nonempty_cells = n_segs;
    ctx->tm.grid_iters = 1;
    ctx->tm.cells = n_segs;
    ctx->tm.cell_size = g.cell;
    ctx->tm.grid_points = n;
    ctx->tm.occupancy = (double)n / (double)n_segs;
    ctx->tm.occupied_cells = n_items;
    ctx->grid_valid = false;          // not a uniform cell list: nothing that walks cell_cnt may use it
    return PCT_OK;
}
