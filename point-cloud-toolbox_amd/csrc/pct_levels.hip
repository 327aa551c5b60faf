// Density-adaptive neighbour sweep: a chain of cell lists instead of one.
//
// One cell size serves one density: where the cloud is much sparser than the cells were sized for, the 27-cell
// stencil does not hold k+1 points and the query falls to the exact sweep's ring-by-ring widening; where it is
// much denser the stencil overflows the LDS staging area.  Both are correct but an order of magnitude slower
// (tools/density_probe.py: two densities 10:1 -> 3x, a 1/r scan -> 10x).  Here every query a pass could not
// answer records WHY (cells too small / too large for it) and from that the cell edge it wants: a first estimate
// from the population of its cell, then a bisection between the largest edge known too small and the smallest
// known too large.  The next pass takes the most populated one-octave band of wanted edges, sizes its cells for
// it, owns exactly those queries (every point is still binned as a candidate; the cells only span the owned
// points' bounding box) and sweeps them.  Answered rows are merged into a public-space table (row = public index
// - q_begin, entries = public indices -- the representation of the exhaustive sweep, so every consumer already
// understands it).  What is left after the passes goes to the exact sweep.
#include "pct_internal.h"

#include <math.h>
#include <stdlib.h>
#include <time.h>

namespace {

#ifndef PCT_LEVEL_STEP_MAX
#define PCT_LEVEL_STEP_MAX 3.f
#endif
constexpr float kStepMax = PCT_LEVEL_STEP_MAX;   // largest jump (octaves of cell edge) taken on the population estimate alone
constexpr float kWantExact = 1.0e30f;          // wanted-edge value of a query only the exact sweep can answer
constexpr int kBins = 160;                      // histogram of wanted log2 edges, quarter octaves
constexpr float kBinLo = -20.f;                 // relative to the first pass's log2 edge

__device__ __forceinline__ int pub_of(const float4* __restrict__ sorted4, int pos) { return __float_as_int(sorted4[pos].w); }

__global__ __launch_bounds__(256) void k_init_want(float* __restrict__ want, float2* __restrict__ bracket, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    want[i] = __int_as_float(0x7fc00000);                  // NaN: not pending
    bracket[i] = make_float2(-INFINITY, INFINITY);
}

// answered rows of the pass (sorted space) -> public-space table; they stop being pending
__global__ __launch_bounds__(256) void k_merge_rows(const float4* __restrict__ sorted4, const int* __restrict__ owned_pos, int q_begin,
                                                    const int* __restrict__ nbr_pos, const float* __restrict__ nbr_dist,
                                                    const int* __restrict__ nbr_cnt, const int* __restrict__ row_done,
                                                    int64_t n_rows, int k, int pitch, int* __restrict__ pub_pos,
                                                    float* __restrict__ pub_dist, int* __restrict__ pub_cnt,
                                                    float* __restrict__ want) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / k;
    const int j = (int)(t - row * k);
    if (row >= n_rows || !row_done[row]) return;
    const int pub = pub_of(sorted4, owned_pos[row]);
    const int64_t out_row = (int64_t)pub - q_begin;
    const int pos = nbr_pos[row * pitch + j];
    pub_pos[out_row * pitch + j] = pos < 0 ? -1 : pub_of(sorted4, pos);
    pub_dist[out_row * pitch + j] = nbr_dist[row * pitch + j];
    if (j == 0) {
        if (pub_cnt) pub_cnt[out_row] = nbr_cnt ? nbr_cnt[row] : k;
        want[pub] = __int_as_float(0x7fc00000);
    }
}

// rows the pass left unanswered: update the bracket of the query's wanted edge and the wanted edge itself
__global__ __launch_bounds__(256) void k_classify(const int* __restrict__ row_done, const int* __restrict__ redo_m, int64_t n_rows,
                                                  const float4* __restrict__ sorted4, const int* __restrict__ owned_pos,
                                                  float log_edge, float target,
                                                  float* __restrict__ want, float2* __restrict__ bracket) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * 256) {
        if (row_done[i]) continue;
        const unsigned e = (unsigned)redo_m[i];      // the row's own slot: why << 29 | stencil population
        const int why = (int)(e >> 29);
        const int pub = pub_of(sorted4, owned_pos[i]);
        const float pop = (float)(e & 0x1FFFFFFFu);  // candidates the 27-cell stencil of the query's item held
        float2 b = bracket[pub];
        float w = kWantExact;
        // a failure that tells nothing new (the pass edge lay outside the bracket already known) ends the search
        const bool news = why == 1 ? log_edge > b.x : why == 2 ? log_edge < b.y : false;
        if (why == 1) b.x = fmaxf(b.x, log_edge);          // these cells were too small for it
        else if (why == 2) b.y = fminf(b.y, log_edge);     // ... too large
        if (news) {
            if (b.x > -INFINITY && b.y < INFINITY) {
                // bisection (log scale); a pass serves a one-octave band, so a bracket much narrower than that cannot
                // be resolved further (volumes: the workable window between "too few" and "overflow" is ~0.2 octaves)
                w = b.y - b.x < 0.6f ? kWantExact : 0.5f * (b.x + b.y);
            } else {
                // population ~ edge^2 on a surface, ~ edge^3 in a volume: the square root over-shoots in a volume,
                // the bracket then takes over
                const float step = 0.5f * log2f(fmaxf(target / fmaxf(pop, 0.5f), 1e-6f));
                w = why == 1 ? log_edge + fminf(fmaxf(step, 0.75f), kStepMax) : log_edge + fmaxf(fminf(step, -0.75f), -kStepMax);
            }
        }
        bracket[pub] = b;
        want[pub] = w;
    }
}

struct BandStats {
    unsigned hist[kBins];          // pending queries per quarter octave of wanted edge
    unsigned exact;                // pending queries only the exact sweep can answer
    unsigned pad;
};

__global__ __launch_bounds__(256) void k_band_hist(const float* __restrict__ want, int64_t begin, int64_t end, float log_edge0,
                                                   BandStats* __restrict__ out) {
    __shared__ unsigned sh[kBins + 1];
    for (int i = threadIdx.x; i <= kBins; i += 256) sh[i] = 0;
    __syncthreads();
    for (int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x; i < end; i += (int64_t)gridDim.x * 256) {
        const float w = want[i];
        if (w == w) {
            if (w >= 0.5f * kWantExact) atomicAdd(&sh[kBins], 1u);
            else {
                // the bin whose float bounds -- the very expressions the host turns a window of bins into [lo, hi)
                // with -- hold w: floor() of the scaled offset can land one bin off at a boundary, and a window
                // that the histogram says holds 2313 queries then owned none (box [inf, -inf], a launch of no blocks)
                int bin = (int)floorf((w - log_edge0 - kBinLo) * 4.f);
                bin = bin < 0 ? 0 : bin >= kBins ? kBins - 1 : bin;
                while (bin > 0 && w < log_edge0 + kBinLo + 0.25f * bin) --bin;
                while (bin < kBins - 1 && w >= log_edge0 + kBinLo + 0.25f * (bin + 1)) ++bin;
                atomicAdd(&sh[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins; i += 256)
        if (sh[i]) atomicAdd(&out->hist[i], sh[i]);
    if (threadIdx.x == 0 && sh[kBins]) atomicAdd(&out->exact, sh[kBins]);
}

__device__ __forceinline__ int float_order_i(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

// bounding box (ordered ints) and count of the points whose wanted edge lies in [lo, hi)
__global__ __launch_bounds__(256) void k_band_box(const float* __restrict__ want, const float* __restrict__ xyz, int64_t begin,
                                                  int64_t end, float lo, float hi, int* __restrict__ box, unsigned* __restrict__ count) {
    __shared__ int s_box[4][6];
    __shared__ unsigned s_cnt[4];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned cnt = 0;
    for (int64_t i = begin + (int64_t)blockIdx.x * 256 + threadIdx.x; i < end; i += (int64_t)gridDim.x * 256) {
        const float w = want[i];
        if (w >= lo && w < hi) {
            ++cnt;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = xyz[3 * i + a];
                mn[a] = fminf(mn[a], v);
                mx[a] = fmaxf(mx[a], v);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        for (int a = 0; a < 3; ++a) { s_box[w][a] = float_order_i(mn[a]); s_box[w][3 + a] = float_order_i(mx[a]); }
        s_cnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        int lo_i = s_box[0][a], hi_i = s_box[0][3 + a];
        for (int w = 1; w < 4; ++w) { lo_i = min(lo_i, s_box[w][a]); hi_i = max(hi_i, s_box[w][3 + a]); }
        atomicMin(&box[a], lo_i);
        atomicMax(&box[3 + a], hi_i);
    }
    if (threadIdx.x == 3) atomicAdd(count, s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
}

float order_to_float(int i) {
    const int j = i >= 0 ? i : i ^ 0x7fffffff;
    float f;
    memcpy(&f, &j, 4);
    return f;
}

template <class T>
void swap_buf(T& a, T& b) { T t = a; a = b; b = t; }

}  // namespace

int pct_knn_levels(pct_ctx* ctx, int32_t k, double eps) {
    *(long long*)(ctx->pin + 2048 + 8 * ctx->fit_par) = 0;        // rows the passes' fits hand to k_fit_svd: summed over the passes of this call
    const int64_t nq = ctx->q_end - ctx->q_begin;
    if (ctx->n >= ((int64_t)1 << 29)) return pct_fail(ctx, PCT_ERR_INVALID, "the chained sweep handles clouds below 2^29 points");
    const int pitch = (k + 3) & ~3;
    const double factor = ctx->occupancy_factor > 0 ? ctx->occupancy_factor : pct_default_factor(k);
    const double target = factor * (k + 1);
    constexpr int kMaxPasses = 14;
    const int64_t enough = nq / 256 > 1024 ? nq / 256 : 1024;      // leftovers of this size go to the exact sweep
    const bool debug = pct_getenv("PCT_LEVELS_DEBUG") != nullptr;

    ctx->no_cull = true;            // the merged table and the public-space fit need every point packed
    ctx->level_mode = true;
    ctx->own_flag = nullptr;
    ctx->own_count = 0;
    ctx->level_edge = 0;
    ctx->level_box_valid = false;
    int passes = 0;

    // one pass over the owned set in place: cell list, fast sweep (or the exact one), wanted edges, merge
    const auto tick = [&]() -> double {
        (void)hipStreamSynchronize(ctx->stream);
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    const auto run_pass = [&](int64_t owned, bool exact) -> int {
        const double t0 = debug ? tick() : 0;
        PCT_TRY(pct_build_grid(ctx, k, eps));
        const double t1 = debug ? tick() : 0;
        PCT_TRY(pct_reserve(ctx, &ctx->row_done, (size_t)owned * sizeof(int)));
        PCT_HIP(ctx, hipMemsetAsync(ctx->row_done.p, 0, (size_t)owned * sizeof(int), ctx->stream));
        PCT_TRY(pct_launch_knn_grid(ctx, k, eps, exact, exact ? 0 : 1));
        if (!exact) {
            // the stencil of a well-sized pass holds about 11 cells' worth of points on a surface
            PCT_LAUNCH(k_classify, dim3(1024), dim3(256), 0, ctx->stream, (const int*)ctx->row_done.p, (const int*)ctx->redo_m.p,
                               owned, (const float4*)ctx->sorted4.p, (const int*)ctx->owned_pos.p,
                               (float)log2(ctx->grid.cell), (float)(11.0 * target),
                               (float*)ctx->flag_buf.p, (float2*)ctx->dens_buf.p);
            PCT_HIP(ctx, hipGetLastError());
        }
        if (ctx->levels_fuse_fit) PCT_TRY(pct_launch_fit_pass(ctx, owned));       // before the next pass reorders the cloud
        const double t2 = debug ? tick() : 0;
        const int64_t total = owned * k;
        PCT_LAUNCH(k_merge_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float4*)ctx->sorted4.p, (const int*)ctx->owned_pos.p, (int)ctx->q_begin,
                           (const int*)ctx->nbr_pos.p, (const float*)ctx->nbr_dist.p,
                           eps > 0 ? (const int*)ctx->nbr_cnt.p : nullptr, (const int*)ctx->row_done.p, owned, k, pitch,
                           (int*)ctx->pub_pos.p, (float*)ctx->pub_dist.p, eps > 0 ? (int*)ctx->pub_cnt.p : nullptr,
                           (float*)ctx->flag_buf.p);
        PCT_HIP(ctx, hipGetLastError());
        if (debug) {
            const double t3 = tick();
            fprintf(stderr, "[levels] pass %d: owned %lld exact %d | build %.3f ms (iters %d, %lld cells) sweep+classify %.3f merge %.3f\n", passes,
                    (long long)owned, (int)exact, t1 - t0, ctx->tm.grid_iters, (long long)ctx->grid.ncell, t2 - t1, t3 - t2);
        }
        ++passes;
        return PCT_OK;
    };

    const auto body = [&]() -> int {
        PCT_TRY(pct_reserve(ctx, &ctx->pub_pos, (size_t)nq * pitch * sizeof(int)));
        PCT_TRY(pct_reserve(ctx, &ctx->pub_dist, (size_t)nq * pitch * sizeof(float)));
        if (eps > 0) PCT_TRY(pct_reserve(ctx, &ctx->pub_cnt, (size_t)nq * sizeof(int)));
        PCT_TRY(pct_reserve(ctx, &ctx->flag_buf, (size_t)ctx->n * sizeof(float)));
        PCT_TRY(pct_reserve(ctx, &ctx->dens_buf, (size_t)ctx->n * sizeof(float2)));
        PCT_TRY(pct_reserve(ctx, &ctx->stage_d, sizeof(BandStats) + 64));
        PCT_LAUNCH(k_init_want, dim3((unsigned)((ctx->n + 255) / 256)), dim3(256), 0, ctx->stream, (float*)ctx->flag_buf.p,
                           (float2*)ctx->dens_buf.p, ctx->n);
        PCT_HIP(ctx, hipGetLastError());
        // pass 0: every owned query, cells sized as for a plain sweep
        ctx->lvl_src_valid = false;
        PCT_TRY(run_pass(nq, false));
        // the later passes bin the points in THIS pass's cell order (lanes of a wave then share their cells: the
        // histogram's atomics combine, k_hist_agg) and reuse its box
        PCT_TRY(pct_reserve(ctx, &ctx->lvl_src, (size_t)ctx->n * sizeof(float4)));
        PCT_HIP(ctx, hipMemcpyAsync(ctx->lvl_src.p, ctx->sorted4.p, (size_t)ctx->n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->lvl_src_valid = ctx->n_grid == ctx->n;
        const float log_edge0 = (float)log2(ctx->grid.cell);
        BandStats* d_st = (BandStats*)ctx->stage_d.p;
        int* d_box = (int*)((char*)ctx->stage_d.p + sizeof(BandStats));        // 6 ints + count
        BandStats st;
        int64_t pending = 0;
        for (;;) {
            PCT_HIP(ctx, hipMemsetAsync(d_st, 0, sizeof(BandStats), ctx->stream));
            PCT_LAUNCH(k_band_hist, dim3(512), dim3(256), 0, ctx->stream, (const float*)ctx->flag_buf.p, ctx->q_begin,
                               ctx->q_end, log_edge0, d_st);
            PCT_HIP(ctx, hipGetLastError());
            PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 256, d_st, sizeof(BandStats), hipMemcpyDeviceToHost, ctx->stream));
            PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            memcpy(&st, ctx->pin + 256, sizeof(BandStats));
            int64_t banded = 0;
            for (int b = 0; b < kBins; ++b) banded += st.hist[b];
            pending = banded + st.exact;
            // the one-octave window (4 bins) that holds the most queries
            int best = 0;
            int64_t best_cnt = -1;
            for (int b = 0; b + 4 <= kBins; ++b) {
                const int64_t c = (int64_t)st.hist[b] + st.hist[b + 1] + st.hist[b + 2] + st.hist[b + 3];
                if (c > best_cnt) { best_cnt = c; best = b; }
            }
            if (debug)
                fprintf(stderr, "[levels] after pass %d (edge %.5g, %lld cells): pending %lld (exact-only %u), best octave at edge %.5g holds %lld\n",
                        passes - 1, ctx->grid.cell, (long long)ctx->grid.ncell, (long long)pending, st.exact,
                        exp2(log_edge0 + kBinLo + 0.25 * (best + 2)), (long long)best_cnt);
            if (banded <= enough || passes >= kMaxPasses || best_cnt <= enough / 8) break;
            const float lo = best == 0 ? -INFINITY : log_edge0 + kBinLo + 0.25f * best;
            const float hi = best + 4 >= kBins ? 0.4f * kWantExact : log_edge0 + kBinLo + 0.25f * (best + 4);
            const int init[7] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0};
            PCT_HIP(ctx, hipMemcpyAsync(d_box, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
            PCT_LAUNCH(k_band_box, dim3(512), dim3(256), 0, ctx->stream, (const float*)ctx->flag_buf.p, ctx->xyz_view,
                               ctx->q_begin, ctx->q_end, lo, hi, d_box, (unsigned*)(d_box + 6));
            PCT_HIP(ctx, hipGetLastError());
            PCT_HIP(ctx, hipMemcpyAsync(ctx->pin + 1024, d_box, sizeof(init), hipMemcpyDeviceToHost, ctx->stream));
            PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const int* hb = (const int*)(ctx->pin + 1024);
            for (int a = 0; a < 6; ++a) ctx->level_box[a] = order_to_float(hb[a]);
            ctx->level_box_valid = true;
            ctx->own_flag = ctx->flag_buf.p;
            ctx->own_lo = lo;
            ctx->own_hi = hi;
            ctx->own_count = (int64_t)(unsigned)hb[6];
            if (ctx->own_count <= 0) break;            // (cannot happen while histogram and window agree; never build a cell list for nobody)
            ctx->level_edge = exp2((double)log_edge0 + kBinLo + 0.25 * (best + 2));        // centre of the window
            PCT_TRY(run_pass(ctx->own_count, false));
        }
        if (pending > 0) {               // the exact sweep answers whatever is left, over the whole box
            ctx->own_flag = ctx->flag_buf.p;
            ctx->own_lo = -INFINITY;
            ctx->own_hi = INFINITY;
            ctx->own_count = pending;
            ctx->level_box_valid = false;
            // cells sized for the median wanted edge of the banded leftovers (the exact sweep widens ring by ring)
            int64_t acc = 0, banded = 0;
            for (int b = 0; b < kBins; ++b) banded += st.hist[b];
            int med = kBins / 2;
            for (int b = 0; b < kBins; ++b) { acc += st.hist[b]; if (acc * 2 >= banded) { med = b; break; } }
            ctx->level_edge = banded > 0 ? exp2((double)log_edge0 + kBinLo + 0.25 * (med + 0.5)) : exp2((double)log_edge0);
            PCT_TRY(run_pass(pending, true));
        }
        return PCT_OK;
    };
    const int st = body();
    ctx->own_flag = nullptr;
    ctx->own_count = 0;
    ctx->level_edge = 0;
    ctx->level_box_valid = false;
    ctx->level_mode = false;
    ctx->lvl_src_valid = false;
    if (st != PCT_OK) return st;
    // the merged table takes the place of the pass table: public space, as after the exhaustive sweep
    swap_buf(ctx->nbr_pos, ctx->pub_pos);
    swap_buf(ctx->nbr_dist, ctx->pub_dist);
    if (eps > 0) swap_buf(ctx->nbr_cnt, ctx->pub_cnt);
    ctx->nbr_pitch = pitch;
    ctx->knn_sorted_space = false;
    ctx->tm.levels = passes;
    ctx->levels_fitted = ctx->levels_fuse_fit;
    return PCT_OK;
}
