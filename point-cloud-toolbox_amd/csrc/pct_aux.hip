// Next row N4 of the scope table: the two scan-preparation steps that share building blocks with the path.
//
// 1. Voxel-grid down-sampling, /root/reference/convert_asc_to_ply.py:20-51: voxel = floor(coordinate / voxel_size)
//    in the coordinates' dtype, keep the FIRST point (lowest input index) of every voxel, output in order of first occurrence.
//    GPU form: open-addressing hash table on the packed voxel key (64-bit atomicCAS) holding the minimum point index
//    (atomicMin), then an order-preserving compaction of the representatives.
// 2. PCA surface variation, /root/reference/utils.py:778-829: k = min(max(5, int(0.025 N)), 100) neighbours INCLUDING
//    the point itself, covariance about their mean with 1/(k-1), curvature = lambda_min / (sum lambda + 1e-10).
//    GPU form: the neighbour table of the sweep (k-1 neighbours, self excluded) + one thread per point with the
//    same cyclic Jacobi as the fit kernel, fp64 throughout.  NOTE: as written, the reference's einsum subscripts
//    (utils.py:822) build the k x k Gram matrix instead of the 3 x 3 covariance, so its result is ~0 for k > 3;
//    this kernel computes what its docstring specifies (see oracle/pct_oracle.py::surface_variation).
#include "pct_internal.h"

#include <math.h>

namespace {

constexpr unsigned long long kEmptyKey = ~0ull;

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// voxel coordinates as the reference computes them: np.floor(coordinates / voxel_size).astype(np.int32) IN THE ARRAY'S
// DTYPE -- NumPy divides a float32 array by a Python float in float32 (the scalar is rounded to float32 first), and on
// coordinates that are multiples of the voxel size the quotient's last bit decides the voxel
template <typename T>
__global__ __launch_bounds__(256) void k_voxel_keys(const T* __restrict__ xyz, int64_t n, T voxel, int* __restrict__ vox,
                                                    int* __restrict__ red) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int v[3] = {0, 0, 0};
    if (i < n) {
        for (int a = 0; a < 3; ++a) {
            v[a] = (int)floor(xyz[3 * i + a] / voxel);          // IEEE division and floor in T
            vox[3 * i + a] = v[a];
        }
    }
    // extent of the voxel coordinates: wave reduction, then one atomic per wave and component
    for (int a = 0; a < 3; ++a) {
        int lo = i < n ? v[a] : INT32_MAX, hi = i < n ? v[a] : INT32_MIN;
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&red[a], lo); atomicMax(&red[3 + a], hi); }
    }
}

__global__ __launch_bounds__(256) void k_voxel_insert(const int* __restrict__ vox, int64_t n, int ox, int oy, int oz,
                                                      unsigned long long* __restrict__ keys, int* __restrict__ vals,
                                                      unsigned long long mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = ((unsigned long long)(unsigned)(vox[3 * i] - ox) << 42) |
                                   ((unsigned long long)(unsigned)(vox[3 * i + 1] - oy) << 21) |
                                   (unsigned long long)(unsigned)(vox[3 * i + 2] - oz);
    unsigned long long slot = mix64(key) & mask;
    for (unsigned long long probe = 0; probe <= mask; ++probe) {          // bounded: the table is never full (>= 2n slots)
        const unsigned long long prev = atomicCAS(&keys[slot], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) {
            atomicMin(&vals[slot], (int)i);
            return;
        }
        slot = (slot + 1) & mask;
    }
}

// flag = 1 for the first point of every voxel; per-block counts for the compaction
__global__ __launch_bounds__(256) void k_voxel_flag(const int* __restrict__ vox, int64_t n, int ox, int oy, int oz,
                                                    const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                    unsigned long long mask, unsigned char* __restrict__ flag,
                                                    int* __restrict__ block_cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int f = 0;
    if (i < n) {
        const unsigned long long key = ((unsigned long long)(unsigned)(vox[3 * i] - ox) << 42) |
                                       ((unsigned long long)(unsigned)(vox[3 * i + 1] - oy) << 21) |
                                       (unsigned long long)(unsigned)(vox[3 * i + 2] - oz);
        unsigned long long slot = mix64(key) & mask;
        for (unsigned long long probe = 0; probe <= mask; ++probe) {
            const unsigned long long kk = keys[slot];
            if (kk == key) { f = vals[slot] == (int)i; break; }
            if (kk == kEmptyKey) break;
            slot = (slot + 1) & mask;
        }
        flag[i] = (unsigned char)f;
    }
    const unsigned long long m = __ballot(f);
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (int)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(1024) void k_scan_int(int* __restrict__ v, int n) {   // exclusive, single block, total -> v[n]
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int x = i < n ? v[i] : 0;
        sh[threadIdx.x] = x;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += a;
            __syncthreads();
        }
        const int incl = sh[threadIdx.x], c = carry;
        if (i < n) v[i] = c + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) v[n] = carry;
}

__global__ __launch_bounds__(256) void k_voxel_compact(const unsigned char* __restrict__ flag, int64_t n, const int* __restrict__ block_off,
                                                       int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int f = i < n ? flag[i] : 0;
    const unsigned long long m = __ballot(f);
    __shared__ int sh[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = (int)__popcll(m);
    __syncthreads();
    int off = block_off[blockIdx.x];
    for (int j = 0; j < w; ++j) off += sh[j];
    if (f) out[off + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = i;
}

// ---- PCA surface variation ------------------------------------------------------------------------------------
#define JROT(app, aqq, apq, arp, arq)                                                   \
    do {                                                                                \
        if (apq != 0.0) {                                                               \
            const double alpha = 0.5 * (aqq - app), beta = apq;                         \
            const double t = (alpha >= 0.0 ? beta : -beta) / (fabs(alpha) + sqrt(alpha * alpha + beta * beta)); \
            const double c = rsqrt(t * t + 1.0), s = t * c;                             \
            app -= t * apq; aqq += t * apq; apq = 0.0;                                  \
            const double rp = arp, rq = arq;                                            \
            arp = c * rp - s * rq; arq = s * rp + c * rq;                               \
        }                                                                               \
    } while (0)

__global__ __launch_bounds__(64) void k_surface_variation(const float4* __restrict__ pts, const int* __restrict__ owned_pos,
                                                          int row_offset, const int* __restrict__ table, int pitch, int kn,
                                                          int64_t rows, int64_t out_base, float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const int qid = owned_pos ? owned_pos[row] : (int)row + row_offset;
    const float4 q = pts[qid];
    const int* my = table + row * pitch;
    // neighbourhood = the point itself + its kn nearest neighbours (sklearn's kneighbors(points) returns self first)
    double sx = q.x, sy = q.y, sz = q.z;
    for (int j = 0; j < kn; ++j) { const float4 p = pts[my[j]]; sx += p.x; sy += p.y; sz += p.z; }
    const double inv = 1.0 / (double)(kn + 1);
    const double mx = sx * inv, my_ = sy * inv, mz = sz * inv;
    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
    for (int j = -1; j < kn; ++j) {
        const float4 p = j < 0 ? q : pts[my[j]];
        const double x = p.x - mx, y = p.y - my_, z = p.z - mz;
        a00 = fma(x, x, a00); a01 = fma(x, y, a01); a02 = fma(x, z, a02);
        a11 = fma(y, y, a11); a12 = fma(y, z, a12); a22 = fma(z, z, a22);
    }
    const double s = 1.0 / (double)kn;                        // 1 / (k - 1), k = kn + 1 (utils.py:822)
    a00 *= s; a01 *= s; a02 *= s; a11 *= s; a12 *= s; a22 *= s;
#pragma unroll 1
    for (int sweep = 0; sweep < 8; ++sweep) {
        const double off = fabs(a01) + fabs(a02) + fabs(a12);
        if (off <= 1e-22 * (fabs(a00) + fabs(a11) + fabs(a22))) break;
        JROT(a00, a11, a01, a02, a12);
        JROT(a00, a22, a02, a01, a12);
        JROT(a11, a22, a12, a01, a02);
    }
    const double lmin = fmin(a00, fmin(a11, a22));
    out[(int64_t)__float_as_int(q.w) - out_base] = (float)(lmin / (((a00 + a11) + a22) + 1e-10));   // utils.py:827-828
}

}  // namespace

int pct_voxel_downsample_device(pct_ctx* ctx, const void* d_xyz, bool f64, int64_t n, double voxel, int64_t* d_out, int64_t* count) {
    PCT_TRY(pct_reserve(ctx, &ctx->red, 64));
    int init[8] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0, 0};
    PCT_HIP(ctx, hipMemcpyAsync(ctx->red.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    PCT_TRY(pct_reserve(ctx, &ctx->cell_of, (size_t)n * 3 * sizeof(int)));
    const int blocks = (int)((n + 255) / 256);
    if (f64)
        PCT_LAUNCH(k_voxel_keys<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)d_xyz, n, voxel, (int*)ctx->cell_of.p, (int*)ctx->red.p);
    else
        PCT_LAUNCH(k_voxel_keys<float>, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)d_xyz, n, (float)voxel, (int*)ctx->cell_of.p, (int*)ctx->red.p);
    int mm[8];
    PCT_HIP(ctx, hipMemcpyAsync(mm, ctx->red.p, sizeof(mm), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int a = 0; a < 3; ++a)
        if ((long long)mm[3 + a] - mm[a] >= (1 << 21))
            return pct_fail(ctx, PCT_ERR_INVALID, "voxel grid spans more than 2^21 voxels along an axis");
    unsigned long long slots = 1024;
    while (slots < (unsigned long long)n * 2) slots <<= 1;
    PCT_TRY(pct_reserve(ctx, &ctx->cell_cnt, (size_t)slots * sizeof(unsigned long long)));
    PCT_TRY(pct_reserve(ctx, &ctx->cell_own, (size_t)slots * sizeof(int)));
    PCT_TRY(pct_reserve(ctx, &ctx->cell_fill, (size_t)n));
    PCT_TRY(pct_reserve(ctx, &ctx->scan_tmp, (size_t)(blocks + 1) * sizeof(int)));
    PCT_HIP(ctx, hipMemsetAsync(ctx->cell_cnt.p, 0xFF, (size_t)slots * sizeof(unsigned long long), ctx->stream));
    PCT_HIP(ctx, hipMemsetAsync(ctx->cell_own.p, 0x7F, (size_t)slots * sizeof(int), ctx->stream));     // 0x7F7F7F7F > any index
    PCT_LAUNCH(k_voxel_insert, dim3(blocks), dim3(256), 0, ctx->stream, (const int*)ctx->cell_of.p, n, mm[0], mm[1], mm[2],
                       (unsigned long long*)ctx->cell_cnt.p, (int*)ctx->cell_own.p, slots - 1);
    PCT_LAUNCH(k_voxel_flag, dim3(blocks), dim3(256), 0, ctx->stream, (const int*)ctx->cell_of.p, n, mm[0], mm[1], mm[2],
                       (const unsigned long long*)ctx->cell_cnt.p, (const int*)ctx->cell_own.p, slots - 1,
                       (unsigned char*)ctx->cell_fill.p, (int*)ctx->scan_tmp.p);
    PCT_LAUNCH(k_scan_int, dim3(1), dim3(1024), 0, ctx->stream, (int*)ctx->scan_tmp.p, blocks);
    PCT_LAUNCH(k_voxel_compact, dim3(blocks), dim3(256), 0, ctx->stream, (const unsigned char*)ctx->cell_fill.p, n,
                       (const int*)ctx->scan_tmp.p, d_out);
    PCT_HIP(ctx, hipGetLastError());
    int total = 0;
    PCT_HIP(ctx, hipMemcpyAsync(&total, (int*)ctx->scan_tmp.p + blocks, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PCT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = total;
    ctx->grid_valid = ctx->knn_valid = false;      // scratch buffers of the cell list were reused
    return PCT_OK;
}

int pct_launch_surface_variation(pct_ctx* ctx, float* d_out) {
    const bool sorted = ctx->knn_sorted_space;
    const int64_t rows = ctx->q_end - ctx->q_begin;
    PCT_LAUNCH(k_surface_variation, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, ctx->stream,
                       (const float4*)(sorted ? ctx->sorted4.p : ctx->pts4.p), sorted ? (const int*)ctx->owned_pos.p : nullptr,
                       (int)ctx->q_begin, (const int*)ctx->nbr_pos.p, ctx->nbr_pitch, ctx->k, rows, ctx->q_begin, d_out);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
