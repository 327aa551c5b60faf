// Energy integrals over a triangle mesh: the consumer of the path's K/H
// (next row N3 of the scope table).  Restates load_mesh_compute_energies,
// /root/reference/utils.py:702-765:
//   area_t      = 0.5 * || (v1 - v0) x (v2 - v0) ||              float64      (utils.py:723-728)
//   face_K_t    = mean(K[tri]), face_H2_t = mean((H**2)[tri])    in the dtype of the curvature arrays
//                                                                 (float32 when they come from the path,
//                                                                  utils.py:741-752)
//   bending     = nansum(face_H2 * area), stretching = nansum(face_K * area), total = sum(area)
//                                                                 (utils.py:755-757)
// The reference recomputes the three sums inside its per-triangle loop (O(T^2), 43 % of its profiled run);
// only the values after the last triangle are returned, which is what is computed here: one streaming pass,
// fixed-order two-stage fp64 reduction (bitwise reproducible).
#include "pct_internal.h"

#include <math.h>

namespace {

constexpr int kMeshBlock = 256;

template <typename T>
__global__ __launch_bounds__(kMeshBlock) void k_mesh_energy(const double* __restrict__ v, const int* __restrict__ tri, int64_t n_tri,
                                                            const T* __restrict__ K, const T* __restrict__ H,
                                                            double* __restrict__ partial) {
    double bend = 0, stretch = 0, area_sum = 0;
    for (int64_t t = (int64_t)blockIdx.x * kMeshBlock + threadIdx.x; t < n_tri; t += (int64_t)gridDim.x * kMeshBlock) {
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        const double ax = v[3 * i1] - v[3 * i0], ay = v[3 * i1 + 1] - v[3 * i0 + 1], az = v[3 * i1 + 2] - v[3 * i0 + 2];
        const double bx = v[3 * i2] - v[3 * i0], by = v[3 * i2 + 1] - v[3 * i0 + 1], bz = v[3 * i2 + 2] - v[3 * i0 + 2];
        const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        const double area = 0.5 * sqrt((cx * cx + cy * cy) + cz * cz);
        const T three = (T)3;
        const T fk = ((K[i0] + K[i1]) + K[i2]) / three;                       // np.mean in the array's dtype
        const T h0 = H[i0], h1 = H[i1], h2 = H[i2];
        const T fh2 = ((h0 * h0 + h1 * h1) + h2 * h2) / three;
        const double pb = (double)fh2 * area, ps = (double)fk * area;
        bend += isnan(pb) ? 0.0 : pb;                                          // nansum
        stretch += isnan(ps) ? 0.0 : ps;
        area_sum += area;
    }
    __shared__ double sh[3][kMeshBlock / 64];
    for (int o = 32; o > 0; o >>= 1) {
        bend += __shfl_xor(bend, o);
        stretch += __shfl_xor(stretch, o);
        area_sum += __shfl_xor(area_sum, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = bend; sh[1][w] = stretch; sh[2][w] = area_sum; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int i = 0; i < kMeshBlock / 64; ++i) s += sh[threadIdx.x][i];
        partial[(int64_t)blockIdx.x * 3 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(64) void k_mesh_final(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
    if (threadIdx.x < 3) {
        double s = 0;
        for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * 3 + threadIdx.x];   // fixed order
        out[threadIdx.x] = s;
    }
}

}  // namespace

int pct_launch_mesh_energies(pct_ctx* ctx, const double* d_v, const int* d_tri, int64_t n_tri, const void* d_K, const void* d_H,
                             bool f64, double* d_partial, int nblk, double* d_out) {
    if (f64)
        PCT_LAUNCH(k_mesh_energy<double>, dim3(nblk), dim3(kMeshBlock), 0, ctx->stream, d_v, d_tri, n_tri,
                           (const double*)d_K, (const double*)d_H, d_partial);
    else
        PCT_LAUNCH(k_mesh_energy<float>, dim3(nblk), dim3(kMeshBlock), 0, ctx->stream, d_v, d_tri, n_tri,
                           (const float*)d_K, (const float*)d_H, d_partial);
    PCT_LAUNCH(k_mesh_final, dim3(1), dim3(64), 0, ctx->stream, (const double*)d_partial, nblk, d_out);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
