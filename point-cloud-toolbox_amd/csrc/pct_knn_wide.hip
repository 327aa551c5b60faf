// cKDTree.query(point, k + 1) takes any k (pointCloudToolbox.py:83).  The fast sweeps sort lists of one or two registers
// per lane (k <= 127); rows of 128 .. PCT_K_MAX neighbours go through the wave-per-query sweeps with a running list of
// 64 R elements, R = 4 | 8 -- the same kernels as R = 1 | 2 (pct_knn_sweep.h), instantiated here so that the hot
// kernels' translation unit does not carry their compile time (a 512-wide bitonic network with exact tie-breaks).
#include "pct_knn_sweep.h"

int pct_launch_knn_exact_wide(pct_ctx* ctx, const KnnArgs& a, int blocks, const int* list, const int* list_count) {
    const dim3 block(64 * kWavesPerBlock);
    if (a.k + 1 <= 256)
        PCT_LAUNCH(k_knn_exact<4>, dim3(blocks), block, 0, ctx->stream, a, list, list_count);
    else
        PCT_LAUNCH(k_knn_exact<8>, dim3(blocks), block, 0, ctx->stream, a, list, list_count);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_launch_knn_brute_wide(pct_ctx* ctx, const KnnArgs& a, int blocks) {
    const dim3 block(64 * kWavesPerBlock);
    if (a.k + 1 <= 256)
        PCT_LAUNCH(k_knn_brute<4>, dim3(blocks), block, 0, ctx->stream, a);
    else
        PCT_LAUNCH(k_knn_brute<8>, dim3(blocks), block, 0, ctx->stream, a);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
