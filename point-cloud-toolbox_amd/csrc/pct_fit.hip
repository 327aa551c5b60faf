// Fused per-point kernel for the three per-neighbourhood staticmethods of the
// reference and the two all-points loops around them:
//   get_best_fit_plane_and_rotate            pointCloudToolbox.py:270-321
//   fit_quadratic_surface                    pointCloudToolbox.py:331-360
//   calculate_explicit_quadratic_curvatures  pointCloudToolbox.py:398-431
//   loops                                    pointCloudToolbox.py:635-647, 657-674
//
// One thread per point (tiny dense per-point algebra, no MFMA).  A 64-thread
// workgroup stages its 64 neighbour-index rows into LDS with coalesced loads
// (odd row pitch -> conflict-free column reads; letting every lane stream its
// own row from global instead was measured 2x slower), then every lane walks
// its own row twice, gathers issued eight neighbours ahead of their use:
//   pass 1  centred neighbours (native dtype, pct:641) -> fp64 sums -> 3x3
//           covariance about the neighbour mean, ddof=1 (pct:277) -> cyclic
//           Jacobi eigen-solve -> normal = eigenvector of the smallest
//           eigenvalue (== Vt[-1], pct:283) -> sign flip by far-minus-near
//           neighbour (pct:286-297) -> Rodrigues rotation to +z (pct:300-312)
//   pass 2  rotate (pct:315), round to float32 (pct:350), float32 design row
//           [a^2,b^2,ab,a,b,1] (pct:358), fp64 normal equations, Cholesky
//           solve, coefficients rounded to float32 (pct:359)
//   then    K, H, H^2 in float32 arithmetic in the reference's operation order
//           (pct:403-422).
// Compiled with -ffp-contract=off so the float32 sequences are not fused.
#include "pct_internal.h"

#include <math.h>

namespace {

constexpr int kFitBlock = 64;

struct FitArgs {
    const float4* pts;       // records {x,y,z,public index}
    const double4* ptsd;     // native fp64 coordinates (nullable), same order
    const int* table;        // (rows,k) neighbour ids into pts (-1 = missing)
    const int* cnt;          // (rows) valid neighbours per row (nullable -> k)
    const int64_t* row_query;// (rows) query id into pts per row (nullable -> row + row_offset)
    const int* row_query32;  // the same as int32 (device-built lists); takes precedence
    int64_t row_offset;
    int64_t rows;
    int k;
    int pitch;               // table row pitch in elements, multiple of 4
    int kp;                  // LDS row pitch (odd)
    int out_by_row;          // 1: outputs indexed by row; 0: by public index of the query
    int64_t out_base;        // subtracted from the public index when out_by_row == 0
    int q_begin, q_end;      // owned public range (rows outside are skipped)
    float* coefs;
    float* K;
    float* H;
    float* H2;
    double* coefs64;         // diagnostics variant (OUT64): unrounded coefficients and float64 curvatures, row-aligned
    double* K64;
    double* H64;
    int* flag_list;          // rows k_fit hands to k_fit_svd (ill-conditioned or under-determined design matrices)
    int* flag_count;
    int* flag_next;          // the count the NEXT launch will use: zeroed by this launch's first block (saves a memset launch)
    const unsigned long long* stat_src;   // fused call: the sweep's eight statistics words ...
    unsigned long long* stat_dst;         // ... copied to pinned host memory by the first block (saves a D2H copy)
    unsigned n_pts;          // records in pts: a table entry outside [0, n_pts) is never dereferenced (the row reads NaN)
    const int* row_mask;     // MASKED instantiation: only the rows with a non-zero entry are fitted (passes of the density-adaptive sweep)
    int svd_accumulate;      // the count of rows handed to k_fit_svd is ADDED to the host word (the passes of one fused call)
    int blocks_per_xcd;      // blocks are dealt to the 8 XCDs in turn: block b takes rows of block (b % 8) * blocks_per_xcd + b / 8, so
                             // that one XCD's L2 serves neighbouring rows (table rows are in cell order: their neighbours overlap)
};

// Smallest Cholesky pivot ratio d_j / g_jj (= sin^2 of the angle between design column j and the span of the columns
// before it; invariant under column scaling) below which the normal equations are not trusted.  Their error grows like
// eps / ratio: measured against LAPACK's gelsd on anisotropic lattices (tools notes in DESIGN 4.3), ratios >= 1e-9 stay
// at float32 rounding noise (<= 2.4e-7 relative in K, H), [1e-10, 1e-9) reach 1.3e-5.  1e-6 leaves three decades.
constexpr double kPivotRatioMin = 1e-6;

// One Jacobi rotation annihilating a_pq of a symmetric 3x3 (r = third index).
// With alpha = (a_qq - a_pp)/2 and beta = a_pq the tangent of the rotation is
// t = sgn(alpha) beta / (|alpha| + sqrt(alpha^2 + beta^2))  (the small root), c = 1/sqrt(1+t^2):
// one sqrt, one division and one reciprocal square root per rotation.
#define JACOBI_ROT(app, aqq, apq, arp, arq, vp0, vp1, vp2, vq0, vq1, vq2)          \
    do {                                                                            \
        if (apq != 0.0) {                                                           \
            const double alpha = 0.5 * (aqq - app);                                 \
            const double beta = apq;                                                \
            const double t = (alpha >= 0.0 ? beta : -beta) / (fabs(alpha) + sqrt(alpha * alpha + beta * beta)); \
            const double c = rsqrt(t * t + 1.0);                                    \
            const double s = t * c;                                                 \
            app -= t * apq;                                                         \
            aqq += t * apq;                                                         \
            apq = 0.0;                                                              \
            const double rp = arp, rq = arq;                                        \
            arp = c * rp - s * rq;                                                  \
            arq = s * rp + c * rq;                                                  \
            double a0 = vp0, b0 = vq0; vp0 = c * a0 - s * b0; vq0 = s * a0 + c * b0; \
            double a1 = vp1, b1 = vq1; vp1 = c * a1 - s * b1; vq1 = s * a1 + c * b1; \
            double a2 = vp2, b2 = vq2; vp2 = c * a2 - s * b2; vq2 = s * a2 + c * b2; \
        }                                                                           \
    } while (0)

// K and H of z = A a^2 + B b^2 + C ab + D a + E b + F at the origin, float32
// arithmetic in the reference's operation order (pct:403-419).
__device__ __forceinline__ void monge_curvatures(float A, float B, float C, float D, float E, float& Kg, float& Kh) {
    const float Fx = D, Fy = E;
    const float Fxx = 2.0f * A, Fyy = 2.0f * B, Fxy = C;
    const float fx2 = Fx * Fx, fy2 = Fy * Fy;
    const float wgt = (1.0f + fx2) + fy2;
    const float den_k = wgt * wgt;                                            // (...) ** 2
    const float den_h = (float)((double)wgt * sqrt((double)wgt));            // (...) ** 1.5
    Kg = (Fxx * Fyy - Fxy * Fxy) / den_k;
    Kh = (((1.0f + fx2) * Fyy - ((2.0f * Fx) * Fy) * Fxy) + (1.0f + fy2) * Fxx) / (2.0f * den_h);
}

template <bool F64>
__device__ __forceinline__ void load_centred(const FitArgs& a, int id, double qx, double qy, double qz,
                                             float qxf, float qyf, float qzf, double& x, double& y, double& z) {
    if (F64) {
        const double4 p = a.ptsd[id];
        x = p.x - qx; y = p.y - qy; z = p.z - qz;         // float64 cloud: float64 centring
    } else {
        const float4 p = a.pts[id];
        x = (double)(p.x - qxf);                           // float32 cloud: float32 centring (pct:641)
        y = (double)(p.y - qyf);
        z = (double)(p.z - qzf);
    }
}

// Tangent-plane rotation from the moment sums of one centred neighbourhood:
// covariance about the neighbour mean with ddof=1 (pct:277), cyclic Jacobi,
// normal = eigenvector of the smallest eigenvalue (== Vt[-1], pct:283), sign
// flip by the far-minus-near reference vector (pct:286-297), Rodrigues rotation
// taking the normal to +z (pct:300-312).  rot = {r00,r01,r02, r10,.., r22}.
template <bool F64>
__device__ __forceinline__ void plane_rotation(int m, double sx, double sy, double sz, double sxx, double sxy, double sxz,
                                               double syy, double syz, double szz, double fx, double fy, double fz,
                                               double lx, double ly, double lz, double (&rot)[9]) {
    const double inv_m = 1.0 / (double)m, inv_m1 = 1.0 / (double)(m - 1);
    const double mx = sx * inv_m, my_ = sy * inv_m, mz = sz * inv_m;
    double a00 = (sxx - sx * mx) * inv_m1, a01 = (sxy - sx * my_) * inv_m1, a02 = (sxz - sx * mz) * inv_m1;
    double a11 = (syy - sy * my_) * inv_m1, a12 = (syz - sy * mz) * inv_m1, a22 = (szz - sz * mz) * inv_m1;

    // ---- cyclic Jacobi on the symmetric 3x3 --------------------------------
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;   // v[row][col]
#pragma unroll 1
    for (int sweep = 0; sweep < 8; ++sweep) {
        // off-diagonal mass below 1e-22 of the trace: eigenvectors are converged far beyond fp64 round-off
        const double off = fabs(a01) + fabs(a02) + fabs(a12);
        if (off <= 1e-22 * (fabs(a00) + fabs(a11) + fabs(a22))) break;
        JACOBI_ROT(a00, a11, a01, a02, a12, v00, v10, v20, v01, v11, v21);   // (p,q)=(0,1), r=2
        JACOBI_ROT(a00, a22, a02, a01, a12, v00, v10, v20, v02, v12, v22);   // (0,2), r=1
        JACOBI_ROT(a11, a22, a12, a01, a02, v01, v11, v21, v02, v12, v22);   // (1,2), r=0
    }
    double n0, n1, n2;
    if (a00 <= a11 && a00 <= a22) { n0 = v00; n1 = v10; n2 = v20; }
    else if (a11 <= a22)          { n0 = v01; n1 = v11; n2 = v21; }
    else                          { n0 = v02; n1 = v12; n2 = v22; }

    // ---- orientation: far-minus-near neighbour (pct:286-297) ---------------
    double rx, ry, rz;
    if (F64) { rx = lx - fx; ry = ly - fy; rz = lz - fz; }
    else     { rx = (double)((float)lx - (float)fx); ry = (double)((float)ly - (float)fy); rz = (double)((float)lz - (float)fz); }
    {
        const double nn = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
        const double rn = sqrt((rx * rx + ry * ry) + rz * rz);
        const double dot = ((n0 / nn) * (rx / rn) + (n1 / nn) * (ry / rn)) + (n2 / nn) * (rz / rn);
        if (dot < 0) { n0 = -n0; n1 = -n1; n2 = -n2; }
    }
    // ---- Rodrigues rotation taking the normal to +z (pct:300-312) -----------
    double r00 = 1, r01 = 0, r02 = 0, r10 = 0, r11 = 1, r12 = 0, r20 = 0, r21 = 0, r22 = 1;
    {
        const double nn = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
        const double ax = n0 / nn, ay = n1 / nn, az = n2 / nn;
        const double v0 = ay, v1 = -ax;                    // a x (0,0,1)
        const double c = az;
        const double s = sqrt(v0 * v0 + v1 * v1);
        if (s != 0.0) {
            const double f = (1.0 - c) / (s * s);
            r00 = 1.0 + (-(v1 * v1)) * f;  r01 = (v1 * v0) * f;            r02 = v1;
            r10 = (v0 * v1) * f;           r11 = 1.0 + (-(v0 * v0)) * f;   r12 = -v0;
            r20 = -v1;                     r21 = v0;                       r22 = 1.0 + (-(v1 * v1) + -(v0 * v0)) * f;
        }
    }

    rot[0] = r00; rot[1] = r01; rot[2] = r02; rot[3] = r10; rot[4] = r11; rot[5] = r12; rot[6] = r20; rot[7] = r21; rot[8] = r22;
}

// STAGED = false: rows too long for the LDS staging area (k > 255, only reachable through caller-supplied rows:
// pct_fit_indices takes any k the reference's fit would) are walked in global memory instead.
// MASKED: a compile-time variant, so that the row test leaves the hot instantiation's register allocation alone.
template <bool F64, bool OUT64 = false, bool STAGED = true, bool MASKED = false>
__global__ __launch_bounds__(kFitBlock) void k_fit(FitArgs a) {
    extern __shared__ int s_idx[];   // 64 rows x kp
    const int lane = threadIdx.x;
    if (blockIdx.x == 0) {
        if (lane == 0 && a.flag_next) *a.flag_next = 0;
        if (a.stat_dst && lane < 8) a.stat_dst[lane] = a.stat_src[lane];
    }
    const int64_t row0 = (a.blocks_per_xcd ? (int64_t)(blockIdx.x & 7u) * a.blocks_per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x) * kFitBlock;
    if (row0 >= a.rows) return;
    const int k = a.k, kp = a.kp;

    // ---- stage 64 index rows: 16 lanes x int4 per row, 4 rows per load instruction, all loads independent
    // (rows are 16-byte aligned: pitch is a multiple of 4) -------------------------------------------------
    const int nrow = (int)min((int64_t)kFitBlock, a.rows - row0);
    if constexpr (STAGED) {
        const int sub = lane >> 4, c4 = (lane & 15) << 2;
        for (int cb = 0; cb < k; cb += 64) {
            const int c = cb + c4;
#pragma unroll 4
            for (int r0 = 0; r0 < kFitBlock; r0 += 4) {
                const int r = r0 + sub;
                if (r < nrow && c < a.pitch) {
                    const int4 v = *(const int4*)(a.table + (row0 + r) * a.pitch + c);
                    int* dst = s_idx + r * kp + c;
                    if (c + 0 < k) dst[0] = v.x;
                    if (c + 1 < k) dst[1] = v.y;
                    if (c + 2 < k) dst[2] = v.z;
                    if (c + 3 < k) dst[3] = v.w;
                }
            }
        }
    }
    __syncthreads();

    const int64_t row = row0 + lane;
    if (row >= a.rows) return;
    if constexpr (MASKED) {
        if (!a.row_mask[row]) return;
    }
    const int64_t qid = a.row_query32 ? (int64_t)a.row_query32[row] : a.row_query ? a.row_query[row] : row + a.row_offset;
    const float4 qp = a.pts[qid];
    const int pub = __float_as_int(qp.w);
    if (!a.out_by_row && (pub < a.q_begin || pub >= a.q_end)) return;
    const int64_t out = a.out_by_row ? row : (int64_t)pub - a.out_base;

    double qx = qp.x, qy = qp.y, qz = qp.z;
    if (F64) {
        const double4 qd = a.ptsd[qid];
        qx = qd.x; qy = qd.y; qz = qd.z;
    }
    const int m = a.cnt ? a.cnt[row] : k;
    const int* my = STAGED ? s_idx + lane * kp : a.table + row * a.pitch;
    const float nanf_ = __int_as_float(0x7fc00000);

    // rows this kernel does not finish go to k_fit_svd: one counter increment per wave
    const auto hand_over = [&]() {
        const unsigned long long fm = __ballot(1);                  // the lanes that are here together
        const int leader = (int)__builtin_ctzll(fm);
        int base = 0;
        if (lane == leader) base = atomicAdd(a.flag_count, (int)__popcll(fm));
        base = __shfl(base, leader);
        a.flag_list[base + (int)__popcll(fm & ((1ull << lane) - 1ull))] = (int)row;
    };
    if (m < 6) {   // under-determined quadric (eps bound, neighbour study): lstsq's minimum-norm answer, k_fit_svd
        if constexpr (OUT64) {
            for (int j = 0; j < 6; ++j) a.coefs64[out * 6 + j] = (double)nanf_;
            a.K64[out] = (double)nanf_; a.H64[out] = (double)nanf_;
        } else {
            for (int j = 0; j < 6; ++j) a.coefs[out * 6 + j] = nanf_;
            a.K[out] = nanf_; a.H[out] = nanf_; a.H2[out] = nanf_;
        }
        if (m >= 2) hand_over();       // (np.cov of fewer than two points is NaN in the reference as well)
        return;
    }

    // ---- pass 1: moments of the centred neighbourhood ---------------------
    double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    double fx = 0, fy = 0, fz = 0, lx = 0, ly = 0, lz = 0;
    // gathers are issued four neighbours ahead of their use (the row walk is latency-bound otherwise)
#define PASS1_ACC(x, y, z)                                                          \
    do {                                                                            \
        sx += x; sy += y; sz += z;                                                  \
        sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); sxz = fma(x, z, sxz);           \
        syy = fma(y, y, syy); syz = fma(y, z, syz); szz = fma(z, z, szz);           \
    } while (0)
    // A table entry that is not a record of the cloud (host-supplied rows are validated before they get here, the
    // sweep's rows are the sweep's responsibility: this is the last line of defence) is clamped, never dereferenced,
    // and the row reads NaN.
    bool bad_id = false;
    const auto checked = [&](int id) {
        bad_id |= (unsigned)id >= a.n_pts;
        return (int)min((unsigned)id, a.n_pts - 1u);
    };
    {
        int j = 0;
        for (; j + 8 <= m; j += 8) {
            double x[8], y[8], z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) load_centred<F64>(a, checked(my[j + u]), qx, qy, qz, qp.x, qp.y, qp.z, x[u], y[u], z[u]);
            if (j == 0) { fx = x[0]; fy = y[0]; fz = z[0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) PASS1_ACC(x[u], y[u], z[u]);
            lx = x[7]; ly = y[7]; lz = z[7];
        }
        for (; j < m; ++j) {
            double x, y, z;
            load_centred<F64>(a, checked(my[j]), qx, qy, qz, qp.x, qp.y, qp.z, x, y, z);
            if (j == 0) { fx = x; fy = y; fz = z; }
            PASS1_ACC(x, y, z);
            lx = x; ly = y; lz = z;
        }
    }
#undef PASS1_ACC
    if (bad_id) {
        if constexpr (OUT64) {
            for (int j = 0; j < 6; ++j) a.coefs64[out * 6 + j] = (double)nanf_;
            a.K64[out] = (double)nanf_; a.H64[out] = (double)nanf_;
        } else {
            for (int j = 0; j < 6; ++j) a.coefs[out * 6 + j] = nanf_;
            a.K[out] = nanf_; a.H[out] = nanf_; a.H2[out] = nanf_;
        }
        return;
    }
    double rot[9];
    plane_rotation<F64>(m, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, fx, fy, fz, lx, ly, lz, rot);
    const double r00 = rot[0], r01 = rot[1], r02 = rot[2], r10 = rot[3], r11 = rot[4], r12 = rot[5], r20 = rot[6], r21 = rot[7], r22 = rot[8];

    // ---- pass 2: float32 design rows -> scaled fp64 normal equations --------
    // No column scaling: scaling the Gram matrix by powers of two commutes with every rounding of an unpivoted
    // Cholesky solve, so it cannot change a single bit of the result (only the overflow/underflow range, which fp64
    // covers down to neighbourhood radii of ~1e-70).
    double g[21];
    double b[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) g[i] = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = 0;
#define PASS2_ACC(x, y, z)                                                          \
    do {                                                                            \
        const float pa = (float)((r00 * x + r01 * y) + r02 * z);                    \
        const float pb = (float)((r10 * x + r11 * y) + r12 * z);                    \
        const float pz = (float)((r20 * x + r21 * y) + r22 * z);                    \
        double c[6];                                                                \
        c[0] = (double)(pa * pa);                                                   \
        c[1] = (double)(pb * pb);                                                   \
        c[2] = (double)(pa * pb);                                                   \
        c[3] = (double)pa;                                                          \
        c[4] = (double)pb;                                                          \
        c[5] = 1.0;                                                                 \
        const double zz = (double)pz;                                               \
        int t = 0;                                                                  \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                             \
            _Pragma("unroll") for (int jj = 0; jj <= i; ++jj) { g[t] = fma(c[i], c[jj], g[t]); ++t; } \
            b[i] = fma(c[i], zz, b[i]);                                             \
        }                                                                           \
    } while (0)
    {
        int j = 0;
        for (; j + 8 <= m; j += 8) {
            double x[8], y[8], z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) load_centred<F64>(a, my[j + u], qx, qy, qz, qp.x, qp.y, qp.z, x[u], y[u], z[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) PASS2_ACC(x[u], y[u], z[u]);
        }
        for (; j < m; ++j) {
            double x, y, z;
            load_centred<F64>(a, my[j], qx, qy, qz, qp.x, qp.y, qp.z, x, y, z);
            PASS2_ACC(x, y, z);
        }
    }
#undef PASS2_ACC

    // ---- Cholesky  G = L L^T  (lower triangle packed row-wise), solve -------
    // packed index of (i,j), j<=i : i*(i+1)/2 + j
#define GI(i, j) ((i) * ((i) + 1) / 2 + (j))
    double dinv[6];                 // 1 / L_jj
    bool well = true;               // every pivot ratio d_j / g_jj above kPivotRatioMin: the columns are far from dependent
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = g[GI(j, j)];
#pragma unroll
        for (int p = 0; p < j; ++p) d -= g[GI(j, p)] * g[GI(j, p)];
        well = well && d > kPivotRatioMin * g[GI(j, j)];               // (false for NaN and for an all-zero column too)
        const double inv = rsqrt(d);                    // NaN for a non-positive pivot: such rows are handed over
        dinv[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = g[GI(i, j)];
#pragma unroll
            for (int p = 0; p < j; ++p) s -= g[GI(i, p)] * g[GI(j, p)];
            g[GI(i, j)] = s * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {   // L y = b
        double s = b[i];
#pragma unroll
        for (int p = 0; p < i; ++p) s -= g[GI(i, p)] * b[p];
        b[i] = s * dinv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {  // L^T x = y
        double s = b[i];
#pragma unroll
        for (int p = i + 1; p < 6; ++p) s -= g[GI(p, i)] * b[p];
        b[i] = s * dinv[i];
    }
#undef GI
    // An ill-conditioned design matrix (neighbours on a line or a planar curve, anisotropic lattices, duplicates): the
    // reference's lstsq (pct:359) is LAPACK gelsd -- an SVD-based solve with the singular values below
    // eps * max(m, 6) * sigma_1 cut off -- whose answer the normal equations cannot follow.  k_fit_svd redoes the row.
    if (!well) hand_over();
    if constexpr (OUT64) {
        // diagnostics: what the float32 rounding of the coefficients (pct:359) and the float32 curvature arithmetic
        // (pct:403-422) cost -- the solution as solved, and the same formulas in float64
        double* co64 = a.coefs64 + out * 6;
        for (int j = 0; j < 6; ++j) co64[j] = b[j];
        const double Fx = b[3], Fy = b[4], Fxx = 2.0 * b[0], Fyy = 2.0 * b[1], Fxy = b[2];
        const double wgt = (1.0 + Fx * Fx) + Fy * Fy;
        a.K64[out] = (Fxx * Fyy - Fxy * Fxy) / (wgt * wgt);
        a.H64[out] = (((1.0 + Fx * Fx) * Fyy - ((2.0 * Fx) * Fy) * Fxy) + (1.0 + Fy * Fy) * Fxx) / (2.0 * (wgt * sqrt(wgt)));
        return;
    }
    const float A = (float)b[0], B = (float)b[1], C = (float)b[2];
    const float D = (float)b[3], E = (float)b[4], F = (float)b[5];
    float* co = a.coefs + out * 6;
    co[0] = A; co[1] = B; co[2] = C; co[3] = D; co[4] = E; co[5] = F;

    float Kg, Kh;
    monge_curvatures(A, B, C, D, E, Kg, Kh);
    a.K[out] = Kg;
    a.H[out] = Kh;
    a.H2[out] = Kh * Kh;
}

// calculate_explicit_quadratic_curvatures alone (pct:398-431), one thread per row
__global__ __launch_bounds__(256) void k_curv(const float* __restrict__ coefs, int64_t rows, float* __restrict__ K,
                                              float* __restrict__ H, float* __restrict__ H2) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* c = coefs + r * 6;
    float Kg, Kh;
    monge_curvatures(c[0], c[1], c[2], c[3], c[4], Kg, Kh);
    K[r] = Kg;
    H[r] = Kh;
    H2[r] = Kh * Kh;
}

// ---------------------------------------------------------------------------
// numpy.linalg.lstsq(X, z, rcond=None) (pct:359) for the rows k_fit handed over: float32 design rows widened to
// float64, LAPACK gelsd semantics -- the least-squares solution through the singular value decomposition of X, with
// the singular values sigma_i <= eps * max(m, 6) * sigma_1 treated as zero (eps = 2^-52: NumPy takes the machine
// epsilon of the float64 COMPUTE type, also for float32 input), i.e. the minimum-norm solution of a rank-deficient
// system -- and the result rounded to float32.
//   1. the m x 6 design matrix is never stored: its rows are folded one by one into the 6 x 6 triangular factor R
//      and the rotated right-hand side y by Givens rotations (X = Q R, y = Q^T z: what gelsd's first QR step yields,
//      and no squaring of the condition number as in the normal equations);
//   2. one-sided Jacobi on R: R V = U Sigma, singular values with high relative accuracy;
//   3. c = V Sigma^+ U^T y over the singular values above the cut-off.
// One thread per listed row; the list length is read on the device (no host round trip), the grid is fixed.
// ---------------------------------------------------------------------------
// Streaming least squares on the reference's design rows [a^2, b^2, ab, a, b, 1] (pct:358), gelsd semantics.
struct Lstsq6 {
    double A[6][6];        // A[c][r]: column c of the triangular factor (rows r <= c in use), after solve(): R V
    double y[6];           // Q^T z

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            y[c] = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) A[c][r] = 0;
        }
    }

    // one neighbour: its float32 coordinates in the rotated frame; the row is annihilated into R by six Givens rotations
    __device__ __forceinline__ void add(float pa, float pb, float pc) {
        double x[6] = {(double)(pa * pa), (double)(pb * pb), (double)(pa * pb), (double)pa, (double)pb, 1.0};   // pct:358
        double zi = (double)pc;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double xr = x[r];
            if (xr != 0.0) {
                const double rr = A[r][r];
                const double h = sqrt(rr * rr + xr * xr);
                const double inv = 1.0 / h;
                const double c = rr * inv, sn = xr * inv;
                A[r][r] = h;
#pragma unroll
                for (int c2 = r + 1; c2 < 6; ++c2) {
                    const double t = A[c2][r];
                    A[c2][r] = c * t + sn * x[c2];
                    x[c2] = c * x[c2] - sn * t;
                }
                const double t = y[r];
                y[r] = c * t + sn * zi;
                zi = c * zi - sn * t;
            }
        }
    }

    // one-sided Jacobi on R (R V = U Sigma), then c = V Sigma^+ U^T y over the singular values above
    // eps * max(m, 6) * sigma_1 (numpy.linalg.lstsq(rcond=None) on float32 input computes in float64: eps = 2^-52)
    __device__ __forceinline__ void solve(int m, double (&sol)[6]) {
        double V[6][6];        // V[c][r]: column c
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int r = 0; r < 6; ++r) V[c][r] = c == r ? 1.0 : 0.0;
#pragma unroll 1
        for (int sweep = 0; sweep < 40; ++sweep) {
            bool rotated = false;
#pragma unroll
            for (int p = 0; p < 5; ++p) {
#pragma unroll
                for (int q = p + 1; q < 6; ++q) {
                    double al = 0, be = 0, ga = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        al = fma(A[p][r], A[p][r], al);
                        be = fma(A[q][r], A[q][r], be);
                        ga = fma(A[p][r], A[q][r], ga);
                    }
                    if (ga != 0.0 && fabs(ga) > 1e-15 * sqrt(al * be)) {
                        rotated = true;
                        const double zeta = (be - al) / (2.0 * ga);
                        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = rsqrt(1.0 + t * t), sn = c * t;
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
                            const double ap = A[p][r], aq = A[q][r];
                            A[p][r] = c * ap - sn * aq;
                            A[q][r] = sn * ap + c * aq;
                            const double vp = V[p][r], vq = V[q][r];
                            V[p][r] = c * vp - sn * vq;
                            V[q][r] = sn * vp + c * vq;
                        }
                    }
                }
            }
            if (!__any(rotated)) break;
        }
        double sig2[6], smax2 = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double t = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) t = fma(A[c][r], A[c][r], t);
            sig2[c] = t;
            smax2 = fmax(smax2, t);
        }
        const double rcond = 2.220446049250313e-16 * (double)(m > 6 ? m : 6);
        const double cut = rcond * sqrt(smax2);
#pragma unroll
        for (int r = 0; r < 6; ++r) sol[r] = 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (sqrt(sig2[c]) > cut) {
                double t = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r) t = fma(A[c][r], y[r], t);
                const double w = t / sig2[c];
#pragma unroll
                for (int r = 0; r < 6; ++r) sol[r] = fma(V[c][r], w, sol[r]);
            }
        }
    }
};

template <bool F64, bool OUT64>
__global__ __launch_bounds__(64) void k_fit_svd(FitArgs a, const int* __restrict__ list, const int* __restrict__ list_count,
                                                long long* __restrict__ host_count) {
    const int total = *list_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *host_count = (a.svd_accumulate ? *host_count : 0ll) + total;   // pinned host word: read at the caller's next synchronisation
    const float nanf_ = __int_as_float(0x7fc00000);
    for (int64_t it = (int64_t)blockIdx.x * 64 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 64) {
        const int64_t row = list[it];
        const int64_t qid = a.row_query32 ? (int64_t)a.row_query32[row] : a.row_query ? a.row_query[row] : row + a.row_offset;
        const float4 qp = a.pts[qid];
        const int64_t out = a.out_by_row ? row : (int64_t)__float_as_int(qp.w) - a.out_base;
        double qx = qp.x, qy = qp.y, qz = qp.z;
        if (F64) { const double4 qd = a.ptsd[qid]; qx = qd.x; qy = qd.y; qz = qd.z; }
        const int m = a.cnt ? a.cnt[row] : a.k;
        const int* my = a.table + row * a.pitch;

        // ---- pass 1: moments -> tangent-plane rotation (as k_fit) ----------------------------------------------
        double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
        double fx = 0, fy = 0, fz = 0, lx = 0, ly = 0, lz = 0;
        for (int j = 0; j < m; ++j) {
            double x, y, z;
            load_centred<F64>(a, (int)min((unsigned)my[j], a.n_pts - 1u), qx, qy, qz, qp.x, qp.y, qp.z, x, y, z);
            if (j == 0) { fx = x; fy = y; fz = z; }
            lx = x; ly = y; lz = z;
            sx += x; sy += y; sz += z;
            sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); sxz = fma(x, z, sxz);
            syy = fma(y, y, syy); syz = fma(y, z, syz); szz = fma(z, z, szz);
        }
        double rot[9];
        plane_rotation<F64>(m, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz, fx, fy, fz, lx, ly, lz, rot);

        // ---- pass 2: the design rows, one by one ---------------------------------------------------------------
        Lstsq6 ls;
        ls.init();
#pragma unroll 1
        for (int j = 0; j < m; ++j) {
            double px, py, pz;
            load_centred<F64>(a, (int)min((unsigned)my[j], a.n_pts - 1u), qx, qy, qz, qp.x, qp.y, qp.z, px, py, pz);
            ls.add((float)((rot[0] * px + rot[1] * py) + rot[2] * pz), (float)((rot[3] * px + rot[4] * py) + rot[5] * pz),
                   (float)((rot[6] * px + rot[7] * py) + rot[8] * pz));
        }
        double sol[6];
        ls.solve(m, sol);
        if (m < 2) {
#pragma unroll
            for (int r = 0; r < 6; ++r) sol[r] = (double)nanf_;
        }
        if constexpr (OUT64) {
            double* co64 = a.coefs64 + out * 6;
            for (int j = 0; j < 6; ++j) co64[j] = sol[j];
            const double Fx = sol[3], Fy = sol[4], Fxx = 2.0 * sol[0], Fyy = 2.0 * sol[1], Fxy = sol[2];
            const double wgt = (1.0 + Fx * Fx) + Fy * Fy;
            a.K64[out] = (Fxx * Fyy - Fxy * Fxy) / (wgt * wgt);
            a.H64[out] = (((1.0 + Fx * Fx) * Fyy - ((2.0 * Fx) * Fy) * Fxy) + (1.0 + Fy * Fy) * Fxx) / (2.0 * (wgt * sqrt(wgt)));
        } else {
            const float Af = (float)sol[0], Bf = (float)sol[1], Cf = (float)sol[2], Df = (float)sol[3], Ef = (float)sol[4];
            float* co = a.coefs + out * 6;
            co[0] = Af; co[1] = Bf; co[2] = Cf; co[3] = Df; co[4] = Ef; co[5] = (float)sol[5];
            float Kg, Kh;
            monge_curvatures(Af, Bf, Cf, Df, Ef, Kg, Kh);
            a.K[out] = Kg;
            a.H[out] = Kh;
            a.H2[out] = Kh * Kh;
        }
    }
}

// ---------------------------------------------------------------------------
// The two per-neighbourhood staticmethods of the class surface on their own ("methods of rows A1-A10"),
// batched: one thread per neighbourhood of m points.
//   k_plane_rotate   get_best_fit_plane_and_rotate (pct:270-321): np.cov (mean first, then centred products, float64,
//                    ddof = 1) -> normal -> flip by points[-1] - points[0] (native dtype) -> Rodrigues -> R p, float64
//   k_quadric_rows   fit_quadratic_surface (pct:331-360): float32 points -> lstsq (gelsd semantics) -> float32 (6,)
// ---------------------------------------------------------------------------
template <bool F64>
__global__ __launch_bounds__(64) void k_plane_rotate(const void* __restrict__ nbrs, int64_t batch, int m, double* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    const auto at = [&](int j, double& x, double& y, double& z) {
        if (F64) { const double* p = (const double*)nbrs + (b * m + j) * 3; x = p[0]; y = p[1]; z = p[2]; }
        else { const float* p = (const float*)nbrs + (b * m + j) * 3; x = (double)p[0]; y = (double)p[1]; z = (double)p[2]; }
    };
    double mx = 0, my = 0, mz = 0;
    for (int j = 0; j < m; ++j) { double x, y, z; at(j, x, y, z); mx += x; my += y; mz += z; }
    mx /= (double)m; my /= (double)m; mz /= (double)m;
    double sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
    for (int j = 0; j < m; ++j) {
        double x, y, z;
        at(j, x, y, z);
        x -= mx; y -= my; z -= mz;
        sxx = fma(x, x, sxx); sxy = fma(x, y, sxy); sxz = fma(x, z, sxz);
        syy = fma(y, y, syy); syz = fma(y, z, syz); szz = fma(z, z, szz);
    }
    double fx, fy, fz, lx, ly, lz;
    at(0, fx, fy, fz);
    at(m - 1, lx, ly, lz);
    double rot[9];
    plane_rotation<F64>(m, 0.0, 0.0, 0.0, sxx, sxy, sxz, syy, syz, szz, fx, fy, fz, lx, ly, lz, rot);
    for (int j = 0; j < m; ++j) {
        double x, y, z;
        at(j, x, y, z);
        double* o = out + (b * m + j) * 3;
        o[0] = (rot[0] * x + rot[1] * y) + rot[2] * z;
        o[1] = (rot[3] * x + rot[4] * y) + rot[5] * z;
        o[2] = (rot[6] * x + rot[7] * y) + rot[8] * z;
    }
}

__global__ __launch_bounds__(64) void k_quadric_rows(const float* __restrict__ pts, int64_t batch, int m, float* __restrict__ coefs) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    Lstsq6 ls;
    ls.init();
#pragma unroll 1
    for (int j = 0; j < m; ++j) {
        const float* p = pts + (b * m + j) * 3;
        ls.add(p[0], p[1], p[2]);
    }
    double sol[6];
    ls.solve(m, sol);
    for (int j = 0; j < 6; ++j) coefs[b * 6 + j] = (float)sol[j];
}

// explicit_quadratic_neighbor_study (pct:756-761): row (s, n) = the sample point itself followed by its n nearest
// neighbours, n = n_lo .. n_hi, taken from the resident neighbour table
__global__ __launch_bounds__(256) void k_prefix_rows(const int* __restrict__ sample_row, const int* __restrict__ owned_pos,
                                                     int q_begin, int64_t n_samples, int n_lo, int n_hi,
                                                     const int* __restrict__ nbr_pos, int nbr_pitch,
                                                     int* __restrict__ table, int pitch, int* __restrict__ cnt,
                                                     int64_t* __restrict__ row_query) {
    const int nn = n_hi - n_lo + 1;
    const int64_t row = blockIdx.x;
    if (row >= n_samples * nn) return;
    const int64_t s = row / nn;
    const int n = n_lo + (int)(row - s * nn);
    const int trow = sample_row[s];                                       // neighbour-table row of the sample
    const int q = owned_pos ? owned_pos[trow] : trow + q_begin;           // its id in the point array the table refers to
    int* dst = table + row * pitch;
    for (int j = threadIdx.x; j <= n; j += 256) dst[j] = j == 0 ? q : nbr_pos[(int64_t)trow * nbr_pitch + (j - 1)];
    if (threadIdx.x == 0) { cnt[row] = n + 1; row_query[row] = q; }
}

// public rows [first, first + rows) of the owned range out of row-ordered results
__global__ __launch_bounds__(256) void k_gather_fit(const int* __restrict__ row_of, int64_t first, int64_t rows,
                                                    const float* __restrict__ coefs, const float* __restrict__ K,
                                                    const float* __restrict__ H, const float* __restrict__ H2,
                                                    float* __restrict__ o_coefs, float* __restrict__ o_K,
                                                    float* __restrict__ o_H, float* __restrict__ o_H2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const int64_t r = row_of[first + i];
    if (o_coefs) {
#pragma unroll
        for (int j = 0; j < 6; ++j) o_coefs[i * 6 + j] = coefs[r * 6 + j];
    }
    if (o_K) o_K[i] = K[r];
    if (o_H) o_H[i] = H[r];
    if (o_H2) o_H2[i] = H2[r];
}

int launch(pct_ctx* ctx, const FitArgs& a0, bool f64) {
    FitArgs a = a0;
    a.kp = a.k | 1;
    if (a.row_mask && ((size_t)kFitBlock * a.kp * sizeof(int) > 64 * 1024 || a.coefs64))
        return pct_fail(ctx, PCT_ERR_INVALID, "masked fit: rows too long");       // (k <= 127 on this path)
    size_t lds = (size_t)kFitBlock * a.kp * sizeof(int);
    const bool staged = lds <= 64 * 1024;                  // the default dynamic LDS limit of a launch
    if (!staged) lds = 0;
    int blocks = (int)((a.rows + kFitBlock - 1) / kFitBlock);
    if (blocks <= 0) return PCT_OK;
    a.blocks_per_xcd = pct_getenv("PCT_NO_XCD_MAP") ? 0 : (blocks + 7) / 8;
    if (a.blocks_per_xcd) blocks = a.blocks_per_xcd * 8;
    // rows for k_fit_svd: list + its length (first word of the buffer's 64-byte head)
    // two counts used in turn: a launch counts in one and clears the other for the launch after it (both cleared once per
    // allocation) -- the 64-byte memset in front of every fit was a launch of its own, 4 us and a dispatch gap
    PCT_TRY(pct_reserve(ctx, &ctx->fit_flag, 64 + (size_t)a.rows * sizeof(int)));
    if (ctx->fit_flag.p != ctx->fit_flag_seen || ctx->fit_flag.cap != ctx->fit_flag_cap_seen) {
        PCT_HIP(ctx, hipMemsetAsync(ctx->fit_flag.p, 0, 64, ctx->stream));
        ctx->fit_flag_seen = ctx->fit_flag.p;
        ctx->fit_flag_cap_seen = ctx->fit_flag.cap;
        ctx->fit_parity = 0;
    }
    a.stat_src = nullptr;
    a.stat_dst = nullptr;
    if (ctx->stats_mirror_req && ctx->counters.p) {
        a.stat_src = (const unsigned long long*)ctx->counters.p;
        a.stat_dst = (unsigned long long*)(ctx->pin + (ctx->fit_par ? 256 : 192));
        ctx->stats_mirrored = true;
    }
    ctx->stats_mirror_req = false;
    a.flag_count = (int*)ctx->fit_flag.p + ctx->fit_parity;
    a.flag_next = (int*)ctx->fit_flag.p + (ctx->fit_parity ^ 1);
    ctx->fit_parity ^= 1;
    a.flag_list = (int*)ctx->fit_flag.p + 16;
#define PCT_FIT_LAUNCH(F_, O_)                                                                                      \
    do {                                                                                                            \
        if (a.row_mask && staged && !O_) PCT_LAUNCH((k_fit<F_, false, true, true>), dim3(blocks), dim3(kFitBlock), lds, ctx->stream, a); \
        else if (staged) PCT_LAUNCH((k_fit<F_, O_, true>), dim3(blocks), dim3(kFitBlock), lds, ctx->stream, a);  \
        else PCT_LAUNCH((k_fit<F_, O_, false>), dim3(blocks), dim3(kFitBlock), 0, ctx->stream, a);          \
    } while (0)
    if (a.coefs64) {
        if (f64) PCT_FIT_LAUNCH(true, true);
        else PCT_FIT_LAUNCH(false, true);
    } else if (f64) PCT_FIT_LAUNCH(true, false);
    else PCT_FIT_LAUNCH(false, false);
#undef PCT_FIT_LAUNCH
    PCT_HIP(ctx, hipGetLastError());
    // the rows handed over: fixed grid, the list length is read on the device
    const int sblocks = blocks < 1024 ? blocks : 1024;
    long long* note = (long long*)(ctx->pin + 2048 + 8 * ctx->fit_par);
    if (a.coefs64) {
        if (f64)
            PCT_LAUNCH((k_fit_svd<true, true>), dim3(sblocks), dim3(64), 0, ctx->stream, a, a.flag_list, a.flag_count, note);
        else
            PCT_LAUNCH((k_fit_svd<false, true>), dim3(sblocks), dim3(64), 0, ctx->stream, a, a.flag_list, a.flag_count, note);
    } else if (f64)
        PCT_LAUNCH((k_fit_svd<true, false>), dim3(sblocks), dim3(64), 0, ctx->stream, a, a.flag_list, a.flag_count, note);
    else
        PCT_LAUNCH((k_fit_svd<false, false>), dim3(sblocks), dim3(64), 0, ctx->stream, a, a.flag_list, a.flag_count, note);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

}  // namespace

// one pass of the density-adaptive sweep (pct_levels.hip): the rows it answered (row_done), fitted while its table is
// still in ITS cell order -- gathers stay local; results go to public order.  The merged public-space table is fitted
// only when a caller asks for the table first and the fit later (1.3 ms instead of 0.25 at 1 M points).
int pct_launch_fit_pass(pct_ctx* ctx, int64_t rows) {
    const int64_t nq = ctx->q_end - ctx->q_begin;
    PCT_TRY(pct_reserve(ctx, &ctx->coefs, (size_t)nq * 6 * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->K, (size_t)nq * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->H, (size_t)nq * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->H2, (size_t)nq * sizeof(float)));
    FitArgs a = {};
    a.pts = (const float4*)ctx->sorted4.p;
    a.ptsd = ctx->has_f64 ? (const double4*)ctx->sorted4d.p : nullptr;
    a.table = (const int*)ctx->nbr_pos.p;
    a.cnt = ctx->eps > 0 ? (const int*)ctx->nbr_cnt.p : nullptr;
    a.row_query32 = (const int*)ctx->owned_pos.p;
    a.rows = rows;
    a.k = ctx->k;
    a.pitch = (ctx->k + 3) & ~3;
    a.out_by_row = 0;
    a.out_base = ctx->q_begin;
    a.q_begin = (int)ctx->q_begin;
    a.q_end = (int)ctx->q_end;
    a.coefs = (float*)ctx->coefs.p;
    a.K = (float*)ctx->K.p;
    a.H = (float*)ctx->H.p;
    a.H2 = (float*)ctx->H2.p;
    a.n_pts = (unsigned)ctx->n_grid;
    a.row_mask = (const int*)ctx->row_done.p;
    a.svd_accumulate = 1;                      // (pct_knn_levels zeroes the host word before the first pass)
    return launch(ctx, a, ctx->has_f64);
}

// fit from the device-resident neighbour table left by the sweep
int pct_launch_fit_table(pct_ctx* ctx) {
    if (ctx->levels_fitted) {                // the passes of the density-adaptive sweep have fitted their rows already
        ctx->levels_fitted = false;
        ctx->fit_row_order = false;
        return PCT_OK;
    }
    const int64_t nq = ctx->q_end - ctx->q_begin;
    PCT_TRY(pct_reserve(ctx, &ctx->coefs, (size_t)nq * 6 * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->K, (size_t)nq * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->H, (size_t)nq * sizeof(float)));
    PCT_TRY(pct_reserve(ctx, &ctx->H2, (size_t)nq * sizeof(float)));
    FitArgs a = {};
    const bool sorted = ctx->knn_sorted_space;
    a.pts = (const float4*)(sorted ? ctx->sorted4.p : ctx->pts4.p);
    a.ptsd = ctx->has_f64 ? (const double4*)(sorted ? ctx->sorted4d.p : ctx->pts4d.p) : nullptr;
    a.table = (const int*)ctx->nbr_pos.p;
    a.cnt = ctx->eps > 0 ? (const int*)ctx->nbr_cnt.p : nullptr;
    a.row_query = nullptr;
    a.row_query32 = sorted ? (const int*)ctx->owned_pos.p : nullptr;    // table row -> sorted position of its query
    a.row_offset = sorted ? 0 : ctx->q_begin;                           // exhaustive sweep: row = public index - q_begin
    a.rows = nq;
    a.k = ctx->k;
    a.pitch = ctx->nbr_pitch;
    // Grid sweep: results stay in TABLE-ROW (cell) order -- a wave's 64 rows are consecutive, so the stores
    // coalesce; written in public order they were 36 B scattered per point and cost 4x the bytes at the HBM
    // (WRITE_SIZE 145 MB vs 36 MB).  pct_get_fit gathers public rows on request (pct_launch_gather_fit).
    a.out_by_row = sorted ? 1 : 0;
    ctx->fit_row_order = sorted;
    a.out_base = ctx->q_begin;
    a.q_begin = (int)ctx->q_begin;
    a.q_end = (int)ctx->q_end;
    a.coefs = (float*)ctx->coefs.p;
    a.K = (float*)ctx->K.p;
    a.H = (float*)ctx->H.p;
    a.H2 = (float*)ctx->H2.p;
    a.n_pts = (unsigned)(sorted ? ctx->n_grid : ctx->n);
    return launch(ctx, a, ctx->has_f64);
}

int pct_launch_gather_fit(pct_ctx* ctx, int64_t first, int64_t rows, float* d_coefs, float* d_K, float* d_H, float* d_H2) {
    PCT_LAUNCH(k_gather_fit, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, (const int*)ctx->row_of.p,
                       first, rows, (const float*)ctx->coefs.p, (const float*)ctx->K.p, (const float*)ctx->H.p,
                       (const float*)ctx->H2.p, d_coefs, d_K, d_H, d_H2);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

// fit from explicit neighbour rows, outputs row-aligned.  sorted_space: ids refer to the cell-sorted arrays.
int pct_launch_fit_rows(pct_ctx* ctx, const int32_t* d_idx, const int32_t* d_cnt, const int64_t* d_query,
                        int64_t rows, int32_t k, int32_t pitch, float* d_coefs, float* d_K, float* d_H, float* d_H2,
                        bool sorted_space) {
    FitArgs a = {};
    a.pts = (const float4*)(sorted_space ? ctx->sorted4.p : ctx->pts4.p);
    a.ptsd = ctx->has_f64 ? (const double4*)(sorted_space ? ctx->sorted4d.p : ctx->pts4d.p) : nullptr;
    a.table = d_idx;
    a.cnt = d_cnt;
    a.row_query = d_query;
    a.rows = rows;
    a.k = k;
    a.pitch = pitch;
    a.out_by_row = 1;
    a.out_base = 0;
    a.q_begin = 0;
    a.q_end = (int)ctx->n;
    a.coefs = d_coefs;
    a.K = d_K;
    a.H = d_H;
    a.H2 = d_H2;
    a.n_pts = (unsigned)(sorted_space ? ctx->n_grid : ctx->n);
    // (rows with fewer than 6 points get lstsq's minimum-norm answer from k_fit_svd, as in every launch)
    return launch(ctx, a, ctx->has_f64);
}

// the diagnostics variant of pct_launch_fit_rows: float64 coefficients and curvatures, public-space ids
int pct_launch_fit_rows_f64(pct_ctx* ctx, const int32_t* d_idx, const int32_t* d_cnt, const int64_t* d_query, int64_t rows,
                            int32_t k, int32_t pitch, double* d_coefs, double* d_K, double* d_H) {
    PCT_TRY(pct_ensure_plain_records(ctx));
    FitArgs a = {};
    a.pts = (const float4*)ctx->qpts4.p;
    a.ptsd = ctx->has_f64 ? (const double4*)ctx->pts4d.p : nullptr;
    a.table = d_idx;
    a.cnt = d_cnt;
    a.row_query = d_query;
    a.rows = rows;
    a.k = k;
    a.pitch = pitch;
    a.out_by_row = 1;
    a.q_begin = 0;
    a.q_end = (int)ctx->n;
    a.coefs64 = d_coefs;
    a.K64 = d_K;
    a.H64 = d_H;
    a.n_pts = (unsigned)ctx->n;
    return launch(ctx, a, ctx->has_f64);
}

int pct_launch_prefix_rows(pct_ctx* ctx, const int* d_sample_row, int64_t n_samples, int n_lo, int n_hi, int* d_table,
                           int pitch, int* d_cnt, int64_t* d_row_query) {
    const int64_t rows = n_samples * (n_hi - n_lo + 1);
    PCT_LAUNCH(k_prefix_rows, dim3((unsigned)rows), dim3(256), 0, ctx->stream, d_sample_row,
                       ctx->knn_sorted_space ? (const int*)ctx->owned_pos.p : nullptr, (int)ctx->q_begin, n_samples, n_lo, n_hi,
                       (const int*)ctx->nbr_pos.p, ctx->nbr_pitch, d_table, pitch, d_cnt, d_row_query);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_launch_curvatures(pct_ctx* ctx, const float* d_coefs, int64_t rows, float* d_K, float* d_H, float* d_H2) {
    const int blocks = (int)((rows + 255) / 256);
    if (blocks <= 0) return PCT_OK;
    PCT_LAUNCH(k_curv, dim3(blocks), dim3(256), 0, ctx->stream, d_coefs, rows, d_K, d_H, d_H2);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_launch_plane_rotate(pct_ctx* ctx, const void* d_nbrs, bool f64, int64_t batch, int32_t m, double* d_out) {
    const unsigned blocks = (unsigned)((batch + 63) / 64);
    if (f64)
        PCT_LAUNCH(k_plane_rotate<true>, dim3(blocks), dim3(64), 0, ctx->stream, d_nbrs, batch, m, d_out);
    else
        PCT_LAUNCH(k_plane_rotate<false>, dim3(blocks), dim3(64), 0, ctx->stream, d_nbrs, batch, m, d_out);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}

int pct_launch_quadric_rows(pct_ctx* ctx, const float* d_pts, int64_t batch, int32_t m, float* d_coefs) {
    PCT_LAUNCH(k_quadric_rows, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, ctx->stream, d_pts, batch, m, d_coefs);
    PCT_HIP(ctx, hipGetLastError());
    return PCT_OK;
}
