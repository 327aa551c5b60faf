// developer tool: issue cost of VALU instruction kinds on gfx950, per SIMD, as a function of waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 2000, UNROLL = 16;

#define KERNEL(NAME, BODY, ...)                                                              \
    __global__ void NAME(float* out, long long* cyc) {                                       \
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, fb = 1.0001f;                         \
        double d0 = threadIdx.x + 1.0, d1 = 1, d2 = 2, d3 = 3, db = 1.0001;                   \
        unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, ub = 7;                            \
        float2 p0 = make_float2(threadIdx.x, 1.f), p1 = p0, p2 = p0, p3 = p0, pb = make_float2(1.0001f, 1.0002f); \
        int s0 = 1, s1 = 2, s2 = 3, s3 = 0x55555555;                                          \
        const long long t0 = clock64();                                                      \
        for (int i = 0; i < ITERS; ++i) {                                                    \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY BODY __VA_ARGS__); \
        }                                                                                    \
        const long long t1 = clock64();                                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + (float)(u0 + u1 + u2 + u3) + p0.x + p1.y + p2.x + p3.y + (float)(s0 + s1 + s2 + s3); \
    }

KERNEL(k_fma32, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n",
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fb))
KERNEL(k_med3, "v_med3_u32 %0, %0, %4, %1\n v_med3_u32 %1, %1, %4, %2\n v_med3_u32 %2, %2, %4, %3\n v_med3_u32 %3, %3, %4, %0\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub))
KERNEL(k_dpp, "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %4 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n",
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fb))
KERNEL(k_pkfma, "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n",
       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb))
KERNEL(k_fma64, "v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n",
       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db))
KERNEL(k_mul64, "v_mul_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n",
       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db))
KERNEL(k_cvt64, "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n",
       : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3))
KERNEL(k_cvtu32_64, "v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %5\n v_cvt_u32_f64 %2, %6\n v_cvt_u32_f64 %3, %7\n",
       : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3))
KERNEL(k_rsq64, "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n",
       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3))
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n",
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fb) : "vcc")
KERNEL(k_cmp_bcnt, "v_cmp_lt_f32 vcc, %0, %4\n s_bcnt1_i32_b64 %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n s_bcnt1_i32_b64 %6, vcc\n",
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fb), "s"(s0), "s"(s1) : "vcc", "scc")
KERNEL(k_salu, "s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %2, %2, 1\n s_add_i32 %3, %3, 1\n",
       : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc")
KERNEL(k_mix, "v_fma_f32 %0, %0, %4, %0\n s_add_i32 %5, %5, 1\n v_fma_f32 %1, %1, %4, %1\n s_add_i32 %6, %6, 1\n",
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fb), "s"(s0), "s"(s1) : "scc")
KERNEL(k_readlane, "v_readlane_b32 %4, %0, 3\n v_readlane_b32 %5, %1, 5\n v_readlane_b32 %4, %2, 3\n v_readlane_b32 %5, %3, 5\n",
       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1))
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_hi_u32_b32 %3, %4, %3\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(s3))
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_u32 %3, %3, %4\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub) : "vcc")

KERNEL(k_swizzle, "ds_swizzle_b32 %0, %0 offset:0x041f\n ds_swizzle_b32 %1, %1 offset:0x081f\n ds_swizzle_b32 %2, %2 offset:0x101f\n ds_swizzle_b32 %3, %3 offset:0x201f\n s_waitcnt lgkmcnt(0)\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3))
KERNEL(k_bperm, "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub))
KERNEL(k_swz_med3, "ds_swizzle_b32 %2, %0 offset:0x041f\n ds_swizzle_b32 %3, %1 offset:0x041f\n s_waitcnt lgkmcnt(1)\n v_med3_u32 %0, %0, %2, %4\n s_waitcnt lgkmcnt(0)\n v_med3_u32 %1, %1, %3, %4\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub))
KERNEL(k_dpp_med3, "v_mov_b32_dpp %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_med3_u32 %0, %0, %2, %4\n v_mov_b32_dpp %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_med3_u32 %1, %1, %3, %4\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub))
KERNEL(k_permlane32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3))
KERNEL(k_valu_lds_mix, "ds_swizzle_b32 %2, %0 offset:0x041f\n v_med3_u32 %0, %0, %4, %1\n v_med3_u32 %1, %1, %4, %0\n v_med3_u32 %3, %3, %4, %1\n s_waitcnt lgkmcnt(0)\n",
       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(ub))

template <class K>
int run(const char* name, K kern, int n_instr_per_body) {
    float* out; long long* cyc;
    CHECK(hipMalloc(&out, 1 << 24));
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-12s", name);
    for (int wps : {1, 2, 3, 4, 6, 8}) {            // waves per SIMD: blocks of 256 threads (one wave per SIMD each), wps blocks per CU
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, out, cyc);     // warm-up
        CHECK(hipDeviceSynchronize());
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, out, cyc);
        hipEventRecord(e1);
        CHECK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c;
        CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double per_wave = (double)ITERS * UNROLL * n_instr_per_body;
        // whole-grid time in 2.4 GHz cycles per instruction per SIMD (every SIMD runs wps waves), and wave 0's own clock64 view
        printf("  w%d: %6.3f (%5.2f)", wps, ms * 1e-3 * 2.4e9 / (per_wave * wps), (double)c / per_wave);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    printf("cycles (at 2.4 GHz, from the whole grid's hipEvent time) per instruction per SIMD; in brackets: clock64 ticks per instruction seen by one wave\n");
    run("fma32", k_fma32, 4);
    run("med3", k_med3, 4);
    run("dpp_mov", k_dpp, 4);
    run("pk_fma", k_pkfma, 4);
    run("fma64", k_fma64, 4);
    run("mul/add64", k_mul64, 4);
    run("cvt_f64_f32", k_cvt64, 4);
    run("cvt_u32_f64", k_cvtu32_64, 4);
    run("rsq64", k_rsq64, 4);
    run("cmp", k_cmp, 4);
    run("cmp+bcnt", k_cmp_bcnt, 4);
    run("salu", k_salu, 4);
    run("fma+salu", k_mix, 4);
    run("readlane", k_readlane, 4);
    run("mbcnt", k_mbcnt, 4);
    run("int misc", k_lshl_or, 4);
    run("ds_swizzle", k_swizzle, 4);
    run("ds_bpermute", k_bperm, 4);
    run("swz+med3 x2", k_swz_med3, 4);
    run("dpp+med3 x2", k_dpp_med3, 4);
    run("permlane swap", k_permlane32, 4);
    run("1swz+3med3", k_valu_lds_mix, 4);
    // wall-clock calibration of the tick
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fma32, dim3(256), dim3(256), 0, 0, out, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("fma32 one wave per SIMD: %lld ticks in %.3f ms -> %.1f MHz tick\n", c, ms, c / ms / 1e3);
    return 0;
}
