// Developer microbenchmark: returning integer atomics on random addresses (the cell-list histogram's pattern),
// agent scope vs workgroup scope (performed in the XCD's L2), 1 M operations over `cells` counters.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE>
__global__ __launch_bounds__(256) void k_atomics(const int* __restrict__ idx, int n, int* __restrict__ counters, int* __restrict__ out, int copies, int cells) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = idx[i];
    if (copies > 1) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        c += (int)(xcc & 7u) * cells;
    }
    out[i] = __hip_atomic_fetch_add(&counters[c], 1, __ATOMIC_RELAXED, SCOPE);
}
int main() {
    const int n = 1 << 20;
    for (int cells : {56000, 1 << 20}) {
        std::vector<int> h(n);
        unsigned s = 12345;
        for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 8) % (unsigned)cells); }
        int *d_idx, *d_cnt, *d_out;
        hipMalloc(&d_idx, n * 4); hipMalloc(&d_cnt, (size_t)cells * 8 * 4); hipMalloc(&d_out, n * 4);
        hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipMemset(d_cnt, 0, (size_t)cells * 8 * 4);
                hipEventRecord(e0);
                if (mode == 0) k_atomics<__HIP_MEMORY_SCOPE_AGENT><<<n / 256, 256>>>(d_idx, n, d_cnt, d_out, 1, cells);
                if (mode == 1) k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP><<<n / 256, 256>>>(d_idx, n, d_cnt, d_out, 1, cells);
                if (mode == 2) k_atomics<__HIP_MEMORY_SCOPE_WORKGROUP><<<n / 256, 256>>>(d_idx, n, d_cnt, d_out, 8, cells);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            // do the counts add up? (workgroup scope on ONE copy from eight XCDs need not)
            std::vector<int> c((size_t)cells * 8);
            hipMemcpy(c.data(), d_cnt, c.size() * 4, hipMemcpyDeviceToHost);
            long long tot = 0; for (int v : c) tot += v;
            printf("cells %8d  %-34s %7.1f us   sum of counters %lld / %d\n", cells,
                   mode == 0 ? "agent scope" : mode == 1 ? "workgroup scope, one copy" : "workgroup scope, copy per XCC_ID", best * 1e3f, tot, n);
        }
        hipFree(d_idx); hipFree(d_cnt); hipFree(d_out);
    }
    return 0;
}
