// Micro-probe (debugging aid, not part of the product): operand / result lane layout and issue cost of v_mfma_f64_4x4x4_4b_f64
// (four independent 4x4x4 FP64 products per instruction) next to v_mfma_f64_16x16x4_f64 on gfx950.
//   layout: B = 1 in ONE lane, A[lane] = lane + 1  ->  the non-zero result lanes and their values show which A lanes meet that B lane
//   cost:   shader clocks per instruction in a dependent chain and with 4 independent accumulators, one wave per SIMD
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma444.hip -o ingvio_amd/lib/micro_mfma444
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_f __attribute__((ext_vector_type(4)));

__global__ void k_layout(double* out)      // out[64 probes][64 lanes]
{
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        const double a = lane + 1.0, b = lane == p ? 1.0 : 0.0;
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[p * 64 + lane] = d;
    }
}

template <int MODE>
__global__ void k_cost(double* out, long long* cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    const double a = 1.0 + 1e-9 * lane, b = 1e-9;
    double d1[4] = { 0, 0, 0, 0 };
    double4_f d4[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) d1[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1[0], 0, 0, 0);                 // dependent chain
            if (MODE == 1) d1[u & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1[u & 3], 0, 0, 0);         // 4 independent
            if (MODE == 2) d4[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d4[0], 0, 0, 0);
            if (MODE == 3) d4[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d4[u & 3], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = d1[0] + d1[1] + d1[2] + d1[3] + d4[0][0] + d4[1][1] + d4[2][2] + d4[3][3];
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 64 * 64 * 64);
    hipMalloc(&cyc, sizeof(long long) * 8);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
    static double h[64 * 64];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    std::printf("layout of v_mfma_f64_4x4x4_4b_f64: B lane p -> {result lane : A lane}\n");
    for (int p = 0; p < 64; ++p) {
        std::printf("B%2d:", p);
        for (int l = 0; l < 64; ++l) if (h[p * 64 + l] != 0.0) std::printf(" D%d:A%d", l, (int)h[p * 64 + l] - 1);
        std::printf("\n");
    }
    const int iters = 2000;
    const int grid = 256 * 4;                      // one wave per SIMD
    hipLaunchKernelGGL(k_cost<0>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k_cost<1>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k_cost<2>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k_cost<3>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    long long hc[8];
    hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
    const char* names[4] = { "4x4x4 dependent", "4x4x4 4 independent", "16x16x4 dependent", "16x16x4 4 independent" };
    for (int m = 0; m < 4; ++m) std::printf("%-24s %.1f shader clocks per instruction\n", names[m], (double)hc[m] / (iters * 16.0));
    return 0;
}
