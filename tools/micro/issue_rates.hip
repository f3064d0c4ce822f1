// Micro-benchmark (debugging aid, not part of the product): issue cost of the cross-lane / FP64 primitives
// the gate kernel is built from, in shader clocks per instruction per wave, at 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_f __attribute__((ext_vector_type(4)));
#define N 256
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0 + 1e-9 * (lane + i);
    double4_f acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    lds[threadIdx.x] = x[0];
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < N / 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) x[i] = fma(x[i], 1.0000001, 1e-12);                                    // plain FP64 FMA
                if (MODE == 1) {                                                                         // 2 readlane + FMA with SGPR operand
                    const int lo = __builtin_amdgcn_readlane(__double2loint(x[(i + 1) & 7]), 5);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(x[(i + 1) & 7]), 5);
                    x[i] = fma(__hiloint2double(hi, lo), 1e-12, x[i]);
                }
                if (MODE == 2) {                                                                         // 2 ds_bpermute + FMA
                    const int lo = __builtin_amdgcn_ds_bpermute(20, __double2loint(x[(i + 1) & 7]));
                    const int hi = __builtin_amdgcn_ds_bpermute(20, __double2hiint(x[(i + 1) & 7]));
                    x[i] = fma(__hiloint2double(hi, lo), 1e-12, x[i]);
                }
                if (MODE == 3) x[i] = fma(lds[(u * 8 + i) & 1023], 1e-12, x[i]);                         // LDS broadcast read + FMA
                if (MODE == 4) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[(i + 1) & 7], acc[i & 3], 0, 0, 0);
                if (MODE == 8) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[(i + 1) & 7], acc[0], 0, 0, 0);          // ONE dependent chain
                if (MODE == 9) acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[(i + 1) & 7], acc[i & 1], 0, 0, 0);  // two chains
                if (MODE == 7) {                                                                         // 2x 32-bit DPP row_newbcast + FMA
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x[(i + 1) & 7]), 0x150 + 5, 0xf, 0xf, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x[(i + 1) & 7]), 0x150 + 5, 0xf, 0xf, false);
                    x[i] = fma(__hiloint2double(hi, lo), 1e-12, x[i]);
                }
                if (MODE == 5) {                                                                         // DPP quad broadcast + FMA
                    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x[(i + 1) & 7]), 0x55, 0xf, 0xf, true);
                    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x[(i + 1) & 7]), 0x55, 0xf, 0xf, true);
                    x[i] = fma(__hiloint2double(hi, lo), 1e-12, x[i]);
                }
            }
        }
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name)
{
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 256 * 1024 * 16); hipMalloc(&cyc, 8);
    for (int wpb : {64, 256, 1024}) {          // 1 wave per CU / 1 per SIMD / 4 per SIMD (1 block per CU: grid = 256)
        const int iters = 400;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, out, cyc, iters);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wpb), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // wall-clock cost per group per SIMD, assuming 2.4 GHz and waves spread evenly over 4 SIMDs
        const double per_simd = ms * 1e-3 * 2.4e9 / (iters * (double)N) / ((wpb / 64 + 3) / 4);
        printf("%-28s waves/CU=%2d  clock64/group/wave = %7.2f   wall cycles/group/SIMD-wave = %7.2f\n", name, wpb / 64,
               (double)c / (iters * (double)N), per_simd);
    }
}
int main()
{
    run<0>("fma_f64");
    run<1>("2 readlane + fma");
    run<2>("2 ds_bpermute + fma");
    run<3>("ds_read_b64 bcast + fma");
    run<4>("mfma_f64_16x16x4");
    run<8>("mfma_f64_16x16x4, one chain");
    run<9>("mfma_f64_16x16x4, two chains");
    run<5>("2 dpp quad bcast + fma");
    run<7>("2 dpp32 row_newbcast + fma");
    return 0;
}
