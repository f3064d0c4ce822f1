// Micro-probe (debugging aid): relative error of v_rcp_f64 and of one / two Newton steps on it (fast_rcp in dev_common.h uses two).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/rcp_f64.hip -o ingvio_amd/lib/micro_rcp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> hx(n), h0(n), h1(n), h2(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> mant(1.0, 2.0);
    std::uniform_int_distribution<int> ex(-300, 300);
    for (int i = 0; i < n; ++i) hx[i] = std::ldexp(mant(g), ex(g)) * ((i & 1) ? -1.0 : 1.0);
    double *x, *r0, *r1, *r2;
    hipMalloc(&x, 8 * n); hipMalloc(&r0, 8 * n); hipMalloc(&r1, 8 * n); hipMalloc(&r2, 8 * n);
    hipMemcpy(x, hx.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, r0, r1, r2, n);
    hipMemcpy(h0.data(), r0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), r1, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), r2, 8 * n, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)hx[i];
        e0 = std::fmax(e0, (double)fabsl(((long double)h0[i] - t) / t));
        e1 = std::fmax(e1, (double)fabsl(((long double)h1[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)h2[i] - t) / t));
    }
    std::printf("max relative error over %d doubles: v_rcp_f64 %.3e | + 1 Newton %.3e | + 2 Newton %.3e   (2^-53 = 1.11e-16)\n", n, e0, e1, e2);
    return 0;
}
