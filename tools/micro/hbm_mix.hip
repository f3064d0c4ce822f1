// Micro-benchmark (debugging aid, not part of the product): what HBM delivers for the access mixes of k_info_apply - read the lower
// triangle once, write both triangles - against a plain copy: pure reads, pure writes (ordinary / streaming stores), 1 : 2 read : write,
// and 128-byte row segments written at a 48-byte offset (the fused marginalisation shifts rows and columns by 6 doubles).
//   hipcc --offload-arch=gfx950 -O3 -o hbm_mix tools/micro/hbm_mix.hip && ./hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#define NT_LOAD(p) __builtin_nontemporal_load(p)

// n doubles; every thread moves 8-byte elements, consecutive lanes consecutive elements (the apply's pattern: 16 lanes = 128 bytes)
template <int MODE>
__global__ __launch_bounds__(256) void k(const double* __restrict__ src, double* __restrict__ dst, double* __restrict__ dst2, size_t n, int shift)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (MODE == 0) NT_STORE(&dst[i], NT_LOAD(&src[i]));                                   // copy, streaming
        if (MODE == 1) dst[i] = 1.0;                                                          // write only
        if (MODE == 2) NT_STORE(&dst[i], 1.0);                                                // write only, streaming
        if (MODE == 3) { const double v = NT_LOAD(&src[i]); NT_STORE(&dst[i], v); NT_STORE(&dst2[i], v); }      // 1 read : 2 writes
        if (MODE == 4) acc += NT_LOAD(&src[i]);                                               // read only
        if (MODE == 5) {                                                                      // 128-byte segments at a (shift * 8)-byte offset, rows of 256 doubles
            const size_t row = i >> 8, col = i & 255;
            if (col + shift < 256) NT_STORE(&dst[row * 256 + col + shift], 1.0);
        }
        if (MODE == 6) {                                                                      // 1 read : 2 writes, one of them to a transposed 16 x 16 tile position
            const double v = NT_LOAD(&src[i]);
            NT_STORE(&dst[i], v);
            const size_t t = i >> 8, e = i & 255;                                             // tile t, element (r, c) -> (c, r): 16 lanes stay contiguous? no - stride 16
            NT_STORE(&dst2[t * 256 + (e & 15) * 16 + (e >> 4)], v);
        }
    }
    if (MODE == 4 && acc == 123.456) dst[0] = acc;
}

template <int MODE>
void run(const char* name, const double* src, double* dst, double* dst2, size_t n, double bytes_per_elem, int shift = 0)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 16;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, dst2, n, shift);
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, dst2, n, shift);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-44s %7.4f ms  %7.1f GB/s\n", name, ms, bytes_per_elem * n / ms * 1e-6);
}

int main()
{
    const size_t n = (size_t)32 << 20;           // 32 Mi doubles = 256 MiB per buffer
    double *a, *b, *c;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8); hipMemset(c, 0, n * 8);
    run<0>("copy (streaming), 256 MiB", a, b, c, n, 16);
    run<4>("read only", a, b, c, n, 8);
    run<1>("write only, ordinary stores", a, b, c, n, 8);
    run<2>("write only, streaming stores", a, b, c, n, 8);
    run<3>("1 read : 2 writes", a, b, c, n, 24);
    run<3>("1 read : 2 writes, half the elements", a, b, c, n / 2, 24);
    run<5>("write rows, offset 0", a, b, c, n, 8, 0);
    run<5>("write rows, offset 6 doubles (48 B)", a, b, c, n, 8.0 * 250 / 256, 6);
    run<5>("write rows, offset 8 doubles (64 B)", a, b, c, n, 8.0 * 248 / 256, 8);
    run<6>("1 read : 2 writes, one transposed by lanes", a, b, c, n, 24);
    return 0;
}
