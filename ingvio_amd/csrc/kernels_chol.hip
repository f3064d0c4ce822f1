// kernels_chol.hip — batched SPD building blocks that work out of HBM / L2 instead of one workgroup's LDS (gfx950 only).
//
//   k_gemm        FP64 GEMM on v_mfma_f64_16x16x4: one workgroup per 32x32 block of C, its four waves split K and reduce through
//                 LDS (no atomics: deterministic), optional grid-level split of K into partial buffers.
//   k_chol_panel  one 32-column panel of a blocked LEFT-looking Cholesky with carried rows.  Every workgroup owns 16 rows of the
//                 panel: it subtracts the contribution of the finished panels from its rows AND from the 32x32 diagonal block
//                 (redundantly - the operands are the same fragments, 3 extra MFMA per step, no inter-workgroup dependency),
//                 factorises the block with one wave (one lane per row, the identity carried below it so that T = L_d^-T falls
//                 out), and multiplies its rows by T on the matrix cores.  Out of place (X is never written), so a panel is one
//                 launch with no ordering between its workgroups.
//
// What they replace: the single-workgroup 216-pivot Gauss-Jordan of the large-window Kalman solve (kernels_bigwin.hip; reference
// maths StateManager.cpp:399-405) and, through Cholesky-QR, the single-workgroup Householder panel of kernels_qr.hip for tall
// stacks (the thin QR of RemoveLostUpdate.cpp:376-397).
#include "launch_chol.h"
#include "dev_common.h"

typedef double double4_f __attribute__((ext_vector_type(4)));

namespace {

template <int MODE>
__device__ __forceinline__ double ld_op(const double* __restrict__ P, int ld, int i, int k, int I, int K, const double* __restrict__ Px, int ix)
{
    const bool ok = i < I && k < K;
    if (!ok) return 0.0;
    if (Px && i == ix) return Px[k];
    return MODE == 0 ? P[(size_t)i + (size_t)k * ld] : P[(size_t)k + (size_t)i * ld];
}

// ---------------------------------------------------------------------------------------------
// GEMM.  grid = (blocks, ksplit, batch), 256 threads.  The lane's four k of a 16-step are k0 + 4 kq + s (s = 0..3) for BOTH
// operands - any assignment works as long as A and B agree -, so a k-contiguous operand reads 32 consecutive bytes per lane.
// ---------------------------------------------------------------------------------------------
template <int MA, int MB>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g)
{
    __shared__ double sPart[4][4][4][64];                               // [wave][tile][reg][lane]
    const int batch = blockIdx.z;
    if (g.active && !g.active[batch]) return;
    const int nbj = (g.N + 31) / 32;
    int bi, bj;
    if (g.lower) {                                                      // blockIdx.x enumerates bi >= bj
        int t = blockIdx.x; bi = 0;
        while (t >= bi + 1) { t -= bi + 1; ++bi; }
        bj = t;
    } else { bi = blockIdx.x / nbj; bj = blockIdx.x - bi * nbj; }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const double* A = g.A + (size_t)batch * g.sa;
    const double* B = g.B + (size_t)batch * g.sb;
    // K range of this workgroup, then of this wave (multiples of 16)
    const int kchunks = (g.K + 15) / 16;
    const int per_wg = (kchunks + g.ksplit - 1) / g.ksplit;
    const int c_lo = blockIdx.y * per_wg, c_hi = min(kchunks, c_lo + per_wg);
    const int per_wave = (max(0, c_hi - c_lo) + 3) / 4;
    const int w_lo = c_lo + wave * per_wave, w_hi = min(c_hi, w_lo + per_wave);
    const int i0 = bi * 32 + l15, i1 = i0 + 16, j0 = bj * 32 + l15, j1 = j0 + 16;
    double4_f c00 = { 0, 0, 0, 0 }, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll 1
    for (int ch = w_lo; ch < w_hi; ++ch) {
        const int kb = ch * 16 + 4 * kq;
        double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a0[s] = ld_op<MA>(A, g.lda, i0, kb + s, g.M, g.K, g.Ax, g.ax);
            a1[s] = ld_op<MA>(A, g.lda, i1, kb + s, g.M, g.K, g.Ax, g.ax);
            b0[s] = ld_op<MB>(B, g.ldb, j0, kb + s, g.N, g.K, g.Bx, g.bx);
            b1[s] = ld_op<MB>(B, g.ldb, j1, kb + s, g.N, g.K, g.Bx, g.bx);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b1[s], c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b0[s], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], c11, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sPart[wave][0][r][lane] = c00[r]; sPart[wave][1][r][lane] = c01[r];
        sPart[wave][2][r][lane] = c10[r]; sPart[wave][3][r][lane] = c11[r];
    }
    __syncthreads();
    // wave w finishes tile w: (ti, tj) = (w >> 1, w & 1).  Element (row, col) of a tile sits in reg row >> 2 of lane (row & 3) 16 + col.
    const int ti = wave >> 1, tj = wave & 1;
    double* C = g.C + (size_t)batch * g.sc + (size_t)blockIdx.y * g.csplit;
    const bool row_fast = g.rs <= g.cs;                                 // store along the dimension that is contiguous in memory
    const double dadd = g.diag_add_vec ? g.diag_add_vec[batch] : g.diag_add;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = row_fast ? l15 : kq + 4 * q, col = row_fast ? kq + 4 * q : l15;
        const int src_lane = (row & 3) * 16 + col, src_reg = row >> 2;
        double v = sPart[0][wave][src_reg][src_lane] + sPart[1][wave][src_reg][src_lane]
                 + sPart[2][wave][src_reg][src_lane] + sPart[3][wave][src_reg][src_lane];
        const int gi = bi * 32 + ti * 16 + row, gj = bj * 32 + tj * 16 + col;
        if (gi == gj && blockIdx.y == 0) v += dadd;
        if (gi < g.m_lim && gj < g.n_lim) C[(size_t)gi * g.rs + (size_t)gj * g.cs] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// 1 / sqrt(p) to full precision: v_rsq_f64 + two Newton steps
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rsqrt(double p)
{
    double y = __builtin_amdgcn_rsq(p);
    double e = fma(-p * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);
    e = fma(-p * y, y, 1.0);
    y = fma(y * e, 0.5, y);
    return y;
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// Diagonal block of one panel (columns c0 .. c0+31).  grid = batch, 512 threads: the eight waves split K of
// D = X_dd - Y_d Y_d^T (three tiles, both operands the same fragments), one wave factorises [D; I] with a lane per row.
// The column of multipliers travels through LDS (one broadcast read serves two columns); only the NEXT pivot's column takes
// the short way through v_readlane, so the chain pivot -> rsqrt -> scale -> next pivot never waits on LDS.
// Writes L_d into Y's diagonal rows and T = L_d^-T (row-major 32 x 32) into Tb[batch].
// ---------------------------------------------------------------------------------------------
#define CD_NW 4
__global__ __launch_bounds__(64 * CD_NW) void k_chol_diag(CholArgs a, int c0, double* __restrict__ Tb, size_t ts)
{
    __shared__ double sPart[CD_NW][3][4][64];
    __shared__ double sD[64][33];                                       // [D; I]
    __shared__ double sOrig[32];
    __shared__ __attribute__((aligned(16))) double sC[2][32];
    const int batch = blockIdx.x;
    if (a.active && !a.active[batch]) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4, ld = a.ld;
    const double* X = a.X + (size_t)batch * a.xs;
    double* Y = a.Y + (size_t)batch * a.xs;
    double4_f d00 = { 0, 0, 0, 0 }, d10 = d00, d11 = d00;
    {
        const int steps = c0 >> 2, per = (steps + CD_NW - 1) / CD_NW;
        const int s_lo = wave * per, s_hi = min(steps, s_lo + per);
        const double* pb = Y + c0 + l15 + (size_t)kq * ld;
#pragma unroll 2
        for (int s4 = s_lo; s4 < s_hi; ++s4) {
            const double bv = pb[(size_t)(4 * s4) * ld], bw = pb[(size_t)(4 * s4) * ld + 16];
            d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, bv, d00, 0, 0, 0);
            d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(bw, bv, d10, 0, 0, 0);
            d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(bw, bw, d11, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sPart[wave][0][r][lane] = d00[r]; sPart[wave][1][r][lane] = d10[r]; sPart[wave][2][r][lane] = d11[r]; }
    __syncthreads();
    for (int e = tid; e < 1024; e += 64 * CD_NW) {
        const int i = e & 31, j = e >> 5;
        double v = 0.0;
        if (j <= i) {
            const int slot = (i >> 4) + (j >> 4);                         // (0,0) -> 0, (1,0) -> 1, (1,1) -> 2
            const int rr = i & 15, cc = j & 15, sl = (rr & 3) * 16 + cc, sr = rr >> 2;
            const double x = X[(size_t)(c0 + i) + (size_t)(c0 + j) * ld];
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < CD_NW; ++w) sum += sPart[w][slot][sr][sl];
            v = x - sum;
            if (i == j) sOrig[i] = x;
        }
        sD[i][j] = v;
        sD[32 + i][j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    if (wave != 0) return;
    double d[32];
    const int row = lane & 31;
#pragma unroll
    for (int j = 0; j < 32; ++j) d[j] = sD[lane][j];
    int bad = 0;
    const double crel = a.clamp ? a.clamp_rel : 0.0;
    // Software pipeline: step j runs the chain (pivot j -> rsqrt -> scale column j -> column j+1 through v_readlane) and, behind
    // it, the bulk update with column j-1, whose broadcast reads were issued at the top of the step.
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        double lc[32];
        if (j >= 1) {
#pragma unroll
            for (int c = j + 1; c < 32; ++c) lc[c] = sC[(j - 1) & 1][c];
        }
        asm volatile("" ::: "memory");
        const double p = readlane_f64(d[j], j);
        const bool ok = p > crel * sOrig[j];
        if (!ok && !a.clamp) bad = 1;
        const double rs = ok ? fast_rsqrt(p) : 0.0;
        d[j] *= rs;
        if (j < 31) {
            const double ln = readlane_f64(d[j], j + 1);
            d[j + 1] = fma(-d[j], ln, d[j + 1]);
            if (lane < 32) sC[j & 1][lane] = d[j];
        }
        asm volatile("" ::: "memory");
        if (j >= 1) {
#pragma unroll
            for (int c = j + 1; c < 32; ++c) {
                d[c] = fma(-d[j - 1], lc[c], d[c]);
                asm volatile("" : "+v"(d[c]));          // pin the update here: left alone, the scheduler sinks every column's
            }                                            // updates to its pivot step (a 30-deep chain, all broadcasts live)
        }
    }
    double* T = Tb + (size_t)batch * ts;
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 32; ++j) Y[(size_t)(c0 + row) + (size_t)(c0 + j) * ld] = j <= row ? d[j] : 0.0;
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) T[row * 32 + j] = d[j];
    }
    if (bad && lane == 0 && a.status) atomicOr(&a.status[batch], a.fail_bit);
}

// ---------------------------------------------------------------------------------------------
// Rows below the diagonal block of one panel: Y[r, panel] = (X[r, panel] - Y[r, :c0] Y[panel, :c0]^T) T.
// grid = ((rows - c0 - 32) / 16, batch), 256 threads: the four waves split K, reduce through LDS, two waves apply T.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_panel(CholArgs a, int c0, const double* __restrict__ Tb, size_t ts)
{
    __shared__ double sPart[4][2][4][64];
    __shared__ double sU[16][33];
    __shared__ double sT[32][33];
    const int batch = blockIdx.y;
    if (a.active && !a.active[batch]) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int r0 = c0 + 32 + 16 * blockIdx.x, ld = a.ld;
    const double* X = a.X + (size_t)batch * a.xs;
    double* Y = a.Y + (size_t)batch * a.xs;
    double4_f u0 = { 0, 0, 0, 0 }, u1 = u0;
    {
        const int kw = c0 >> 2, k_lo = wave * kw, k_hi = k_lo + kw;       // c0 is a multiple of 32: quarters are multiples of 8
        const double* pa = Y + r0 + l15 + (size_t)kq * ld;
        const double* pb = Y + c0 + l15 + (size_t)kq * ld;
#pragma unroll 1
        for (int k = k_lo; k < k_hi; k += 8) {
            const double av0 = pa[(size_t)k * ld], bv0 = pb[(size_t)k * ld], bw0 = pb[(size_t)k * ld + 16];
            const double av1 = pa[(size_t)(k + 4) * ld], bv1 = pb[(size_t)(k + 4) * ld], bw1 = pb[(size_t)(k + 4) * ld + 16];
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, bv0, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av0, bw0, u1, 0, 0, 0);
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, bv1, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av1, bw1, u1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { sPart[wave][0][r][lane] = u0[r]; sPart[wave][1][r][lane] = u1[r]; }
    {
        const double* T = Tb + (size_t)batch * ts;
        for (int e = tid; e < 1024; e += 256) sT[e >> 5][e & 31] = T[e];
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {
        const int i = e & 15, j = e >> 4;
        const int slot = j >> 4, cc = j & 15, sl = (i & 3) * 16 + cc, sr = i >> 2;
        const double x = X[(size_t)(r0 + i) + (size_t)(c0 + j) * ld];
        sU[i][j] = x - (sPart[0][slot][sr][sl] + sPart[1][slot][sr][sl] + sPart[2][slot][sr][sl] + sPart[3][slot][sr][sl]);
    }
    __syncthreads();
    if (wave < 2) {
        double4_f acc = { 0, 0, 0, 0 };
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sU[l15][4 * k4 + kq], sT[4 * k4 + kq][16 * wave + l15], acc, 0, 0, 0);
        double* so = &sPart[wave][0][0][0];                                 // through LDS so that the store runs along the rows
#pragma unroll
        for (int r = 0; r < 4; ++r) so[(kq + 4 * r) * 17 + l15] = acc[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = kq + 4 * q;
            Y[(size_t)(r0 + l15) + (size_t)(c0 + 16 * wave + col) * ld] = so[l15 * 17 + col];
        }
    }
}

}  // namespace

void launch_gemm(const GemmArgs& g, hipStream_t st)
{
    const int nbi = (g.M + 31) / 32, nbj = (g.N + 31) / 32;
    const int blocks = g.lower ? nbi * (nbi + 1) / 2 : nbi * nbj;
    const dim3 grid(blocks, g.ksplit < 1 ? 1 : g.ksplit, g.batch);
    GemmArgs h = g;
    if (h.ksplit < 1) h.ksplit = 1;
    if (g.modeA == 0 && g.modeB == 0) hipLaunchKernelGGL((k_gemm<0, 0>), grid, dim3(256), 0, st, h);
    else if (g.modeA == 0 && g.modeB == 1) hipLaunchKernelGGL((k_gemm<0, 1>), grid, dim3(256), 0, st, h);
    else if (g.modeA == 1 && g.modeB == 0) hipLaunchKernelGGL((k_gemm<1, 0>), grid, dim3(256), 0, st, h);
    else hipLaunchKernelGGL((k_gemm<1, 1>), grid, dim3(256), 0, st, h);
}

void launch_chol_sweep(const CholArgs& a, hipStream_t st)
{
    for (int c0 = 0; c0 < a.ncols; c0 += 32) {
        hipLaunchKernelGGL(k_chol_diag, dim3(a.batch), dim3(64 * CD_NW), 0, st, a, c0, a.Tb, a.ts);
        if (a.rows > c0 + 32)
            hipLaunchKernelGGL(k_chol_panel, dim3((a.rows - c0 - 32) / 16, a.batch), dim3(256), 0, st, a, c0, a.Tb, a.ts);
    }
}
