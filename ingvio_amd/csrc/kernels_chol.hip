// kernels_chol.hip — batched SPD building blocks that work out of HBM / L2 instead of one workgroup's LDS (gfx950 only).
//
//   k_gemm        FP64 GEMM on v_mfma_f64_16x16x4: one workgroup per 32x32 block of C, its four waves split K and reduce through
//                 LDS (no atomics: deterministic), optional grid-level split of K into partial buffers.
//   k_chol_step   one 32-column panel of a blocked RIGHT-looking Cholesky with carried rows: one launch per panel, one
//                 workgroup per 32 x 32 block of the trailing matrix, every workgroup a single memory round trip deep; the
//                 workgroup that updates the next diagonal block factorises it on the spot with one wave (one lane per row, the
//                 identity carried below it so that T = L_d^-T falls out).  A left-looking version (one K-loop per panel)
//                 measured 23 us per panel against 8 here: its loads come from the other XCDs' writes, ~1 us per dependent trip.
//
// What they replace: the single-workgroup 216-pivot Gauss-Jordan of the large-window Kalman solve (kernels_bigwin.hip; reference
// maths StateManager.cpp:399-405) and, through Cholesky-QR, the single-workgroup Householder panel of kernels_qr.hip for tall
// stacks (the thin QR of RemoveLostUpdate.cpp:376-397).
#include "launch_chol.h"
#include "dev_common.h"
#include "block64.h"

typedef double double4_f __attribute__((ext_vector_type(4)));

namespace {

// One operand element, zero outside [0, I) x [0, K).  The load itself is UNCONDITIONAL, from an index clamped into the operand, and
// zeroed by a factor: as `ok ? P[..] : 0.0` every element became its own exec-masked basic block, the compiler lost count of the
// loads in flight across the branches and drained the queue (s_waitcnt vmcnt(0)) before every use - the next chunk's prefetch under
// this chunk's MFMAs did not exist (round 5; the same trap as block64.h).  Operands are finite inside their range, so 0 * x is 0.
// XCD-aware decode of a 1-D grid of 8 * ceil(nbatch / 8) * per workgroups (workgroup w runs on XCD w % 8): XCD x takes the batch
// elements x, x + 8, ... one after the other, all `per` blocks of an element on the same XCD (its L2 then serves the element's
// operands to every block).  false: beyond the batch.
__device__ __forceinline__ bool xcd_decode(int per, int nbatch, int& batch, int& blk)
{
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    batch = xcd + 8 * (slot / per);
    blk = slot % per;
    return batch < nbatch;
}
static inline unsigned xcd_grid(int per, int nbatch) { return 8u * (unsigned)((nbatch + 7) / 8) * (unsigned)per; }

template <int MODE>
__device__ __forceinline__ double ld_op(const double* __restrict__ P, int ld, int i, int k, int I, int K, const double* __restrict__ Px, int ix)
{
    const double on = (i < I && k < K) ? 1.0 : 0.0;
    const int ic = min(i, I - 1), kc = min(k, K - 1);
    const double* p = MODE == 0 ? P + ((size_t)ic + (size_t)kc * ld) : P + ((size_t)kc + (size_t)ic * ld);
    p = (Px && i == ix) ? Px + kc : p;
    return on * *p;
}

// ---------------------------------------------------------------------------------------------
// GEMM.  grid = (blocks, ksplit, batch), 256 threads.  The lane's four k of a 16-step are k0 + 4 kq + s (s = 0..3) for BOTH
// operands - any assignment works as long as A and B agree -, so a k-contiguous operand reads 32 consecutive bytes per lane.
// ---------------------------------------------------------------------------------------------
template <int MA, int MB>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g)
{
    __shared__ double sPart[4][4][4][64];                               // [wave][tile][reg][lane]
    // XCD-aware order (round 5; workgroup w runs on XCD w % 8): all blocks of ONE batch element on the same XCD, one element after the
    // other - its operands (0.3-0.9 MB for the large-window solve) then come from that XCD's 4 MB L2 once, instead of once per 32 x 32
    // block from MALL / HBM: the three GEMMs of the solve moved 389 MB per 32 filters for 13 MB of operands (FETCH_SIZE, r05 PMC)
    const int nbj = (g.N + 31) / 32, nbi = (g.M + 31) / 32;
    const int blocks = g.lower ? nbi * (nbi + 1) / 2 : nbi * nbj, per = blocks * g.ksplit;
    int batch, rem;
    if (!xcd_decode(per, g.batch, batch, rem)) return;
    const int ky = rem / blocks, bx = rem - ky * blocks;
    if (g.active && !g.active[batch]) return;
    int bi, bj;
    if (g.lower) {                                                      // bx enumerates bi >= bj
        int t = bx; bi = 0;
        while (t >= bi + 1) { t -= bi + 1; ++bi; }
        bj = t;
    } else { bi = bx / nbj; bj = bx - bi * nbj; }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const double* A = g.A + (size_t)batch * g.sa + (g.a_sel ? (size_t)g.a_sel[batch] * g.a_sel_stride : 0);
    const double* B = g.B + (size_t)batch * g.sb;
    // K range of this workgroup, then of this wave (multiples of 16)
    const int kchunks = (g.K + 15) / 16;
    const int c_first = min(kchunks, g.k_from == 1 ? 2 * bi : (g.k_from == 2 ? 2 * bj : 0));      // triangular operand: the chunks before the block's first row / column are zeros
    const int per_wg = (kchunks - c_first + g.ksplit - 1) / g.ksplit;
    const int c_lo = c_first + ky * per_wg, c_hi = min(kchunks, c_lo + per_wg);
    const int per_wave = (max(0, c_hi - c_lo) + 3) / 4;
    const int w_lo = c_lo + wave * per_wave, w_hi = min(c_hi, w_lo + per_wave);
    const int i0 = bi * 32 + l15, i1 = i0 + 16, j0 = bj * 32 + l15, j1 = j0 + 16;
    double4_f c00 = { 0, 0, 0, 0 }, c01 = c00, c10 = c00, c11 = c00;
    double a0[4], a1[4], b0[4], b1[4], na0[4], na1[4], nb0[4], nb1[4];
    auto fetch = [&](int ch, double* x0, double* x1, double* y0, double* y1) {
        const int kb = ch * 16 + 4 * kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x0[s] = ld_op<MA>(A, g.lda, i0, kb + s, g.M, g.K, g.Ax, g.ax);
            x1[s] = ld_op<MA>(A, g.lda, i1, kb + s, g.M, g.K, g.Ax, g.ax);
            y0[s] = ld_op<MB>(B, g.ldb, j0, kb + s, g.N, g.K, g.Bx, g.bx);
            y1[s] = ld_op<MB>(B, g.ldb, j1, kb + s, g.N, g.K, g.Bx, g.bx);
        }
    };
    if (w_lo < w_hi) fetch(w_lo, a0, a1, b0, b1);
#pragma unroll 1
    for (int ch = w_lo; ch < w_hi; ++ch) {
        if (ch + 1 < w_hi) fetch(ch + 1, na0, na1, nb0, nb1);              // the next chunk's operands travel under these MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b1[s], c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b0[s], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], c11, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) { a0[s] = na0[s]; a1[s] = na1[s]; b0[s] = nb0[s]; b1[s] = nb1[s]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sPart[wave][0][r][lane] = c00[r]; sPart[wave][1][r][lane] = c01[r];
        sPart[wave][2][r][lane] = c10[r]; sPart[wave][3][r][lane] = c11[r];
    }
    __syncthreads();
    // wave w finishes tile w: (ti, tj) = (w >> 1, w & 1).  Element (row, col) of a tile sits in reg row >> 2 of lane (row & 3) 16 + col.
    const int ti = wave >> 1, tj = wave & 1;
    double* C = g.C + (size_t)batch * g.sc + (size_t)ky * g.csplit;
    const bool row_fast = g.rs <= g.cs;                                 // store along the dimension that is contiguous in memory
    const double dadd = g.diag_add_vec ? g.diag_add_vec[batch] : g.diag_add;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = row_fast ? l15 : kq + 4 * q, col = row_fast ? kq + 4 * q : l15;
        const int src_lane = (row & 3) * 16 + col, src_reg = row >> 2;
        double v = sPart[0][wave][src_reg][src_lane] + sPart[1][wave][src_reg][src_lane]
                 + sPart[2][wave][src_reg][src_lane] + sPart[3][wave][src_reg][src_lane];
        const int gi = bi * 32 + ti * 16 + row, gj = bj * 32 + tj * 16 + col;
        if (gi == gj && ky == 0) v += dadd;
        if (g.Cx && gj == g.cx_col) { if (gi < g.m_lim) g.Cx[(size_t)batch * g.scx + gi] = v; }
        else if (gi < g.m_lim && gj < g.n_lim) C[(size_t)gi * g.rs + (size_t)gj * g.cs] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// The same product on 64 x 64 blocks with the operand panels staged through LDS two chunks ahead (block64_mma2): for products that
// are large enough to have a K loop worth pipelining (the three GEMMs of the large-window solve: 192^3 per filter).  k_gemm gives
// every 32 x 32 block its own pass over two full-K panels, four waves deep in K with a reduction through LDS: 42 workgroups per filter
// and product, each three dependent memory round trips long; here 12 workgroups per filter walk K once.  ksplit must be 1.
// MA / MB as in k_gemm; an operand that is k-contiguous in memory (mode 1) is staged by threads that take 4 consecutive k.
// ---------------------------------------------------------------------------------------------
template <int MA, int MB>
__global__ __launch_bounds__(256) void k_gemm64(GemmArgs g)
{
    __shared__ union { Block64Lds2 ab; double sV[4][32][33]; } sh;
    const int nbj = (g.N + 63) / 64, nbi = (g.M + 63) / 64;
    const int blocks = g.lower ? nbi * (nbi + 1) / 2 : nbi * nbj;
    int batch, bx;
    if (!xcd_decode(blocks, g.batch, batch, bx)) return;
    if (g.active && !g.active[batch]) return;
    int bi, bj;
    if (g.lower) {
        int t = bx; bi = 0;
        while (t >= bi + 1) { t -= bi + 1; ++bi; }
        bj = t;
    } else { bi = bx / nbj; bj = bx - bi * nbj; }
    const double* A = g.A + (size_t)batch * g.sa + (g.a_sel ? (size_t)g.a_sel[batch] * g.a_sel_stride : 0);
    const double* B = g.B + (size_t)batch * g.sb;
    b64_d4 c[4];
    const int k16 = (g.K + 15) & ~15;
    const int ks = min(k16, g.k_from == 1 ? 64 * bi : (g.k_from == 2 ? 64 * bj : 0));                // triangular operand (GemmArgs::k_from): K starts at the block
    block64_mma2<MA == 1, MB == 1>(sh.ab, k16 - ks,
                                   [&](int r, int k) { return ld_op<MA>(A, g.lda, 64 * bi + r, ks + k, g.M, g.K, g.Ax, g.ax); },
                                   [&](int r, int k) { return ld_op<MB>(B, g.ldb, 64 * bj + r, ks + k, g.N, g.K, g.Bx, g.bx); }, true, c);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wi = wave >> 1, wj = wave & 1;
    block64_to_lds(c, sh.sV[wave]);                                     // the wave's 32 x 32 quadrant, then stores along the contiguous dimension
    double* C = g.C + (size_t)batch * g.sc;
    const bool row_fast = g.rs <= g.cs;
    const double dadd = g.diag_add_vec ? g.diag_add_vec[batch] : g.diag_add;
    const int r0 = 64 * bi + 32 * wi, c0 = 64 * bj + 32 * wj;
#pragma unroll 4
    for (int e = lane; e < 1024; e += 64) {
        const int rr = row_fast ? e & 31 : e >> 5, cc = row_fast ? e >> 5 : e & 31;
        const int gi = r0 + rr, gj = c0 + cc;
        double v = sh.sV[wave][rr][cc];
        if (gi == gj) v += dadd;
        if (g.lower && (gj >> 5) > (gi >> 5)) continue;                 // what k_gemm's lower mode writes: the 32 x 32 blocks on and below the diagonal
        if (g.Cx && gj == g.cx_col) { if (gi < g.m_lim) g.Cx[(size_t)batch * g.scx + gi] = v; }
        else if (gi < g.m_lim && gj < g.n_lim) C[(size_t)gi * g.rs + (size_t)gj * g.cs] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Gram matrix of a tall matrix, lower 64 x 64 blocks: G = [H | r]^T [H | r], H m x n column-major (ld), r [m] (column n).
// grid = (blocks bi >= bj, ksplit): partial p of the K split goes to part + p * pstride (column-major n_ld x n_ld).
// Per 32-row chunk the workgroup stages its two 64-column panels in LDS (each thread 8 consecutive rows of one column: 256
// contiguous bytes per column and wave), the next chunk's loads in flight under the 32 MFMA per wave of the current one.
// ---------------------------------------------------------------------------------------------
#define GR_S 36                                                         // LDS row stride (doubles) of a staged column
__global__ __launch_bounds__(256) void k_gram_tn(const double* __restrict__ H, int ldh, const double* __restrict__ rv, int m, int n,
                                                 double* __restrict__ part, size_t pstride, int n_ld, int ksplit)
{
    __shared__ __attribute__((aligned(16))) double sA[64][GR_S];
    __shared__ __attribute__((aligned(16))) double sB[64][GR_S];
    int t = blockIdx.x, bi = 0;
    while (t >= bi + 1) { t -= bi + 1; ++bi; }
    const int bj = t;
    const bool diag = bi == bj;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int n1 = n + 1;
    const int chunks = (m + 31) / 32, per = (chunks + ksplit - 1) / ksplit;
    const int c_lo = blockIdx.y * per, c_hi = min(chunks, c_lo + per);
    const int col = tid >> 2, kseg = (tid & 3) * 8;                      // staging role: column of the panel, 8 rows of the chunk
    const int ca = 64 * bi + col, cb = 64 * bj + col;
    const double* pa = ca < n ? H + (size_t)ca * ldh : (ca == n ? rv : nullptr);
    const double* pb = cb < n ? H + (size_t)cb * ldh : (cb == n ? rv : nullptr);
    double ra[8], rb[8];
    auto fetch = [&](int ch) {
        const int k0 = 32 * ch + kseg;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool on = k0 + u < m;
            ra[u] = (pa && on) ? pa[k0 + u] : 0.0;
            rb[u] = (!diag && pb && on) ? pb[k0 + u] : 0.0;
        }
    };
    const int wi = wave >> 1, wj = wave & 1;                             // this wave's 32 x 32 quadrant
    double4_f c00 = { 0, 0, 0, 0 }, c01 = c00, c10 = c00, c11 = c00;
    if (c_lo < c_hi) fetch(c_lo);
#pragma unroll 1
    for (int ch = c_lo; ch < c_hi; ++ch) {
        lds_barrier();                                                 // the previous chunk's fragments have been read
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            *reinterpret_cast<double2*>(&sA[col][kseg + u]) = make_double2(ra[u], ra[u + 1]);
            if (!diag) *reinterpret_cast<double2*>(&sB[col][kseg + u]) = make_double2(rb[u], rb[u + 1]);
        }
        lds_barrier();
        if (ch + 1 < c_hi) fetch(ch + 1);
        const double (*sBB)[GR_S] = diag ? sA : sB;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const double a0 = sA[32 * wi + l15][4 * k4 + kq], a1 = sA[32 * wi + 16 + l15][4 * k4 + kq];
            const double b0 = sBB[32 * wj + l15][4 * k4 + kq], b1 = sBB[32 * wj + 16 + l15][4 * k4 + kq];
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
        }
    }
    double* C = part + (size_t)blockIdx.y * pstride;
    const int gi0 = 64 * bi + 32 * wi, gj0 = 64 * bj + 32 * wj;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = gi0 + kq + 4 * r, j = gj0 + l15;
        if (i < n1 && j < n1) C[(size_t)i + (size_t)j * n_ld] = c00[r];
        if (i < n1 && j + 16 < n1) C[(size_t)i + (size_t)(j + 16) * n_ld] = c01[r];
        if (i + 16 < n1 && j < n1) C[(size_t)(i + 16) + (size_t)j * n_ld] = c10[r];
        if (i + 16 < n1 && j + 16 < n1) C[(size_t)(i + 16) + (size_t)(j + 16) * n_ld] = c11[r];
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
// Factorisation of one 32 x 32 diagonal block by ONE wave: lane l < 32 holds row l of the block, lanes 32..63 the rows of an
// identity carried below it, so that they end up as T = L_d^-T.  sDI = [D; I] (64 x 33), floorv = this lane's pivot floor
// (lane j: pivot j).  Elimination in LDL^T form, scaled to Cholesky at the end: the column that is broadcast is the UNSCALED
// one, so it leaves for LDS before the pivot's reciprocal exists, and the chain pivot j -> pivot j+1 is rcp + mul + fma +
// v_readlane.  Software pipeline: step j runs that chain and, behind it, the bulk update with column j-1 (its broadcast reads
// are issued at the top of the step).  Returns through d[] the scaled rows; `bad`: a pivot of THIS lane's row was rejected.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void factor32(double (*sDI)[33], double (*sC)[32], int lane, double floorv, double (&d)[32], bool& bad)
{
#pragma unroll
    for (int j = 0; j < 32; ++j) d[j] = sDI[lane][j];
    double m_prev = 0.0, pv = 1.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        double lc[32];
        if (j >= 1) {
#pragma unroll
            for (int c = j + 1; c < 32; ++c) lc[c] = sC[(j - 1) & 1][c];
        }
        if (lane < 32) sC[j & 1][lane] = d[j];
        asm volatile("" ::: "memory");
        const double p = readlane_f64(d[j], j);
        const bool ok = p > readlane_f64(floorv, j);
        const double rinv = ok ? fast_rcp(p) : 0.0;
        const double m = d[j] * rinv;
        if (lane == j) pv = d[j];
        if (j < 31) {
            const double x = readlane_f64(d[j], j + 1);
            d[j + 1] = fma(-m, x, d[j + 1]);
        }
        asm volatile("" ::: "memory");
        if (j >= 1) {
#pragma unroll
            for (int c = j + 1; c < 32; ++c) {
                d[c] = fma(-m_prev, lc[c], d[c]);
                asm volatile("" : "+v"(d[c]));          // pin the update here: left alone, the scheduler sinks every column's
            }                                            // updates to its pivot step (a 30-deep chain, all broadcasts live)
        }
        m_prev = m;
    }
    // 1 / sqrt(pivot) per column (0 for a rejected pivot), broadcast, and the final column scaling
    const bool okp = pv > floorv;
    if (lane < 32) sC[0][lane] = okp ? fast_rsqrt(pv) : 0.0;
    asm volatile("" ::: "memory");
    bad = lane < 32 && !okp;
#pragma unroll
    for (int j = 0; j < 32; ++j) d[j] *= sC[0][j];
}

// stores of a factorised diagonal block: L_d into Lout's block (upper part zero), T (row-major 32 x 32) into Tout
__device__ __forceinline__ void store_factor(const double (&d)[32], int lane, double* __restrict__ Ld, int ld, double* __restrict__ Tout)
{
    const int row = lane & 31;
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 32; ++j) Ld[(size_t)row + (size_t)j * ld] = j <= row ? d[j] : 0.0;
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) Tout[row * 32 + j] = d[j];
    }
}

// Resources (round 6).  On the side stream of the large-window solve these kernels run UNDER the gate (k_feat_gate4_big: 2-wave workgroups of
// 232 VGPRs and 40.1 KB, four per CU = the whole register file and LDS).  Tried: [Y_i; Y_j] in the panel blocks' LDS (25.9 instead of 42.7 KB,
// one more barrier) and a cap of 128 VGPRs, so that a step's workgroup fits what ONE retiring gate workgroup leaves behind - the first step
// (2016 workgroups) still ends with the gate, 220 us after it was issued (rocprofv3 timeline), and the cap spills 26 registers in factor32
// (config 5: 1.011 -> 1.032 ms per step).  Whatever arbitrates between the two queues, it is not the workgroup's footprint.  The smaller
// LDS stays (it costs nothing), the register cap does not (CHOL_STEP_WPE 4 rebuilds it).
#ifndef CHOL_STEP_WPE
#define CHOL_STEP_WPE 3
#endif
// ---------------------------------------------------------------------------------------------
// First diagonal block + the original diagonal (the reference of the pivot floor).  grid = batch, 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, CHOL_STEP_WPE) void k_chol_first(CholArgs a)
{
    __shared__ double sDI[64][33];
    __shared__ __attribute__((aligned(16))) double sC[2][32];
    const int batch = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (a.active && !a.active[batch]) return;
    double* W = a.W + (size_t)batch * a.xs;
    double* Y = a.Y + (size_t)batch * a.xs;
    double* od = a.Tb + (size_t)batch * a.ts + (size_t)a.t_slots * 1024;
    for (int c = tid; c < a.ncols; c += 256) od[c] = W[(size_t)c + (size_t)c * a.ld];
    for (int e = tid; e < 1024; e += 256) {
        const int i = e & 31, j = e >> 5;
        sDI[i][j] = W[(size_t)i + (size_t)j * a.ld];
        sDI[32 + i][j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    if (tid >= 64) return;
    const double floorv = a.clamp ? a.clamp_rel * sDI[lane & 31][lane & 31] : 0.0;
    double d[32]; bool bad;
    factor32(sDI, sC, lane, floorv, d, bad);
    store_factor(d, lane, Y, a.ld, a.Tb + (size_t)batch * a.ts);
    if (__any(bad && !a.clamp) && lane == 0 && a.status) atomicOr(&a.status[batch], a.fail_bit);
}

// ---------------------------------------------------------------------------------------------
// One step of the RIGHT-looking sweep: panel k (32 columns) is scaled with T_k and the whole trailing part is updated, every
// workgroup one 32 x 32 block and one memory round trip deep:
//     Y_i = W[i, k] T_k,  Y_j = W[j, k] T_k   (both recomputed by every workgroup that needs them: 2 x 32 MFMA, no ordering)
//     W[i, j] -= Y_i Y_j^T                     (i >= j > k; block rows beyond the columns are the carried rows)
// The workgroups of column j = k+1 also store Y_i into the output; the workgroup of the next diagonal block (k+1, k+1)
// factorises it on the spot (one wave, factor32) and leaves T_k+1 for the next launch.  The last panel has no trailing blocks:
// its launch only scales the rows below.  grid = (blocks, batch), 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, CHOL_STEP_WPE) void k_chol_step(CholArgs a, int k, int per)
{
    __shared__ double sT[32][33];
    __shared__ double sUD[64][33];                                       // the panel blocks U_i (rows 0..31), U_j (32..63); then Y_i, Y_j; later [D; I]
    double (*sUi)[33] = sUD;
    double (*sUj)[33] = sUD + 32;
    double (*sDI)[33] = sUD;
    __shared__ __attribute__((aligned(16))) double sC[2][32];
    int batch, blk;
    if (!xcd_decode(per, a.batch, batch, blk)) return;
    if (a.active && !a.active[batch]) return;
    const int nbr = a.rows >> 5, ncb = a.ncols >> 5, ld = a.ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    // block of this workgroup: column blocks j = k+1 .. ncb-1, rows i = j .. nbr-1; or (last panel) rows only
    int i, j;
    const bool tail_only = k + 1 >= ncb;
    if (tail_only) { i = k + 1 + blk; j = -1; }
    else {
        int t = blk; j = k + 1;
        while (t >= nbr - j) { t -= nbr - j; ++j; }
        i = j + t;
    }
    double* W = a.W + (size_t)batch * a.xs;
    double* Y = a.Y + (size_t)batch * a.xs;
    double* Tb = a.Tb + (size_t)batch * a.ts;
    const double* Tk = Tb + (size_t)(k % a.t_slots) * 1024;
    const double* Wi = W + (size_t)32 * i + (size_t)(32 * k) * ld;
    const double* Wj = W + (size_t)32 * (j < 0 ? i : j) + (size_t)(32 * k) * ld;
    const bool diag = i == j;
    // one round trip: T_k, the two panel blocks, the trailing block (tile layout: wave w = tile (w >> 1, w & 1))
    for (int e = tid; e < 1024; e += 256) {
        const int r = e & 31, c = e >> 5;
        sT[c][r] = Tk[e];                                                // Tk row-major: e = row * 32 + col -> sT[row][col]
        sUi[r][c] = Wi[(size_t)r + (size_t)c * ld];
        if (!diag && !tail_only) sUj[r][c] = Wj[(size_t)r + (size_t)c * ld];
    }
    const int ti = wave >> 1, tj = wave & 1;
    double4_f cacc = { 0, 0, 0, 0 };
    double* Cij = tail_only ? nullptr : W + (size_t)(32 * i + 16 * ti) + (size_t)(32 * j + 16 * tj) * ld;
    if (!tail_only) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cacc[r] = Cij[(size_t)(kq + 4 * r) + (size_t)l15 * ld];
    }
    lds_barrier();
    // Y_i (and Y_j) on the matrix cores: tile (ti, tj) of U T
    {
        double4_f yi = { 0, 0, 0, 0 }, yj = { 0, 0, 0, 0 };
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const double tb = sT[4 * k4 + kq][16 * tj + l15];
            yi = __builtin_amdgcn_mfma_f64_16x16x4f64(sUi[16 * ti + l15][4 * k4 + kq], tb, yi, 0, 0, 0);
            if (!diag && !tail_only) yj = __builtin_amdgcn_mfma_f64_16x16x4f64(sUj[16 * ti + l15][4 * k4 + kq], tb, yj, 0, 0, 0);
        }
        lds_barrier();                                                   // every wave has read its U fragments: Y takes their place
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sDI[16 * ti + kq + 4 * r][16 * tj + l15] = yi[r];
            sDI[32 + 16 * ti + kq + 4 * r][16 * tj + l15] = diag ? yi[r] : yj[r];
        }
    }
    lds_barrier();
    if (tail_only || j == k + 1) {                                       // this workgroup owns the output of block row i
        double* Yo = Y + (size_t)32 * i + (size_t)(32 * k) * ld;
        for (int e = tid; e < 1024; e += 256) { const int r = e & 31, c = e >> 5; Yo[(size_t)r + (size_t)c * ld] = sDI[r][c]; }
        if (a.Y2 && i >= ncb) {
            double* Y2o = a.Y2 + (size_t)batch * a.y2s + (size_t)(a.y2_row0 + 32 * (i - ncb)) + (size_t)(32 * k) * a.ld_y2;
            for (int e = tid; e < 1024; e += 256) { const int r = e & 31, c = e >> 5; Y2o[(size_t)r + (size_t)c * a.ld_y2] = sDI[r][c]; }
        }
    }
    if (tail_only) return;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4)
        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sDI[16 * ti + l15][4 * k4 + kq], sDI[32 + 16 * tj + l15][4 * k4 + kq], cacc, 0, 0, 0);
    const bool next_diag = diag && j == k + 1;
    if (!next_diag) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Cij[(size_t)(kq + 4 * r) + (size_t)l15 * ld] = cacc[r];
        return;
    }
    // the next diagonal block: factorise it now (its updated value is final), T_k+1 for the next launch
    lds_barrier();                                                     // every wave is done reading Y_i / Y_j from sDI
#pragma unroll
    for (int r = 0; r < 4; ++r) sDI[16 * ti + kq + 4 * r][16 * tj + l15] = cacc[r];
    for (int e = tid; e < 1024; e += 256) { const int r = e & 31, c = e >> 5; sDI[32 + r][c] = r == c ? 1.0 : 0.0; }
    lds_barrier();
    if (wave != 0) return;
    const double* od = Tb + (size_t)a.t_slots * 1024;
    const double floorv = a.clamp ? a.clamp_rel * od[32 * j + (lane & 31)] : 0.0;
    double d[32]; bool bad;
    factor32(sDI, sC, lane, floorv, d, bad);
    store_factor(d, lane, Y + (size_t)32 * j + (size_t)(32 * j) * ld, ld, Tb + (size_t)((k + 1) % a.t_slots) * 1024);
    if (__any(bad && !a.clamp) && lane == 0 && a.status) atomicOr(&a.status[batch], a.fail_bit);
}

// ---------------------------------------------------------------------------------------------
// Carried rows after the factorisation (split sweep): one workgroup per 32-row block i >= ncols / 32 walks the panels itself,
//     Y_i[:, k] = (W_i[:, k] - sum_{p<k} Y_i[:, p] L[k, p]^T) T_k,
// its row block resident in LDS, L and the T_k of all panels read from L2.  The right-looking step kernel would re-read and
// re-write the block's trailing part once per panel (10 MB per filter for a 224-column S with 288 carried rows; here 1.3 MB).
// grid = (carried block rows, batch), 256 threads; needs a.t_slots >= ncols / 32 (every T_k kept).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_carried(CholArgs a, int per)
{
    extern __shared__ __attribute__((aligned(16))) double sRow[];          // [32][ncols + 4]: W_i, overwritten panel by panel with Y_i (stride = 4 mod 32: the A-operand reads spread over the banks)
    __shared__ double sU[32][36];
    __shared__ double sT[32][36];
    int batch, blk;
    if (!xcd_decode(per, a.batch, batch, blk)) return;
    if (a.active && !a.active[batch]) return;
    const int ncb = a.ncols >> 5, ld = a.ld, lds = a.ncols + 4;
    const int i = ncb + blk;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int ti = wave >> 1, tj = wave & 1;
    const double* W = a.W + (size_t)batch * a.xs;
    double* Y = a.Y + (size_t)batch * a.xs;
    const double* Tb = a.Tb + (size_t)batch * a.ts;
    for (int e0 = tid; e0 < 32 * a.ncols; e0 += 256 * 8) {                 // eight loads in flight per thread (ncols a multiple of 32)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, r = e & 31, c = min(e >> 5, a.ncols - 1);
            v[u] = W[(size_t)(32 * i + r) + (size_t)c * ld];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, r = e & 31, c = e >> 5;
            if (c < a.ncols) sRow[r * lds + c] = v[u];
        }
    }
    lds_barrier();
    double tn[4];                                                          // T of the next panel, in flight during this one
#pragma unroll
    for (int u = 0; u < 4; ++u) tn[u] = Tb[tid + 256 * u];
    for (int k = 0; k < ncb; ++k) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = tid + 256 * u; sT[e >> 5][e & 31] = tn[u]; }
        if (k + 1 < ncb) {
            const double* Tn = Tb + (size_t)((k + 1) % a.t_slots) * 1024;
#pragma unroll
            for (int u = 0; u < 4; ++u) tn[u] = Tn[tid + 256 * u];
        }
        // tile (ti, tj) of U: W_i[:, k] minus the contribution of the finished panels
        double4_f acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = sRow[(16 * ti + kq + 4 * r) * lds + 32 * k + 16 * tj + l15];
        const double* Lrow = Y + (size_t)(32 * k + 16 * tj + l15) + (size_t)kq * ld;      // L[32 k + 16 tj + l15][kq + ...]
#pragma unroll 1
        for (int p0 = 0; p0 < k; p0 += 4) {                              // four panels' operands (32 loads) in flight at once
            double bf[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k4 = 0; k4 < 8; ++k4) bf[u][k4] = p0 + u < k ? Lrow[(size_t)(32 * (p0 + u) + 4 * k4) * ld] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u < k) {
#pragma unroll
                    for (int k4 = 0; k4 < 8; ++k4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sRow[(16 * ti + l15) * lds + 32 * (p0 + u) + 4 * k4 + kq], bf[u][k4], acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sU[16 * ti + kq + 4 * r][16 * tj + l15] = acc[r];
        lds_barrier();
        double4_f y = { 0, 0, 0, 0 };
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4)
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(sU[16 * ti + l15][4 * k4 + kq], sT[4 * k4 + kq][16 * tj + l15], y, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) sRow[(16 * ti + kq + 4 * r) * lds + 32 * k + 16 * tj + l15] = y[r];
        lds_barrier();
    }
    for (int e = tid; e < 32 * a.ncols; e += 256) {
        const int r = e & 31, c = e >> 5;
        Y[(size_t)(32 * i + r) + (size_t)c * ld] = sRow[r * lds + c];
        if (a.Y2) a.Y2[(size_t)batch * a.y2s + (size_t)(a.y2_row0 + 32 * blk + r) + (size_t)c * a.ld_y2] = sRow[r * lds + c];
    }
}

}  // namespace

int dbg_read_chol(long long* out, int n) { return dbg_read_local(out, n); }

#ifndef GEMM64_MIN_BATCH
#define GEMM64_MIN_BATCH 8
#endif
void launch_gemm(const GemmArgs& g, hipStream_t st)
{
    const int nbi = (g.M + 31) / 32, nbj = (g.N + 31) / 32;
    const int blocks = g.lower ? nbi * (nbi + 1) / 2 : nbi * nbj;
    GemmArgs h = g;
    if (h.ksplit < 1) h.ksplit = 1;
    // a K loop worth pipelining: 64 x 64 blocks (k_gemm64) - when there are enough filters to fill the chip with them.  A handful of
    // filters is latency-bound: k_gemm's workgroup is three memory round trips deep (its four waves split K), k_gemm64's ten staged
    // chunks (16 against 6 us per product for one filter at 27 clones)
    if (h.ksplit == 1 && g.K >= 96 && g.M >= 96 && g.N >= 64 && g.batch > GEMM64_MIN_BATCH) {
        const int n64i = (g.M + 63) / 64, n64j = (g.N + 63) / 64;
        const dim3 grid64(xcd_grid(g.lower ? n64i * (n64i + 1) / 2 : n64i * n64j, g.batch));
        if (g.modeA == 0 && g.modeB == 0) hipLaunchKernelGGL((k_gemm64<0, 0>), grid64, dim3(256), 0, st, h);
        else if (g.modeA == 0 && g.modeB == 1) hipLaunchKernelGGL((k_gemm64<0, 1>), grid64, dim3(256), 0, st, h);
        else if (g.modeA == 1 && g.modeB == 0) hipLaunchKernelGGL((k_gemm64<1, 0>), grid64, dim3(256), 0, st, h);
        else hipLaunchKernelGGL((k_gemm64<1, 1>), grid64, dim3(256), 0, st, h);
        return;
    }
    const dim3 grid(xcd_grid(blocks * h.ksplit, g.batch));             // XCD-aware 1-D order, decoded in the kernel
    if (g.modeA == 0 && g.modeB == 0) hipLaunchKernelGGL((k_gemm<0, 0>), grid, dim3(256), 0, st, h);
    else if (g.modeA == 0 && g.modeB == 1) hipLaunchKernelGGL((k_gemm<0, 1>), grid, dim3(256), 0, st, h);
    else if (g.modeA == 1 && g.modeB == 0) hipLaunchKernelGGL((k_gemm<1, 0>), grid, dim3(256), 0, st, h);
    else hipLaunchKernelGGL((k_gemm<1, 1>), grid, dim3(256), 0, st, h);
}

int gram_ksplit(int m, int n)
{
    const int nb = (n + 1 + 63) / 64, blocks = nb * (nb + 1) / 2;
    int ks = (1024 + blocks - 1) / blocks;
    const int kmax = (m + 127) / 128;                                       // at least 4 chunks of 32 rows per workgroup
    if (ks > kmax) ks = kmax;
    if (ks > 64) ks = 64;
    return ks < 1 ? 1 : ks;
}

void launch_gram(const double* H, int ldh, const double* rv, int m, int n, double* part, size_t pstride, int n_ld, int ksplit, hipStream_t st)
{
    const int nb = (n + 1 + 63) / 64;
    hipLaunchKernelGGL(k_gram_tn, dim3(nb * (nb + 1) / 2, ksplit), dim3(256), 0, st, H, ldh, rv, m, n, part, pstride, n_ld, ksplit);
}

void launch_chol_sweep(const CholArgs& a0, hipStream_t st)
{
    CholArgs a = a0;
    const int ncb = a.ncols / 32;
    if (a.t_slots < 2) a.t_slots = 2;
    // split sweep: factorise the square part with the step kernel, then every carried 32-row block in ONE launch
    const size_t lds_row = sizeof(double) * 32 * (size_t)(a.ncols + 4);
    // (for a handful of matrices the carried rows ride through the panel launches instead: the resident kernel's walk over the
    // panels is a latency chain that only pays off when there are enough workgroups to hide it)
    const bool split = a.t_slots >= ncb && a.rows > a.ncols && lds_row <= 96 * 1024 && a.batch * ((a.rows - a.ncols) / 32) >= 256;
    const int rows_all = a.rows;
    if (split) a.rows = a.ncols;
    const int nbr = a.rows / 32;
    hipLaunchKernelGGL(k_chol_first, dim3(a.batch), dim3(256), 0, st, a);
    for (int k = 0; k < ncb; ++k) {
        int blocks = 0;
        if (k + 1 >= ncb) blocks = nbr - (k + 1);
        else for (int j = k + 1; j < ncb; ++j) blocks += nbr - j;
        if (blocks > 0) hipLaunchKernelGGL(k_chol_step, dim3(xcd_grid(blocks, a.batch)), dim3(256), 0, st, a, k, blocks);
    }
    if (split) {
        static size_t attr = 0;
        if (lds_row > attr) { hipFuncSetAttribute((const void*)k_chol_carried, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_row); attr = lds_row; }
        a.rows = rows_all;
        hipLaunchKernelGGL(k_chol_carried, dim3(xcd_grid((rows_all - a.ncols) / 32, a.batch)), dim3(256), lds_row, st, a, (rows_all - a.ncols) / 32);
    }
}
