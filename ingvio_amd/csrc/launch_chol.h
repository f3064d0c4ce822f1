// launch_chol.h — batched SPD building blocks out of HBM/L2 (kernels_chol.hip): FP64 MFMA GEMM with split-K and a blocked
// left-looking Cholesky sweep with carried rows.  Used where one workgroup's registers/LDS cannot hold the matrix: the Kalman
// solve of 17..36-clone windows (StateManager.cpp:359-411 at 6C = 102..216), Cholesky-QR of tall stacks (the thin-QR compression
// of RemoveLostUpdate.cpp:376-397), generic ekfUpdate with more rows than S fits in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "dev_common.h"

// C(i, j) = sum_k opA(i, k) opB(k, j)  [+ diag_add on i == j], i < M, j < N, k < K; operands are read with bounds checks (any M, N, K).
//   modeA 0: opA(i, k) = A[i + k lda]   ("row-contiguous": 16 lanes read 16 consecutive i)
//   modeA 1: opA(i, k) = A[k + i lda]   ("k-contiguous")                     -- same for modeB with j in the place of i
//   Ax / Bx (optional): row index ax of opA (column index bx of opB) is read from the vector Ax[k] (Bx[k]) instead
//   C written at C[i rs + j cs] for i < m_lim, j < n_lim;  lower != 0: only 32x32 blocks with bi >= bj are computed
//   ksplit > 1: grid-level split of K, partial p goes to C + p * csplit (the caller reduces); batch stride s*, grid.z = batch
//   active (optional): per batch element, 0 = skip
struct GemmArgs {
    const double* A; size_t sa; int lda, modeA;
    const int* a_sel; size_t a_sel_stride;          // optional: A += a_sel[batch] * a_sel_stride (the live ping-pong half of a covariance)
    const double* B; size_t sb; int ldb, modeB;
    const double* Ax; int ax; const double* Bx; int bx;
    double* C; size_t sc; long rs, cs;
    int M, N, K, m_lim, n_lim;
    int ksplit; size_t csplit;
    int lower;
    double diag_add; const double* diag_add_vec;    // diag_add_vec[batch] (e.g. the per-filter measurement variance) overrides diag_add
    const int* active;
    int batch;
    int k_from;                                     // triangular operands (round 6): 1: opA(i, k) = 0 for k < i (K starts at the block's first row), 2: opB(k, j) = 0 for k < j
                                                    // (at its first column), 0: full K.  The skipped products are exact zeros: same result
    double* Cx; size_t scx; int cx_col;             // optional: column cx_col of the product goes to the VECTOR Cx[batch scx + i] instead of C (a
                                                    // matrix-vector product riding on the GEMM as one more column of opB); Cx == nullptr: off
};
void launch_gemm(const GemmArgs& g, hipStream_t st);

// Lower triangle of [H | r]^T [H | r] (H m x n column-major, r the extra column n) as `ksplit` partial sums of the K split, each
// column-major with n_ld rows at part + p * pstride; entries above the diagonal of a 64-block row may be written too.
int gram_ksplit(int m, int n);
void launch_gram(const double* H, int ldh, const double* rv, int m, int n, double* part, size_t pstride, int n_ld, int ksplit, hipStream_t st);

// Blocked right-looking Cholesky of the leading ncols x ncols block of W (column-major, ld rows, ncols and rows multiples of 32,
// lower triangle read).  W is a WORKING copy (its trailing part is updated in place); the result goes to Y:
// Y[0:ncols] = L (upper part zero), rows ncols..rows are CARRIED: Y[r] = W[r] L^-T.  One launch per 32-column panel.
//   clamp != 0: a pivot <= clamp_rel * (original diagonal) zeroes its column (semi-definite input: rank-deficient gram);
//   clamp == 0: a non-positive pivot also zeroes the column and sets bit `fail_bit` in status[batch] (status may be null).
//   Tb: scratch, t_slots * 1024 + ncols doubles per batch element (stride ts): the T = L_d^-T of the panels and the original
//   diagonal.  t_slots = 2 (default): ping-pong, carried rows ride through the panel launches.  t_slots >= ncols / 32: every T is
//   kept and the carried rows are done by ONE launch after the factorisation (one workgroup per 32 rows, resident in LDS).
struct CholArgs {
    double* W; double* Y; size_t xs; int ld;
    double* Tb; size_t ts; int t_slots;
    int rows, ncols;
    int clamp; double clamp_rel;
    int* status; int fail_bit;
    const int* active;
    int batch;
    double* Y2; size_t y2s; int ld_y2, y2_row0;  // optional: the CARRIED rows of the result (rows ncols.. of Y) are also written to rows y2_row0.. of Y2 (ld ld_y2,
                                                 // batch stride y2s) - the next sweep's working copy takes them without a copy launch in between
};
void launch_chol_sweep(const CholArgs& a, hipStream_t st);

// The same solve for S of up to 256 rows with ONE workgroup per filter and S resident in registers (kernels_lmchol.hip): what
// hundreds of filters want (the sweep above is one memory round trip per 32 x 32 block and launch).  Reads S (whole, symmetric)
// and the carried rows P H^T / the residual row from X as the sweep does, writes Y (carried rows only: what k_downdate64 reads)
// and dx = Y z; a non-positive pivot sets fail_bit in status[filter] and leaves dx = 0.
struct LmCholArgs {
    CovView cv; int b0, nb;                      // dx and cv are indexed by the absolute filter b0 + i, everything else by i
    const double* X; double* Y; size_t xs; int ldx, mc, res_row;
    double* U; size_t us;                        // scratch: lm_chol_ws_doubles(mc) per filter (factor tiles, z)
    const int* m; int* status; int fail_bit;
    double* dx;
    const int* rowmap; int rm_stride;            // optional [nb][rm_stride]: compact row R of the system = row rowmap[R] of X (< 0: padding)
    // generic use (the large-window solve, kernels_bigwin.hip): the drop-in for launch_chol_sweep when ncols <= 256
    int carried_rows;                            // > 0: that many carried rows (rows mc .. of X) instead of the state's; dx / cv are not used then
    int m_fixed;                                 // > 0: factorise that many rows for every active filter (m[i] only says active / not)
    int write_L;                                 // != 0: also write L into rows [0, m) of Y (upper part zero), as the sweep does
};                                               // res_row < 0: no residual row (no z, no dx)
size_t lm_chol_ws_doubles(int mc);
void launch_lm_chol(const LmCholArgs& a, hipStream_t st);
int dbg_read_lmchol(long long* out, int n);
int dbg_read_chol(long long* out, int n);
