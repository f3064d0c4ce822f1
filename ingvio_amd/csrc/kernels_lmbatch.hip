// kernels_lmbatch.hip — the SLAM-landmark update for a BATCH of filters (SURVEY §8 f-2 inside the frame path; gfx950 only).
//
// Reference: LandmarkUpdate::updateLandmark{Mono,Stereo} (LandmarkUpdate.cpp:32-149): for every in-state landmark seen in the
// current frame, rows of the current observation against [extended pose 9 | extrinsics 6 | anchor clone 6 | landmark 3]
// (calcResJacobianSingleLandmark{Mono,Stereo}, :521-572 / :619-686, incl. the as-written right-camera anchor block, quirk Q12),
// a chi^2 gate per landmark on the prior (dof = rows, Update.cpp:81-102), the accepted rows stacked, one ekfUpdate.
//
//   k_lm_build    one workgroup per filter, one wave per landmark at a time: rows, S_j = H_j P H_j^T + s^2 I (4x4, the 24
//                 involved columns gathered from the resident covariance), gamma_j, the gate; then the accepted landmarks'
//                 blocks are written COMPACTED (24 columns + 4 column bases each) and the residual into the carried row of the
//                 Cholesky workspace.
//   k_lm_products P H^T and H P H^T + s^2 I from the compact blocks (24-sparse rows: a dense GEMM would spend 90 % of its
//                 arithmetic on zeros), written straight into the sweep's working matrix.
//   k_lm_finish   dx = Y z from the carried rows of the sweep (Y = P H^T L^-T, z = L^-1 res).
//   k_add_noise   diagonal / dense measurement noise added to S: ingvio_ekf_update hands rows over in the same dense layout when
//                 their S does not fit in LDS (the host scatters the columns).
// Between them: launch_gemm / launch_chol_sweep (kernels_chol.hip) and k_downdate (kernels_ekf.hip).
#include "launch_lmbatch.h"

namespace {

__device__ __forceinline__ void m3_mul(const double* A, const double* B, double* C)      // row-major 3x3
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void m3_T(const double* A, double* T)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}
__device__ __forceinline__ void m3_v(const double* A, const double* v, double* o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void skew3(const double* v, double* S)
{
    S[0] = 0.0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0.0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0.0;
}
// 2x3 (row-major) times 3x3
__device__ __forceinline__ void m23_mul(const double* A, const double* B, double* C)
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// rows of one landmark: Hj [4][24] row-major = [epose 9 | ext 6 | anchor 6 | pf 3], res [4]; returns rows (2 mono / 4 stereo)
// COMPACT: Hj [4][21], the extended pose's three velocity columns (structurally zero) left out: [theta 3 | p 3 | ext 6 | anchor 6 | pf 3]
template <bool COMPACT = false>
__device__ __forceinline__ void lm_rows(const double* pose, const double* pf, const double* uv, const LmOpts& op, double* Hj, double* res)   // Hj, res: LDS
{
    constexpr int RS = COMPACT ? 21 : 24, O_EXT = COMPACT ? 6 : 9, O_ANC = COMPACT ? 12 : 15, O_PF = COMPACT ? 18 : 21;
    const double *R_i2w = pose, *p_i2w = pose + 9, *R_cl2i = pose + 12, *p_c2i = pose + 21;
    double RiT[9], RcT[9], Rw2cl[9], d[3], pf_i[3], d2[3], pf_cl[3], Sw[9], Si[9];
    m3_T(R_i2w, RiT); m3_T(R_cl2i, RcT);
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = pf[i] - p_i2w[i];
    m3_v(RiT, d, pf_i);
#pragma unroll
    for (int i = 0; i < 3; ++i) d2[i] = pf_i[i] - p_c2i[i];
    m3_v(RcT, d2, pf_cl);
    m3_mul(RcT, RiT, Rw2cl);
    skew3(pf, Sw); skew3(pf_i, Si);
#pragma unroll 4
    for (int e = 0; e < 4 * RS; ++e) Hj[e] = 0.0;
    res[0] = res[1] = res[2] = res[3] = 0.0;
    const int eyes = op.stereo ? 2 : 1;
#pragma unroll
    for (int eye = 0; eye < 2; ++eye) {
        if (eye < eyes) {
            double q[3], HL[6];
            if (eye == 0) { q[0] = pf_cl[0]; q[1] = pf_cl[1]; q[2] = pf_cl[2]; }
            else { m3_v(op.R_lr, pf_cl, q); q[0] += op.t_lr[0]; q[1] += op.t_lr[1]; q[2] += op.t_lr[2]; }
            const double iz = 1.0 / q[2];
            const double Hp[6] = { iz, 0.0, -q[0] / (q[2] * q[2]), 0.0, iz, -q[1] / (q[2] * q[2]) };
            if (eye == 0) {
#pragma unroll
                for (int e = 0; e < 6; ++e) HL[e] = Hp[e];
            } else m23_mul(Hp, op.R_lr, HL);
            res[2 * eye] = uv[2 * eye] - q[0] * iz;
            res[2 * eye + 1] = uv[2 * eye + 1] - q[1] * iz;
            double A[6], B[6], Cc[6], D[6], E[6];
            m23_mul(HL, Rw2cl, A);          // d/d pf
            m23_mul(A, Sw, B);              // d/d theta (extended pose)
            m23_mul(HL, RcT, Cc);
            m23_mul(Cc, Si, D);             // d/d theta (extrinsics)
            m23_mul(HL, Sw, E);             // the right rows' anchor block as written (no R_w2cl: quirk Q12)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double* h = Hj + RS * (2 * eye + r);
                    h[c] = B[3 * r + c]; h[3 + c] = -A[3 * r + c];
                    h[O_EXT + c] = D[3 * r + c]; h[O_EXT + 3 + c] = -Cc[3 * r + c];
                    h[O_ANC + c] = eye == 0 ? -B[3 * r + c] : -E[3 * r + c];
                    h[O_PF + c] = A[3 * r + c];
                }
        }
    }
}

#define LMB_NT 1024
__global__ __launch_bounds__(LMB_NT) void k_lm_build(CovView cv, LmView lv, LmOpts op, int b0, double* __restrict__ Hd_all, size_t hstride,
                                                      int n_ld, int m_cap, double* __restrict__ X_all, size_t xstride, int ldx, int res_row,
                                                      double* __restrict__ gamma_out, int* __restrict__ accept_out, int* __restrict__ m_out,
                                                      double* __restrict__ dx_all, int* __restrict__ cidx_all)
{
    __shared__ double sH[LM_MAX][100];                                   // rows of every landmark: H_j 96 + res 4
    __shared__ int sCol[LMB_NT / 64][24];
    __shared__ int sAcc[LM_MAX], sOff[LM_MAX + 1];
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int L = min(lv.n_lm[b], LM_MAX), n = cv.n[b], ld = cv.ldp;
    const int per = op.stereo ? 4 : 2;
    const double* P = cov_ptr(cv, b);
    const double* pose = lv.pose + (size_t)b * 24;
    const int ie = lv.idx[2 * b], ix = lv.idx[2 * b + 1];
    for (int l = wave; l < L; l += LMB_NT / 64) {
        const size_t o = (size_t)b * lv.lmax + l;
        const int il = lv.lm_idx[o], ia = lv.anchor_idx[o];
        const bool on = lv.tracked[o] != 0 && il >= 0 && il + 3 <= n && ia >= 0 && ia + 6 <= n;
        lm_rows(pose, lv.pf + 3 * o, lv.uv + 4 * o, op, &sH[l][0], &sH[l][96]);   // every lane the same arithmetic and the same stores
        if (lane < 24) sCol[wave][lane] = lane < 9 ? ie + lane : (lane < 15 ? ix + lane - 9 : (lane < 21 ? ia + lane - 15 : il + lane - 21));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // S = H P24 H^T + var I: the 576 entries of P24 dealt to the lanes, 10 partial sums each, summed across the wave
        double S[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        if (on) {
            for (int e = lane; e < 576; e += 64) {
                const int a = e / 24, c = e - 24 * a;
                const double p = P[(size_t)sCol[wave][a] + (size_t)sCol[wave][c] * ld];
                int t = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int s = 0; s <= r; ++s) S[t++] += sH[l][24 * r + a] * p * sH[l][24 * s + c];
            }
        }
#pragma unroll
        for (int t = 0; t < 10; ++t) S[t] = wave_sum(S[t]);
        // gamma = res^T S^-1 res through the Cholesky factor of the leading per x per block (uniform)
        double g = 0.0;
        bool acc = false;
        if (on) {
            double Lc[10], z[4];
            int t = 0;
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s = 0; s <= r; ++s, ++t) {
                    if (r < per) {
                        double v = S[t] + (r == s ? op.var : 0.0);
#pragma unroll
                        for (int k = 0; k < s; ++k) v -= Lc[r * (r + 1) / 2 + k] * Lc[s * (s + 1) / 2 + k];
                        if (r == s) { ok = ok && v > 0.0; Lc[t] = sqrt(v); } else Lc[t] = v / Lc[s * (s + 1) / 2 + s];
                    } else Lc[t] = r == s ? 1.0 : 0.0;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = r < per ? sH[l][96 + r] : 0.0;
#pragma unroll
                for (int k = 0; k < r; ++k) v -= Lc[r * (r + 1) / 2 + k] * z[k];
                z[r] = v / Lc[r * (r + 1) / 2 + r];
                g += z[r] * z[r];
            }
            acc = ok && g < op.chi2_thr;                                 // Update.cpp:98-100 with dof = rows
        }
        if (lane == 0) {
            sAcc[l] = acc ? 1 : 0;
            gamma_out[(size_t)bl * LM_MAX + l] = on ? g : -1.0;
            accept_out[(size_t)bl * LM_MAX + l] = acc ? 1 : 0;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int o = 0;
        for (int l = 0; l < L; ++l) { sOff[l] = o; o += sAcc[l] ? per : 0; }
        if (o > m_cap) {                                                 // capacity: keep the first landmarks that fit
            o = 0;
            for (int l = 0; l < L; ++l) { if (sAcc[l] && o + per > m_cap) { sAcc[l] = 0; accept_out[(size_t)bl * LM_MAX + l] = 0; } sOff[l] = o; o += sAcc[l] ? per : 0; }
        }
        sOff[LM_MAX] = o;
        m_out[bl] = o;
    }
    __syncthreads();
    const int m = sOff[LM_MAX];
    double* Hd = Hd_all + (size_t)bl * hstride;
    double* X = X_all + (size_t)bl * xstride;
    if (m == 0) {
        double* dx = dx_all + (size_t)b * ld;
        for (int r = tid; r < ld; r += LMB_NT) dx[r] = 0.0;
        return;
    }
    // the accepted landmarks' blocks, compacted: [96 row entries | 4 residuals] and the four column bases; residuals into the
    // carried row of the sweep's working matrix
    int* cidx = cidx_all + (size_t)bl * LM_MAX * 4;
    for (int l = wave; l < L; l += LMB_NT / 64) {
        if (!sAcc[l]) continue;
        const int a = sOff[l] / per;
        const size_t o = (size_t)b * lv.lmax + l;
        for (int e = lane; e < 100; e += 64) Hd[(size_t)a * 100 + e] = sH[l][e];
        if (lane < 4) cidx[4 * a + lane] = lane == 0 ? ie : (lane == 1 ? ix : (lane == 2 ? lv.anchor_idx[o] : lv.lm_idx[o]));
        if (lane < per) X[(size_t)res_row + (size_t)(sOff[l] + lane) * ldx] = sH[l][96 + lane];
    }
    for (int R = m + tid; R < m_cap; R += LMB_NT) X[(size_t)res_row + (size_t)R * ldx] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// P H^T and S = H P H^T + s^2 I from the compact blocks.  A landmark's 4 rows share its 24 columns, so one pass over those 24
// columns of P serves four columns of P H^T (a row-by-row gather would read 4x as much; the dense GEMM this replaces did 10x
// the arithmetic on zeros).  grid = (groups of 4 landmarks covering m_cap columns, nb), 256 threads:
//   phase A, thread = state row r:   y_R[r] = sum_c P[r, col_c] h_R[c]  for the group's (up to) 16 rows R  -> carried rows of X
//   phase B, thread = stacked row R2: S[R2, R] = sum_c h_R2[c] y_R[col_R2,c]  (+ s^2 on the diagonal)       -> S block of X
// Columns beyond the accepted rows are written as padding (s^2 on the diagonal: the sweep runs on whole 32-column panels).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lm_products(CovView cv, int b0, const double* __restrict__ Hc_all, size_t hstride,
                                                     const int* __restrict__ cidx_all, const int* __restrict__ m_all, int per, double var,
                                                     double* __restrict__ X_all, size_t xstride, int ldx, int m_cap, int n_rows, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) double sY[];          // [16][n_rows] when it fits: the group's columns of P H^T for phase B
    const int bl = blockIdx.y, b = b0 + bl, m = m_all[bl];
    if (m == 0) return;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int R0 = per * 4 * g, R1 = min(m_cap, R0 + per * 4);          // this workgroup's columns of X
    if (R0 >= m_cap) return;
    const int n = cv.n[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    const double* Hc = Hc_all + (size_t)bl * hstride;
    const int* cidx = cidx_all + (size_t)bl * LM_MAX * 4;
    double* X = X_all + (size_t)bl * xstride;
    double* Yc = X + m_cap;                                              // carried rows: P H^T
    // ---- phase A
    for (int r = tid; r < n_rows; r += 256) {
        for (int q4 = 0; q4 < 4; ++q4) {
            const int Ra = R0 + per * q4;                                // first row of landmark 4 g + q4
            if (Ra >= R1) break;
            if (Ra < m && r < n) {
                const int a = 4 * g + q4;
                const double* h = Hc + (size_t)a * 100;
                const int c0 = cidx[4 * a], c1 = cidx[4 * a + 1], c2 = cidx[4 * a + 2], c3 = cidx[4 * a + 3];
                double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
#pragma unroll
                for (int c = 0; c < 24; ++c) {
                    const int col = c < 9 ? c0 + c : (c < 15 ? c1 + c - 9 : (c < 21 ? c2 + c - 15 : c3 + c - 21));
                    const double p = P[(size_t)r + (size_t)col * ld];
                    y0 += p * h[c]; y1 += p * h[24 + c]; y2 += p * h[48 + c]; y3 += p * h[72 + c];
                }
                Yc[(size_t)r + (size_t)Ra * ldx] = y0;
                Yc[(size_t)r + (size_t)(Ra + 1) * ldx] = y1;
                if (per == 4) { Yc[(size_t)r + (size_t)(Ra + 2) * ldx] = y2; Yc[(size_t)r + (size_t)(Ra + 3) * ldx] = y3; }
                if (use_lds) {
                    double* sy = sY + (size_t)(per * q4) * n_rows + r;
                    sy[0] = y0; sy[n_rows] = y1;
                    if (per == 4) { sy[2 * (size_t)n_rows] = y2; sy[3 * (size_t)n_rows] = y3; }
                }
            } else {
                for (int q = 0; q < per; ++q) Yc[(size_t)r + (size_t)(Ra + q) * ldx] = 0.0;      // padding column / rows beyond the state
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- phase B
    for (int R2 = tid; R2 < m_cap; R2 += 256) {
        double h2[24];
        int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        const bool live = R2 < m;
        if (live) {
            const int a2 = R2 / per, q2 = R2 - per * a2;
            const double* h = Hc + (size_t)a2 * 100 + 24 * q2;
#pragma unroll
            for (int c = 0; c < 24; ++c) h2[c] = h[c];
            k0 = cidx[4 * a2]; k1 = cidx[4 * a2 + 1]; k2 = cidx[4 * a2 + 2]; k3 = cidx[4 * a2 + 3];
        }
        for (int R = R0; R < R1; ++R) {
            double s = 0.0;
            if (live && R < m) {
                const double* y = use_lds ? sY + (size_t)(R - R0) * n_rows : Yc + (size_t)R * ldx;
#pragma unroll
                for (int c = 0; c < 24; ++c) {
                    const int col = c < 9 ? k0 + c : (c < 15 ? k1 + c - 9 : (c < 21 ? k2 + c - 15 : k3 + c - 21));
                    s += h2[c] * y[col];
                }
            }
            if (R2 == R) s += var;
            X[(size_t)R2 + (size_t)R * ldx] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_lm_rows + k_lm_front: k_lm_build + k_lm_products for states of up to 256 rows.
//   k_lm_rows   one lane per landmark: the rows of every TRACKED landmark as compact 4 x 21 blocks (+ residual) and the list of
//               candidates (landmark, anchor column, landmark column) in global memory.  A kernel of its own because its register
//               appetite (the rotation algebra) would cap the occupancy of the product loops - and because rows that the NEXT kernel
//               only reads can be fetched there with SCALAR loads: a row of H is the same for every lane of a wave.
//   k_lm_front  one workgroup per filter, 512 threads, in chunks of 4 landmarks (16 columns):
//                 A  thread = state row r: the chunk's columns of P H^T - the 12 shared columns of P (pose, extrinsics) sit in
//                    registers for the whole kernel, 9 loads per landmark, the entries of H arrive in SGPRs
//                 B  thread = stacked row: the chunk's columns of S = H (P H^T) + s^2 I, lower triangle, from the chunk's P H^T in
//                    LDS and the thread's own row of H
//               then the chi^2 gate per landmark on the DIAGONAL BLOCKS of that S (S_j = H_j P H_j^T + s^2 I is what
//               Update.cpp:81-102 tests; the 24 x 24 gather per landmark of k_lm_build is gone) and the row map of the accepted
//               landmarks: k_lm_factor / k_lm_carry (kernels_lmchol.hip) pick their rows of S and P H^T through it.
// ---------------------------------------------------------------------------------------------
#define LMF_NT 512
#ifndef LMF_WPE
#define LMF_WPE 4                                                        // two workgroups per CU
#endif
#define LMF_HS 90                                                        // 4 x 21 row entries, 4 residuals, pad (even: 16-byte rows)
#define LMF_YS 18                                                        // LDS row stride of the chunk's columns of P H^T (16-byte aligned rows)
__global__ __launch_bounds__(64) void k_lm_rows(CovView cv, LmView lv, LmOpts op, int b0, int lcap, double* __restrict__ Hc_all, size_t hstride,
                                                int* __restrict__ cand_all, double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    const int bl = blockIdx.x, b = b0 + bl, lane = threadIdx.x;
    const int L = min(min(lv.n_lm[b], LM_MAX), lcap), n = cv.n[b];
    double* Hc = Hc_all + (size_t)bl * hstride;
    int* cand = cand_all + (size_t)bl * LM_MAX * 4;                      // [a]: landmark, anchor column, landmark column; cand[3] = candidates
    const int l = lane;
    const size_t o = (size_t)b * lv.lmax + l;
    int il = -1, ia = -1;
    bool on = false;
    if (l < L) {
        il = lv.lm_idx[o]; ia = lv.anchor_idx[o];
        on = lv.tracked[o] != 0 && il >= 0 && il + 3 <= n && ia >= 0 && ia + 6 <= n;
        if (!on) { gamma_out[(size_t)bl * LM_MAX + l] = -1.0; accept_out[(size_t)bl * LM_MAX + l] = 0; }
    }
    const unsigned long long mask = __ballot(on);
    // candidate order: by anchor clone, then by landmark.  The order of the stacked rows is free (the update is invariant to it); with
    // the landmarks of a chunk sharing their anchor, k_lm_front's gathers of the anchor's six columns of P coincide for the four lane
    // groups and repeat from chunk to chunk while the lines are still in L2 (in tracked order the kernel re-read them from HBM:
    // 814 MiB per launch of 512 filters against 566 for P, P H^T and S once).
    int a = 0;
    {
        const int key = on ? ia * 64 + lane : 0x7fffffff;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) a += (__shfl(key, j, WAVE) < key) ? 1 : 0;
    }
    if (on) {
        lm_rows<true>(lv.pose + (size_t)b * 24, lv.pf + 3 * o, lv.uv + 4 * o, op, Hc + (size_t)a * LMF_HS, Hc + (size_t)a * LMF_HS + 84);   // straight to global (a private array: scratch)
        cand[4 * a] = l; cand[4 * a + 1] = ia; cand[4 * a + 2] = il;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) cand[3] = __popcll(mask);                             // (slot 3 of candidate 0 is unused)
}

typedef double double4_l __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(LMF_NT) __attribute__((amdgpu_waves_per_eu(LMF_WPE, LMF_WPE)))
void k_lm_front(CovView cv, LmView lv, LmOpts op, int b0, int lcap, int m_cap, int n_rows, const double* __restrict__ Hc_all, size_t hstride,
                const int* __restrict__ cand_all, double* __restrict__ X_all, size_t xstride, int ldx, int res_row, double* __restrict__ gamma_out,
                int* __restrict__ accept_out, int* __restrict__ m_out, double* __restrict__ dx_all, int* __restrict__ rowmap_all)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];        // sY [256][LMF_YS] | sH [lcap][LMF_HS]
    __shared__ int sAcc[LM_MAX], sIa[LM_MAX], sIl[LM_MAX], sM;
    double* sY = smem;
    double* sH = smem + 256 * LMF_YS;                                    // candidate a: entry c of row q at [a][4 c + q], residuals at [a][84 + q]
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int kq = lane >> 4, l15 = lane & 15;
    const int n = cv.n[b], ld = cv.ldp;
    const int per = op.stereo ? 4 : 2;
    const double* P = cov_ptr(cv, b);
    const int ie = lv.idx[2 * b], ix = lv.idx[2 * b + 1];
    double* X = X_all + (size_t)bl * xstride;
    const double* __restrict__ Hc = Hc_all + (size_t)bl * hstride;
    const int* __restrict__ cand = cand_all + (size_t)bl * LM_MAX * 4;
    const int La = cand[3], M0 = per * La;
    dbg_stamp(0);
    if (La == 0) {
        if (tid == 0) m_out[bl] = 0;
        double* dx = dx_all + (size_t)b * ld;
        for (int q = tid; q < ld; q += LMF_NT) dx[q] = 0.0;
        for (int R = tid; R < m_cap; R += LMF_NT) rowmap_all[(size_t)bl * m_cap + R] = -1;
        return;
    }
    for (int e = tid; e < La * LMF_HS; e += LMF_NT) {                    // k_lm_rows wrote [q][21]: transposed on the way in
        const int a = e / LMF_HS, i = e - a * LMF_HS;
        sH[a * LMF_HS + (i < 84 ? 4 * (i % 21) + i / 21 : i)] = Hc[e];
    }
    if (tid < La) { sIa[tid] = cand[4 * tid + 1]; sIl[tid] = cand[4 * tid + 2]; }
    // ---- phase A on the matrix cores: (P H^T)^T chunk [16 columns] x [16 state rows] = Hblk [16 x 48] Pg^T [48 x 16], the 48
    // gathered columns of P being 12 shared ones (pose theta / p, extrinsics; k-steps 0..2) and 9 per landmark of the chunk
    // (anchor 6, position 3; k-step 3 + c takes entry c of ALL FOUR landmarks, lane group kq = landmark slot) - Hblk is
    // block-sparse there.  Wave w owns the row blocks 2 w, 2 w + 1.  B operand: lane (kq, l15) supplies
    // P[16 blk + l15][gcol(4 s + kq)] for k-step s; the three steps of the shared columns stay in registers for the whole kernel.
    int rowA[2], rrA[2];
    double bsh[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rowA[i] = 16 * (2 * wave + i) + l15;
        rrA[i] = rowA[i] < n ? rowA[i] : 0;                              // rows beyond the state: loads from row 0, results zeroed
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int k = 4 * st + kq;
            bsh[i][st] = P[(size_t)rrA[i] + (size_t)(k < 6 ? ie + k : ix + k - 6) * ld];
        }
    }
    const int slot_m = l15 >> 2, q_m = l15 & 3;                          // the lane's column of the chunk as A operand: landmark slot, row
    // phase B: thread (slot = wave < 4, a2 = lane): the 4 x 4 block S[rows of candidate a2][columns of candidate 4 g + slot]
    const int a2 = lane;
    const bool rowB = wave < 4 && a2 < La;
    __syncthreads();
    const int c2a = rowB ? sIa[a2] : 0, c2l = rowB ? sIl[a2] : 0;
    dbg_stamp(1);
    double* Yc = X + m_cap;                                              // carried rows: P H^T
    const int nch = (La + 3) >> 2;
    double bv[2][9];
    auto fetch_b = [&](int g) {
        const int ak = min(4 * g + kq, La - 1), cia = sIa[ak], cil = sIl[ak];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 9; ++c) bv[i][c] = P[(size_t)rrA[i] + (size_t)(c < 6 ? cia + c : cil + c - 6) * ld];
    };
#pragma unroll 1
    for (int g = 0; g < nch; ++g) {
        // ---- A  (fetching chunk g + 1 under phase B was tried: the conditional stores that follow the loads in program order
        //          make the wait for them a full drain, s_waitcnt vmcnt(0) - 0.19 ms against 0.16)
        {
            fetch_b(g);
            const int a_m = 4 * g + slot_m;
            const bool col_on = a_m < La && q_m < per;
            const double* hm = sH + (size_t)(col_on ? a_m : 0) * LMF_HS + q_m;
            double af[12];
#pragma unroll
            for (int st = 0; st < 3; ++st) af[st] = col_on ? hm[4 * (4 * st + kq)] : 0.0;
            const bool own = col_on && slot_m == kq;
#pragma unroll
            for (int st = 3; st < 12; ++st) af[st] = own ? hm[4 * (9 + st)] : 0.0;
            double4_l accs[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                double4_l acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int st = 0; st < 3; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[st], bsh[i][st], acc, 0, 0, 0);
#pragma unroll
                for (int st = 3; st < 12; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[st], bv[i][st - 3], acc, 0, 0, 0);
                accs[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double4_l acc = accs[i];
                // acc[r] = (P H^T)[row rowA][column 4 r + kq of the chunk]: landmark slot r, row kq
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double y = rowA[i] < n ? acc[r] : 0.0;
                    const int a = 4 * g + r;
                    if (a < La && kq < per && rowA[i] < n_rows) Yc[(size_t)rowA[i] + (size_t)(per * a + kq) * ldx] = y;      // the carried part has n_rows (n32) rows
                    sY[(size_t)rowA[i] * LMF_YS + 4 * r + kq] = y;
                }
            }
        }
        lds_barrier();
        if (g == 0) dbg_stamp(2);
        // ---- B (lower triangle: the row's landmark is not before the column's)
        {
            const int ac = 4 * g + wave;
            if (rowB && ac < La && a2 >= ac) {
                double s[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[i][j] = 0.0;
                const double* hrow = sH + (size_t)a2 * LMF_HS;
#pragma unroll
                for (int c = 0; c < 21; ++c) {
                    const int row = c < 6 ? ie + c : (c < 12 ? ix + c - 6 : (c < 18 ? c2a + c - 12 : c2l + c - 18));
                    const double2 v0 = *reinterpret_cast<const double2*>(&sY[(size_t)row * LMF_YS + 4 * wave]);
                    const double2 v1 = *reinterpret_cast<const double2*>(&sY[(size_t)row * LMF_YS + 4 * wave + 2]);
                    const double2 h0 = *reinterpret_cast<const double2*>(&hrow[4 * c]);
                    const double2 h1 = *reinterpret_cast<const double2*>(&hrow[4 * c + 2]);
                    const double hv[4] = { h0.x, h0.y, h1.x, h1.y }, vv[4] = { v0.x, v0.y, v1.x, v1.y };
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            s[i][j] = fma(hv[i], vv[j], s[i][j]);
                            asm volatile("" : "+v"(s[i][j]));            // pinned: left alone, the scheduler runs one accumulator's 21 steps at a
                        }                                                // time, every operand of the block live (spills)
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < per && j < per) X[(size_t)(per * a2 + i) + (size_t)(per * ac + j) * ldx] = s[i][j] + ((a2 == ac && i == j) ? op.var : 0.0);
            }
        }
        lds_barrier();
        if (g == 0) dbg_stamp(3);
    }
    dbg_stamp(4);
    __syncthreads();                                                     // the diagonal blocks of S are visible to the gate
    // ---- the gate: gamma = res^T S_j^-1 res through the Cholesky factor of the per x per diagonal block
    if (tid < La) {
        const int a = tid, l = cand[4 * a];
        const double* h = sH + (size_t)a * LMF_HS;
        double Lc[10], z[4], g = 0.0;
        bool ok = true;
        int t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j, ++t) {
                if (i < per) {
                    double v = X[(size_t)(per * a + i) + (size_t)(per * a + j) * ldx];
#pragma unroll
                    for (int k = 0; k < j; ++k) v -= Lc[i * (i + 1) / 2 + k] * Lc[j * (j + 1) / 2 + k];
                    if (i == j) { ok = ok && v > 0.0; Lc[t] = sqrt(v); } else Lc[t] = v / Lc[j * (j + 1) / 2 + j];
                } else Lc[t] = i == j ? 1.0 : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = i < per ? h[84 + i] : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) v -= Lc[i * (i + 1) / 2 + k] * z[k];
            z[i] = v / Lc[i * (i + 1) / 2 + i];
            g += z[i] * z[i];
        }
        const bool acc = ok && g < op.chi2_thr;                          // Update.cpp:98-100 with dof = rows
        sAcc[a] = acc ? 1 : 0;
        gamma_out[(size_t)bl * LM_MAX + l] = g;
        accept_out[(size_t)bl * LM_MAX + l] = acc ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (q < per) X[(size_t)res_row + (size_t)(per * a + q) * ldx] = h[84 + q];
    }
    __syncthreads();
    if (wave == 0) {
        const bool acc = lane < La && sAcc[lane] != 0;
        const unsigned long long mask = __ballot(acc);
        const int pos = __popcll(mask & ((1ull << lane) - 1ull)), m = per * __popcll(mask);
        int* rowmap = rowmap_all + (size_t)bl * m_cap;
        if (acc) for (int q = 0; q < per; ++q) rowmap[per * pos + q] = per * lane + q;
        for (int R = m + lane; R < m_cap; R += 64) rowmap[R] = -1;
        if (lane == 0) { m_out[bl] = m; sM = m; }
    }
    __syncthreads();
    if (sM == 0) {
        double* dx = dx_all + (size_t)b * ld;
        for (int q = tid; q < ld; q += LMF_NT) dx[q] = 0.0;
    }
    dbg_stamp(5);
    (void)M0;
}

// dx = Y z (rows < n), Y = carried rows [y_row0, y_row0 + n) of the sweep's output, z = row z_row
__global__ __launch_bounds__(256) void k_lm_finish(CovView cv, int b0, const double* __restrict__ Y_all, size_t ystride, int ldy, int y_row0,
                                                   int z_row, const int* __restrict__ m_all, double* __restrict__ dx_all,
                                                   const int* __restrict__ status)
{
    const int bl = blockIdx.y, b = b0 + bl, m = m_all[bl];
    if (m == 0) return;
    const int n = cv.n[b], ld = cv.ldp, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= ld) return;
    if (status[b] & 4) { dx_all[(size_t)b * ld + r] = 0.0; return; }      // S not positive definite (the sweep's fail bit): no update
    const double* Y = Y_all + (size_t)bl * ystride;
    double d0 = 0.0, d1 = 0.0;
    if (r < n) {
        int i = 0;
        for (; i + 1 < m; i += 2) {
            d0 += Y[(size_t)(y_row0 + r) + (size_t)i * ldy] * Y[(size_t)z_row + (size_t)i * ldy];
            d1 += Y[(size_t)(y_row0 + r) + (size_t)(i + 1) * ldy] * Y[(size_t)z_row + (size_t)(i + 1) * ldy];
        }
        if (i < m) d0 += Y[(size_t)(y_row0 + r) + (size_t)i * ldy] * Y[(size_t)z_row + (size_t)i * ldy];
    }
    dx_all[(size_t)b * ld + r] = d0 + d1;
}

// S (lower, in the sweep's working matrix) += R for the diagonal / dense noise models; r_kind 0 is the GEMM's diag_add
__global__ __launch_bounds__(256) void k_add_noise(double* __restrict__ X_all, size_t xstride, int ldx, const double* __restrict__ noise_all,
                                                   int nstride, int r_kind, const int* __restrict__ m_all, int m_cap)
{
    const int bl = blockIdx.y, m = m_all[bl];
    if (m == 0) return;
    double* X = X_all + (size_t)bl * xstride;
    const double* nz = noise_all + (size_t)bl * nstride;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < m_cap * m_cap; e += gridDim.x * 256) {
        const int i = e % m_cap, j = e / m_cap;
        if (i < j) continue;
        double add = 0.0;
        if (r_kind == 0) add = i == j ? (i < m ? nz[0] : 1.0) : 0.0;
        else if (r_kind == 1) add = i == j ? (i < m ? nz[i] : 1.0) : 0.0;
        else add = (i < m && j < m) ? nz[(size_t)i + (size_t)j * m] : (i == j ? 1.0 : 0.0);
        if (add != 0.0) X[(size_t)i + (size_t)j * ldx] += add;
    }
}

}  // namespace

int dbg_read_lmbatch(long long* out, int n) { return dbg_read_local(out, n); }

void launch_lm_build(const LmBuild& L, hipStream_t st)
{
    hipLaunchKernelGGL(k_lm_build, dim3(L.nb), dim3(LMB_NT), 0, st, L.cv, L.lv, L.op, L.b0, L.Hd, L.hstride, L.n_ld, L.m_cap, L.X, L.xstride,
                       L.ldx, L.res_row, L.gamma, L.accept, L.m_out, L.dx, L.cidx);
    const int per4 = (L.op.stereo ? 4 : 2) * 4;
    const size_t lds = sizeof(double) * 16 * (size_t)L.n_rows;
    const int use_lds = lds <= 64 * 1024;
    static size_t attr = 0;
    if (use_lds && lds > attr) { hipFuncSetAttribute((const void*)k_lm_products, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
    hipLaunchKernelGGL(k_lm_products, dim3((L.m_cap + per4 - 1) / per4, L.nb), dim3(256), use_lds ? lds : 0, st, L.cv, L.b0, L.Hd, L.hstride, L.cidx, L.m_out, L.op.stereo ? 4 : 2, L.op.var,
                       L.X, L.xstride, L.ldx, L.m_cap, L.n_rows, use_lds);
}

// the fused front (states of up to 256 rows); returns false when the shape does not fit (the caller falls back to launch_lm_build)
bool launch_lm_front(const LmBuild& L, int lcap, int* rowmap, hipStream_t st)
{
    if (L.n_rows > 256 || lcap > LM_MAX || (L.op.stereo ? 4 : 2) * lcap > L.m_cap || L.hstride < (size_t)LM_MAX * LMF_HS) return false;
    hipLaunchKernelGGL(k_lm_rows, dim3(L.nb), dim3(64), 0, st, L.cv, L.lv, L.op, L.b0, lcap, L.Hd, L.hstride, L.cidx, L.gamma, L.accept);
    const size_t lds = sizeof(double) * ((size_t)256 * LMF_YS + (size_t)lcap * LMF_HS);
    static size_t attr = 0;
    if (lds > attr) { hipFuncSetAttribute((const void*)k_lm_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; }
    hipLaunchKernelGGL(k_lm_front, dim3(L.nb), dim3(LMF_NT), lds, st, L.cv, L.lv, L.op, L.b0, lcap, L.m_cap, L.n_rows, L.Hd, L.hstride, L.cidx, L.X, L.xstride,
                       L.ldx, L.res_row, L.gamma, L.accept, L.m_out, L.dx, rowmap);
    return true;
}

void launch_lm_finish(CovView cv, int b0, int nb, const double* Y, size_t ystride, int ldy, int y_row0, int z_row, const int* m, double* dx,
                      const int* status, hipStream_t st)
{
    hipLaunchKernelGGL(k_lm_finish, dim3((cv.ldp + 255) / 256, nb), dim3(256), 0, st, cv, b0, Y, ystride, ldy, y_row0, z_row, m, dx, status);
}

void launch_add_noise(double* X, size_t xstride, int ldx, const double* noise, int nstride, int r_kind, const int* m, int m_cap, int nb,
                      hipStream_t st)
{
    hipLaunchKernelGGL(k_add_noise, dim3(16, nb), dim3(256), 0, st, X, xstride, ldx, noise, nstride, r_kind, m, m_cap);
}
