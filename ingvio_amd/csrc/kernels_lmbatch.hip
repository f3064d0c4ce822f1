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
__device__ __forceinline__ void lm_rows(const double* pose, const double* pf, const double* uv, const LmOpts& op, double* Hj, double* res)   // Hj, res: LDS
{
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
    for (int e = 0; e < 96; ++e) Hj[e] = 0.0;
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
                    double* h = Hj + 24 * (2 * eye + r);
                    h[c] = B[3 * r + c]; h[3 + c] = -A[3 * r + c];
                    h[9 + c] = D[3 * r + c]; h[12 + c] = -Cc[3 * r + c];
                    h[15 + c] = eye == 0 ? -B[3 * r + c] : -E[3 * r + c];
                    h[21 + c] = A[3 * r + c];
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
