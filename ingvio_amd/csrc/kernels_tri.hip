// kernels_tri.hip — SURVEY.md §8(f) row f-1: batched feature triangulation on the device.
// Triangulator::triangulateMonoObs (Triangulator.cpp:173-318) and its stereo wrapper (:320-359): Levenberg-Marquardt on
// (x/z, y/z, 1/z) in the frame of the LAST mono-equivalent observation, Huber-weighted normal equations, accept test
// on the unweighted cost, depth / parallax / convergence gates.  One LANE per feature (the iteration is inherently
// sequential and features are independent); the window's camera poses (left, and right = left * T_cl2cr^-1) are
// staged once per workgroup in LDS.  Operation order follows the restatement in oracle/ingvio_oracle.c so that both
// walk the same iteration path.  gfx950 only.
#include "dev_common.h"
#include "launch_tri.h"

namespace {

__device__ __forceinline__ void m3mulv(const double* A, const double x[3], double y[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}

// rel = T_i^-1 * T_last (calcRelaSwPose, :70-88)
__device__ __forceinline__ void rel_pose(const double* Ri, const double* pi, const double* Rl, const double* pl, double Rr[9],
                                         double tr[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rr[3 * i + j] = Ri[i] * Rl[j] + Ri[3 + i] * Rl[3 + j] + Ri[6 + i] * Rl[6 + j];      // Ri^T Rl
    const double d[3] = { pl[0] - pi[0], pl[1] - pi[1], pl[2] - pi[2] };
#pragma unroll
    for (int i = 0; i < 3; ++i) tr[i] = Ri[i] * d[0] + Ri[3 + i] * d[1] + Ri[6 + i] * d[2];
}

__device__ __forceinline__ void eye_pose(double Rr[9], double tr[3])
{
#pragma unroll
    for (int i = 0; i < 9; ++i) Rr[i] = (i % 4 == 0) ? 1.0 : 0.0;
    tr[0] = tr[1] = tr[2] = 0.0;
}

// calcUnitCost (:107-125)
__device__ __forceinline__ double unit_cost(double mx, double my, const double Rr[9], const double tr[3], const double sol[3])
{
    double pf0[3], pf[3];
    pf0[2] = 1.0 / sol[2];
    pf0[0] = sol[0] * pf0[2];
    pf0[1] = sol[1] * pf0[2];
    m3mulv(Rr, pf0, pf);
    pf[0] += tr[0]; pf[1] += tr[1]; pf[2] += tr[2];
    const double ex = mx - pf[0] / pf[2], ey = my - pf[1] / pf[2];
    return ex * ex + ey * ey;
}

// (A + lambda I) x = b, symmetric 3x3 LDL^T (:232)
__device__ __forceinline__ void solve3(const double A[9], double lambda, const double b[3], double x[3])
{
    const double a00 = A[0] + lambda, a10 = A[3], a20 = A[6], a11 = A[4] + lambda, a21 = A[7], a22 = A[8] + lambda;
    const double l10 = a10 / a00, l20 = a20 / a00;
    const double d1 = a11 - l10 * a10;
    const double l21 = (a21 - l20 * a10) / d1;
    const double d2 = a22 - l20 * a20 - l21 * (a21 - l20 * a10);
    const double y0 = b[0], y1 = b[1] - l10 * y0, y2 = b[2] - l20 * y0 - l21 * y1;
    x[2] = y2 / d2;
    x[1] = y1 / d1 - l21 * x[2];
    x[0] = y0 / a00 - l10 * x[1] - l20 * x[2];
}

#define TRI_CMAX 64

template <bool STEREO>
__global__ __launch_bounds__(WAVE) void k_triangulate(TriLaunch L)
{
    constexpr int EYES = STEREO ? 2 : 1;
    __shared__ double sR[EYES][TRI_CMAX][9];
    __shared__ double sP[EYES][TRI_CMAX][3];
    const int bl = blockIdx.y, b = L.b0 + bl, tid = threadIdx.x;
    const FrameView& fv = L.fv;
    const int C = fv.n_clones[b], F = fv.n_feat[b];
    // camera poses of the window: left as given, right = left * T_cl2cr^-1 (:349-354)
    for (int s = tid; s < C; s += WAVE) {
        const double* R = fv.clone_R + ((size_t)b * fv.cmax + s) * 9;
        const double* p = fv.clone_p + ((size_t)b * fv.cmax + s) * 3;
#pragma unroll
        for (int i = 0; i < 9; ++i) sR[0][s][i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) sP[0][s][i] = p[i];
        if (STEREO) {
            double tinv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) tinv[i] = -(L.R_lr[i] * L.t_lr[0] + L.R_lr[3 + i] * L.t_lr[1] + L.R_lr[6 + i] * L.t_lr[2]);   // -R_lr^T t
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) sR[EYES - 1][s][3 * i + j] = R[3 * i] * L.R_lr[3 * j] + R[3 * i + 1] * L.R_lr[3 * j + 1] + R[3 * i + 2] * L.R_lr[3 * j + 2];   // R R_lr^T
                sP[EYES - 1][s][i] = p[i] + (R[3 * i] * tinv[0] + R[3 * i + 1] * tinv[1] + R[3 * i + 2] * tinv[2]);
            }
        }
    }
    __syncthreads();
    const int j = blockIdx.x * WAVE + tid;
    if (j >= F) return;
    const size_t oidx = (size_t)b * fv.fmax + j;
    const unsigned long long mask = fv.obs_mask[oidx] & (C >= 64 ? ~0ULL : ((1ULL << C) - 1ULL));
    const double* uv = fv.uv + oidx * fv.cmax * 4;
    double* pf_out = L.pf + oidx * 3;
    auto fail = [&]() {
        pf_out[0] = pf_out[1] = pf_out[2] = 0.0;
        L.ok[oidx] = 0;
        if (L.mask_failed) L.mask_rw[oidx] = 0ULL;
    };
    const int n = EYES * __popcll(mask);
    if (n <= 4) { fail(); return; }                                                 // :183-187
    const int s_last = 63 - __clzll((long long)mask);                              // newest observing clone; its LAST eye
    const double* Rl = sR[EYES - 1][s_last];
    const double* pl = sP[EYES - 1][s_last];
    const double ml0 = uv[4 * s_last + 2 * (EYES - 1)], ml1 = uv[4 * s_last + 2 * (EYES - 1) + 1];
    // findLongestTrans (:31-68)
    int smax = s_last, emax = EYES - 1;
    {
        double fl[3] = { ml0, ml1, 1.0 };
        const double fn = sqrt(fl[0] * fl[0] + fl[1] * fl[1] + fl[2] * fl[2]);
        fl[0] /= fn; fl[1] /= fn; fl[2] /= fn;
        double fw[3];
        m3mulv(Rl, fl, fw);
        double max_trans = -__builtin_inf();
        for (int s = 0; s <= s_last; ++s) {
            if (!((mask >> s) & 1ULL)) continue;
#pragma unroll
            for (int e = 0; e < EYES; ++e) {
                if (s == s_last && e == EYES - 1) continue;
                const double* pi = sP[e][s];
                const double d[3] = { pi[0] - pl[0], pi[1] - pl[1], pi[2] - pl[2] };
                const double dot = fw[0] * d[0] + fw[1] * d[1] + fw[2] * d[2];
                const double q[3] = { d[0] - fw[0] * dot, d[1] - fw[1] * dot, d[2] - fw[2] * dot };
                const double tr = fabs(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]));
                if (tr > max_trans) { max_trans = tr; smax = s; emax = e; }
            }
        }
        if (max_trans < L.trans_thres) { fail(); return; }                          // :192
    }
    // initial guess (:201-203, initDepth :90-105)
    double sol[3];
    {
        double Rr[9], tr[3], tm[3];
        rel_pose(sR[emax][smax], sP[emax][smax], Rl, pl, Rr, tr);
        const double m1[3] = { ml0, ml1, 1.0 };
        m3mulv(Rr, m1, tm);
        const double m20 = uv[4 * smax + 2 * emax], m21 = uv[4 * smax + 2 * emax + 1];
        const double A0 = tm[0] - m20 * tm[2], A1 = tm[1] - m21 * tm[2];
        const double b0 = m20 * tr[2] - tr[0], b1 = m21 * tr[2] - tr[1];
        const double depth = (A0 * b0 + A1 * b1) / (A0 * A0 + A1 * A1);
        sol[0] = ml0; sol[1] = ml1; sol[2] = 1.0 / depth;
    }
    auto total_cost_of = [&](const double s3[3]) {                                  // calcTotalCost :127-137
        double c = 0.0;
        for (int s = 0; s <= s_last; ++s) {
            if (!((mask >> s) & 1ULL)) continue;
#pragma unroll
            for (int e = 0; e < EYES; ++e) {
                double Rr[9], tr[3];
                if (s == s_last && e == EYES - 1) eye_pose(Rr, tr); else rel_pose(sR[e][s], sP[e][s], Rl, pl, Rr, tr);
                c += unit_cost(uv[4 * s + 2 * e], uv[4 * s + 2 * e + 1], Rr, tr, s3);
            }
        }
        return c;
    };
    double total_cost = total_cost_of(sol);
    double lambda = L.init_damping, delta_norm = __builtin_inf();
    int inner = 0, outer = 0;
    bool reduced = false;
    do {                                                                             // :215-262
        double A[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, bv[3] = { 0, 0, 0 };
        for (int s = 0; s <= s_last; ++s) {
            if (!((mask >> s) & 1ULL)) continue;
#pragma unroll
            for (int e = 0; e < EYES; ++e) {                                        // calcResJacobian :139-171
                double Rr[9], tr[3];
                if (s == s_last && e == EYES - 1) eye_pose(Rr, tr); else rel_pose(sR[e][s], sP[e][s], Rl, pl, Rr, tr);
                const double a3[3] = { sol[0], sol[1], 1.0 };
                double h[3];
                m3mulv(Rr, a3, h);
                h[0] += tr[0] * sol[2]; h[1] += tr[1] * sol[2]; h[2] += tr[2] * sol[2];
                const double res0 = h[0] / h[2] - uv[4 * s + 2 * e], res1 = h[1] / h[2] - uv[4 * s + 2 * e + 1];
                const double W00 = 1.0 / h[2], W02 = -h[0] / (h[2] * h[2]), W12 = -h[1] / (h[2] * h[2]);
                double J[6];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double u0 = c < 2 ? Rr[c] : tr[0], u1 = c < 2 ? Rr[3 + c] : tr[1], u2 = c < 2 ? Rr[6 + c] : tr[2];
                    J[c] = W00 * u0 + W02 * u2;
                    J[3 + c] = W00 * u1 + W12 * u2;
                }
                const double en = sqrt(res0 * res0 + res1 * res1);
                const double w = en <= L.huber_epsilon ? 1.0 : sqrt(2.0 * L.huber_epsilon / en);
                const double w2 = w == 1.0 ? 1.0 : w * w;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) A[3 * r + c] += w2 * (J[r] * J[c] + J[3 + r] * J[3 + c]);
                    bv[r] -= w2 * (J[r] * res0 + J[3 + r] * res1);
                }
            }
        }
        do {
            double delta[3], ns[3];
            solve3(A, lambda, bv, delta);
            ns[0] = sol[0] + delta[0]; ns[1] = sol[1] + delta[1]; ns[2] = sol[2] + delta[2];
            delta_norm = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
            const double nc = total_cost_of(ns);
            if (nc < total_cost) {
                total_cost = nc; sol[0] = ns[0]; sol[1] = ns[1]; sol[2] = ns[2]; reduced = true;
                lambda = lambda / 10.0 > 1e-10 ? lambda / 10.0 : 1e-10;
            } else {
                reduced = false;
                lambda = lambda * 10 < 1e12 ? lambda * 10 : 1e12;
            }
        } while (inner++ < L.inner_loop_max_iter && !reduced);
        inner = 0;                                                                   // :259 (reset: see the oracle's note)
    } while (outer++ < L.outer_loop_max_iter && delta_norm > L.conv_precision);
    double plast[3];
    plast[2] = 1.0 / sol[2]; plast[0] = sol[0] * plast[2]; plast[1] = sol[1] * plast[2];
    if ((outer >= L.outer_loop_max_iter && inner >= L.inner_loop_max_iter) || delta_norm > L.conv_precision) { fail(); return; }
    for (int s = 0; s <= s_last; ++s) {                                              // :273-278
        if (!((mask >> s) & 1ULL)) continue;
#pragma unroll
        for (int e = 0; e < EYES; ++e) {
            double Rr[9], tr[3], t3[3];
            if (s == s_last && e == EYES - 1) eye_pose(Rr, tr); else rel_pose(sR[e][s], sP[e][s], Rl, pl, Rr, tr);
            m3mulv(Rr, plast, t3);
            if (t3[2] + tr[2] <= L.min_depth) { fail(); return; }
        }
    }
    if (plast[2] < L.min_depth || plast[2] > L.max_depth) { fail(); return; }        // :296-297
    double w3[3];
    m3mulv(Rl, plast, w3);
    w3[0] += pl[0]; w3[1] += pl[1]; w3[2] += pl[2];
    if (w3[0] != w3[0] || w3[1] != w3[1] || w3[2] != w3[2]) { fail(); return; }
    pf_out[0] = w3[0]; pf_out[1] = w3[1]; pf_out[2] = w3[2];
    L.ok[oidx] = 1;
}

}  // namespace

int launch_triangulate(const TriLaunch& L, int nb, int fmax_used, int stereo, hipStream_t st)
{
    if (L.fv.cmax > TRI_CMAX) return -1;
    const dim3 grid((fmax_used + WAVE - 1) / WAVE, nb);
    if (stereo) hipLaunchKernelGGL(k_triangulate<true>, grid, dim3(WAVE), 0, st, L);
    else hipLaunchKernelGGL(k_triangulate<false>, grid, dim3(WAVE), 0, st, L);
    return 0;
}
