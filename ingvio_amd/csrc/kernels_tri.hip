// kernels_tri.hip — SURVEY.md §8(f) row f-1: batched feature triangulation on the device.
// Triangulator::triangulateMonoObs (Triangulator.cpp:173-318) and its stereo wrapper (:320-359): Levenberg-Marquardt on
// (x/z, y/z, 1/z) in the frame of the LAST mono-equivalent observation, Huber-weighted normal equations, accept test
// on the unweighted cost, depth / parallax / convergence gates.  One DPP QUAD per feature (one DPP ROW of 16 lanes when the call
// holds few features, round 5): the iteration is sequential, but each cost / normal-equation pass is a sum over observations, dealt
// round-robin to the lanes of the group and reduced with lane exchanges; 16 (4) features per wave, 32 (8) per workgroup.  The window's
// camera poses (left, and right = left * T_cl2cr^-1) are staged once per workgroup in LDS.  Per-observation arithmetic follows the
// restatement in oracle/ingvio_oracle.c; only the order of the sums over observations differs (4 or 16 partial sums), which the parity
// tolerance (5e-7 relative) covers.  gfx950 only.
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

// calcUnitCost (:107-125); pf0 = (x/z, y/z, 1)/rho is the same for every observation of a pass and comes in precomputed.
// One reciprocal instead of the two divisions of the reference (last-bit rounding only, inside the parity tolerance).
__device__ __forceinline__ double unit_cost(double mx, double my, const double Rr[9], const double tr[3], const double pf0[3])
{
    double pf[3];
    m3mulv(Rr, pf0, pf);
    pf[0] += tr[0]; pf[1] += tr[1]; pf[2] += tr[2];
    const double r = fast_rcp(pf[2]);
    const double ex = mx - pf[0] * r, ey = my - pf[1] * r;
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
#define TRI_NT 128                     // 32 features per workgroup at 4 lanes per feature, 8 at 16
#define TRI_NOBS 6                     // register path at 4 lanes per feature: up to 24 mono-equivalent observations per feature
#define TRI_NOBS16 4                   // ... at 16 lanes per feature: up to 64
#define TRI_FEW 2048                   // up to this many features in a call: 16 lanes per feature

// quad_perm DPP: full-rate lane exchange inside a quad (0xB1 = lanes ^1, 0x4E = lanes ^2)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_quad_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
// Reductions over the LPF lanes of a feature (4: a DPP quad, 16: a DPP row).  Every step is an EXCHANGE between two lanes (quad_perm
// lanes ^1 / ^2, then the mirror images inside the half row and the row: lane i with 7 - i, lane i with 15 - i), a + b is commutative,
// so all lanes of the group end with the SAME bits and the group's control flow stays uniform - a rotation (row_ror) would leave the
// four quads of a row with differently associated sums, i.e. with different last bits and, once in a while, different branches.
template <int LPF>
__device__ __forceinline__ double grp_sum(double x)
{
    x += dpp_quad<0xB1>(x);
    x += dpp_quad<0x4E>(x);
    if (LPF == 16) {
        x += dpp_quad<0x141>(x);                                         // row_half_mirror
        x += dpp_quad<0x140>(x);                                         // row_mirror
    }
    return x;
}
template <int LPF>
__device__ __forceinline__ int grp_or(int x)
{
    x |= dpp_quad_i<0xB1>(x);
    x |= dpp_quad_i<0x4E>(x);
    if (LPF == 16) {
        x |= dpp_quad_i<0x141>(x);
        x |= dpp_quad_i<0x140>(x);
    }
    return x;
}
// the larger value wins, the lower key among equal values: a symmetric rule, both lanes of an exchange pick the same winner
template <int CTRL>
__device__ __forceinline__ void max_exchange(double& v, int& key)
{
    const double ov = dpp_quad<CTRL>(v); const int ok_ = dpp_quad_i<CTRL>(key);
    if (ov > v || (ov == v && ok_ < key)) { v = ov; key = ok_; }
}

// The mono-equivalent observations of a feature in the reference's order (clone ascending, left eye before right) are
// dealt round-robin to the four lanes of a quad: lane q owns observations q, q+4, q+8, ...  Stereo: that is every second
// observing clone starting at the (q>>1)-th, always eye q&1; mono: every fourth starting at the q-th.  (LPF lanes per feature: every
// LPF / eyes-th observing clone.)
template <bool STEREO, int LPF>
struct ObsCursor {
    unsigned long long m;
    __device__ __forceinline__ ObsCursor(unsigned long long mask, int q)
    {
        m = mask;
        const int skip = STEREO ? (q >> 1) : q;
#pragma unroll
        for (int i = 0; i < (STEREO ? LPF / 2 : LPF) - 1; ++i)
            if (i < skip) m &= m - 1ULL;
    }
    __device__ __forceinline__ bool valid() const { return m != 0ULL; }
    __device__ __forceinline__ int slot() const { return __ffsll((long long)m) - 1; }
    __device__ __forceinline__ void next()
    {
#pragma unroll
        for (int i = 0; i < (STEREO ? LPF / 2 : LPF); ++i) m &= m - 1ULL;
    }
};

// NOBS > 0: a lane keeps its (at most NOBS) observations - clone slot and measurement - in registers and every pass is a
// fully unrolled, branch-free loop over them (independent observations interleave, nothing is re-read from global
// memory).  NOBS == 0: any number of observations, walked with the cursor and read from global memory in every pass.
// LPF lanes per feature: 4 for throughput (16 features per wave; every lane of a group repeats the scalar part of the iteration - the
// 3 x 3 solves, the acceptance tests), 16 when there are few features and the LATENCY of one feature's iteration is what the caller
// waits for (a single real-time filter: 22-54 observations dealt to 16 lanes instead of 4 - a pass over them is 2-4 observations
// per lane instead of 6-14).
template <bool STEREO, int NOBS, int LPF>
__global__ __launch_bounds__(TRI_NT) void k_triangulate(TriLaunch L)
{
    constexpr int EYES = STEREO ? 2 : 1;
    constexpr int NREG = NOBS > 0 ? NOBS : 1;
    __shared__ double sR[EYES][TRI_CMAX][9];
    __shared__ double sP[EYES][TRI_CMAX][3];
    const int bl = blockIdx.y, b = L.b0 + bl, tid = threadIdx.x;
    const FrameView& fv = L.fv;
    const int C = fv.n_clones[b], F = fv.n_feat[b];
    // camera poses of the window: left as given, right = left * T_cl2cr^-1 (:349-354)
    for (int s = tid; s < C; s += TRI_NT) {
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
    const int q = tid & (LPF - 1), eq = STEREO ? (q & 1) : 0;
    const int j = blockIdx.x * (TRI_NT / LPF) + tid / LPF;
    if (j >= F) return;                                                             // whole quads leave together
    const size_t oidx = (size_t)b * fv.fmax + j;
    const unsigned long long mask = (L.tri_mask ? L.tri_mask : fv.obs_mask)[oidx] & (C >= 64 ? ~0ULL : ((1ULL << C) - 1ULL));
    const double* uv = fv.uv + oidx * fv.cmax * 4;
    double* pf_out = L.pf + oidx * 3;
    auto fail = [&](int code = 0) {
        if (q == 0) {
            pf_out[0] = pf_out[1] = pf_out[2] = 0.0;
            L.ok[oidx] = code;
            if (L.mask_failed) L.mask_rw[oidx] = 0ULL;
            if (L.ok2) { L.ok2[oidx] = code; L.pf2[oidx * 3] = L.pf2[oidx * 3 + 1] = L.pf2[oidx * 3 + 2] = 0.0; }
        }
    };
    const int n = EYES * __popcll(mask);
    if (n <= 4) { fail(); return; }                                                 // :183-187
    const int s_last = 63 - __clzll((long long)mask);                              // newest observing clone; its LAST eye
    const double* Rl = sR[EYES - 1][s_last];
    const double* pl = sP[EYES - 1][s_last];
    const double ml0 = uv[4 * s_last + 2 * (EYES - 1)], ml1 = uv[4 * s_last + 2 * (EYES - 1) + 1];
    int sl[NREG], cnt = 0;
    double mu[NREG], mv[NREG];
    if (NOBS > 0) {
        ObsCursor<STEREO, LPF> oc(mask, q);
#pragma unroll
        for (int t = 0; t < NREG; ++t) {
            const bool v = oc.valid();
            const int s = v ? oc.slot() : s_last;                                  // padding entries: a valid address, result masked
            sl[t] = s; mu[t] = uv[4 * s + 2 * eq]; mv[t] = uv[4 * s + 2 * eq + 1];
            cnt += v ? 1 : 0;
            oc.next();
        }
    }
    // body(s, u, v, t, live): live == false only for padding entries of the register path (compute, then discard)
    auto each_obs = [&](auto&& body) {
        if constexpr (NOBS > 0) {
#pragma unroll
            for (int t = 0; t < NREG; ++t) {
                int s = sl[t];
                asm volatile("" : "+v"(s));        // opaque per pass: keeps the 6 x 12 relative-pose values out of loop-invariant registers
                if (t < cnt) body(s, mu[t], mv[t], t, true);
            }
        } else {
            int t = 0;
            for (ObsCursor<STEREO, LPF> oc(mask, q); oc.valid(); oc.next(), ++t) {
                const int s = oc.slot();
                body(s, uv[4 * s + 2 * eq], uv[4 * s + 2 * eq + 1], t, true);
            }
        }
    };
    // relative pose of observation (s, eq) to the last one; the last one itself is the exact identity (:215-221)
    auto rel = [&](int s, double Rr[9], double tr[3]) {
        if (s == s_last && eq == EYES - 1) eye_pose(Rr, tr); else rel_pose(sR[eq][s], sP[eq][s], Rl, pl, Rr, tr);
    };
    // findLongestTrans (:31-68): first maximum in observation order = maximum with the lowest observation number
    int smax, emax;
    {
        double fl[3] = { ml0, ml1, 1.0 };
        const double fn = sqrt(fl[0] * fl[0] + fl[1] * fl[1] + fl[2] * fl[2]);
        fl[0] /= fn; fl[1] /= fn; fl[2] /= fn;
        double fw[3];
        m3mulv(Rl, fl, fw);
        double max_trans = -__builtin_inf();
        int kmax = 0x7fffffff;
        each_obs([&](int s, double, double, int t, bool live) {
            const int k = q + LPF * t;
            const double* pi = sP[eq][s];
            const double d[3] = { pi[0] - pl[0], pi[1] - pl[1], pi[2] - pl[2] };
            const double dot = fw[0] * d[0] + fw[1] * d[1] + fw[2] * d[2];
            const double qv[3] = { d[0] - fw[0] * dot, d[1] - fw[1] * dot, d[2] - fw[2] * dot };
            const double tr = fabs(sqrt(qv[0] * qv[0] + qv[1] * qv[1] + qv[2] * qv[2]));
            if (live && !(s == s_last && eq == EYES - 1) && tr > max_trans) { max_trans = tr; kmax = (k << 8) | (s << 1) | eq; }
        });
        {
            max_exchange<0xB1>(max_trans, kmax);
            max_exchange<0x4E>(max_trans, kmax);
            if (LPF == 16) { max_exchange<0x141>(max_trans, kmax); max_exchange<0x140>(max_trans, kmax); }
        }
        if (max_trans < L.trans_thres) { fail(); return; }                          // :192
        smax = (kmax >> 1) & 127; emax = kmax & 1;
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
        double c = 0.0, pf0[3];
        pf0[2] = 1.0 / s3[2];
        pf0[0] = s3[0] * pf0[2];
        pf0[1] = s3[1] * pf0[2];
        each_obs([&](int s, double u, double v, int, bool live) {
            double Rr[9], tr[3];
            rel(s, Rr, tr);
            const double uc = unit_cost(u, v, Rr, tr, pf0);
            c += live ? uc : 0.0;
        });
        return grp_sum<LPF>(c);
    };
    const double eps2 = L.huber_epsilon * L.huber_epsilon, two_eps = 2.0 * L.huber_epsilon;
    double total_cost = total_cost_of(sol);
    double lambda = L.init_damping, delta_norm = __builtin_inf();
    int inner = 0, outer = 0;
    bool reduced = false;
    do {                                                                             // :215-262
        double A[6] = { 0, 0, 0, 0, 0, 0 }, bv[3] = { 0, 0, 0 };                     // A: 00 01 02 11 12 22
        each_obs([&](int s, double mu_, double mv_, int, bool live) {                // calcResJacobian :139-171
            double Rr[9], tr[3];
            rel(s, Rr, tr);
            const double a3[3] = { sol[0], sol[1], 1.0 };
            double h[3];
            m3mulv(Rr, a3, h);
            h[0] += tr[0] * sol[2]; h[1] += tr[1] * sol[2]; h[2] += tr[2] * sol[2];
            // one reciprocal for the five divisions by h_z / h_z^2 of :150-158
            const double W00 = fast_rcp(h[2]), hx = h[0] * W00, hy = h[1] * W00;
            const double res0 = hx - mu_, res1 = hy - mv_;
            const double W02 = -hx * W00, W12 = -hy * W00;
            double J[6];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double u0 = c < 2 ? Rr[c] : tr[0], u1 = c < 2 ? Rr[3 + c] : tr[1], u2 = c < 2 ? Rr[6 + c] : tr[2];
                J[c] = W00 * u0 + W02 * u2;
                J[3 + c] = W00 * u1 + W12 * u2;
            }
            // Huber (:161-166): w = sqrt(2 eps / |res|) enters only as w^2 = 2 eps / |res|
            const double e2 = res0 * res0 + res1 * res1;
            const double w2 = !live ? 0.0 : (e2 <= eps2 ? 1.0 : two_eps * rsqrt(e2));
            int t = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = r; c < 3; ++c) A[t++] += w2 * (J[r] * J[c] + J[3 + r] * J[3 + c]);
                bv[r] -= w2 * (J[r] * res0 + J[3 + r] * res1);
            }
        });
#pragma unroll
        for (int t = 0; t < 6; ++t) A[t] = grp_sum<LPF>(A[t]);
#pragma unroll
        for (int t = 0; t < 3; ++t) bv[t] = grp_sum<LPF>(bv[t]);
        const double Af[9] = { A[0], A[1], A[2], A[1], A[3], A[4], A[2], A[4], A[5] };
        do {
            double delta[3], ns[3];
            solve3(Af, lambda, bv, delta);
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
    int behind = 0;
    each_obs([&](int s, double, double, int, bool live) {                            // :273-278
        double Rr[9], tr[3], t3[3];
        rel(s, Rr, tr);
        m3mulv(Rr, plast, t3);
        behind |= (live && t3[2] + tr[2] <= L.min_depth) ? 1 : 0;
    });
    if (grp_or<LPF>(behind)) { fail(); return; }
    if (plast[2] < L.min_depth || plast[2] > L.max_depth) { fail(); return; }        // :296-297
    double w3[3];
    m3mulv(Rl, plast, w3);
    w3[0] += pl[0]; w3[1] += pl[1]; w3[2] += pl[2];
    if (w3[0] != w3[0] || w3[1] != w3[1] || w3[2] != w3[2]) { fail(); return; }
    if (L.check_anchor) {                                                           // MapServerManager.cpp:290,325: depth in the anchor's (left) camera
        const int a = fv.anchor[oidx];
        if (a < 0 || a >= C) { fail(2); return; }
        const double* Ra = sR[0][a];
        const double* pa = sP[0][a];
        const double z = Ra[2] * (w3[0] - pa[0]) + Ra[5] * (w3[1] - pa[1]) + Ra[8] * (w3[2] - pa[2]);      // (R_a^T (p_f - p_a)).z
        if (z <= 0.0) { fail(2); return; }
    }
    if (q == 0) {
        pf_out[0] = w3[0]; pf_out[1] = w3[1]; pf_out[2] = w3[2];
        L.ok[oidx] = 1;
        if (L.ok2) { L.ok2[oidx] = 1; L.pf2[oidx * 3] = w3[0]; L.pf2[oidx * 3 + 1] = w3[1]; L.pf2[oidx * 3 + 2] = w3[2]; }
    }
}

}  // namespace

int launch_triangulate(const TriLaunch& L, int nb, int fmax_used, int stereo, hipStream_t st)
{
    if (L.fv.cmax > TRI_CMAX) return -1;
    const int eyes = stereo ? 2 : 1;
    if ((long long)nb * fmax_used <= TRI_FEW) {                                       // few features: 16 lanes each (latency)
        const int fpb = TRI_NT / 16;
        const dim3 grid((fmax_used + fpb - 1) / fpb, nb);
        const bool regs = eyes * L.fv.cmax <= 16 * TRI_NOBS16;
        if (stereo) {
            if (regs) hipLaunchKernelGGL((k_triangulate<true, TRI_NOBS16, 16>), grid, dim3(TRI_NT), 0, st, L);
            else hipLaunchKernelGGL((k_triangulate<true, 0, 16>), grid, dim3(TRI_NT), 0, st, L);
        } else {
            if (regs) hipLaunchKernelGGL((k_triangulate<false, TRI_NOBS16, 16>), grid, dim3(TRI_NT), 0, st, L);
            else hipLaunchKernelGGL((k_triangulate<false, 0, 16>), grid, dim3(TRI_NT), 0, st, L);
        }
        return 0;
    }
    const int fpb = TRI_NT / 4;
    const dim3 grid((fmax_used + fpb - 1) / fpb, nb);
    const bool regs = eyes * L.fv.cmax <= 4 * TRI_NOBS;                               // every lane's share fits the register path
    if (stereo) {
        if (regs) hipLaunchKernelGGL((k_triangulate<true, TRI_NOBS, 4>), grid, dim3(TRI_NT), 0, st, L);
        else hipLaunchKernelGGL((k_triangulate<true, 0, 4>), grid, dim3(TRI_NT), 0, st, L);
    } else {
        if (regs) hipLaunchKernelGGL((k_triangulate<false, TRI_NOBS, 4>), grid, dim3(TRI_NT), 0, st, L);
        else hipLaunchKernelGGL((k_triangulate<false, 0, 4>), grid, dim3(TRI_NT), 0, st, L);
    }
    return 0;
}
