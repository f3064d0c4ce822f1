// feat_build.h — K3 per-observation Jacobian pieces + K4 nullspace reflectors, shared by the dense
// (kernels_msckf.hip) and the factored (kernels_factored.hip) MSCKF paths.  gfx950 only.
#pragma once
#include "dev_common.h"

template <int CMAX, bool STEREO>
struct FeatCfg {
    static constexpr int RPO = STEREO ? 4 : 2;            // rows per observation
    static constexpr int RR = RPO * CMAX;                 // register rows per lane
    static constexpr int RRH = RR - 3;                    // rows after the nullspace projection
    static constexpr int NCOLMAX = 6 * CMAX;
    static constexpr int NT = ((NCOLMAX + 1 + 63) / 64) * 64;
};

// LDS scratch of the per-feature builder.  LEAN drops what only the dense path needs (G [p_f]x, the nullspace
// reflectors, the per-column index table): the factored gate kernel's occupancy is bounded by LDS.
template <int CMAX, bool STEREO, bool LEAN>
struct FeatDenseOnly {
    using Cfg = FeatCfg<CMAX, STEREO>;
    double GX[CMAX][Cfg::RPO][3];     // (Pi~ * R^T) [p_f]x
    double V[3][Cfg::RR];             // the three nullspace reflectors
    double tau[3];
};
template <int CMAX, bool STEREO>
struct FeatDenseOnly<CMAX, STEREO, true> {};

template <int CMAX, bool STEREO, bool LEAN = false>
struct FeatShared : FeatDenseOnly<CMAX, STEREO, LEAN> {
    using Cfg = FeatCfg<CMAX, STEREO>;
    static constexpr bool kLean = LEAN;
    double G[CMAX][Cfg::RPO][3];      // Pi~ * R^T per observation  (also the Hf rows)
    double res[CMAX][Cfg::RPO];
    int slot[CMAX];                   // window slot of dense observation o
    int gidx[LEAN ? CMAX : Cfg::NCOLMAX];   // state index of every column (LEAN: of every clone's first column)
    int nobs;
};

// One observation of one feature: q = R^T(p_f - p), projection Jacobians G = Pi~ R^T (RPO x 3) and the
// residual (RemoveLostUpdate.cpp:435-506).  Returns false when the reference would skip it (NaN guard, :486).
template <bool STEREO>
__device__ __forceinline__ bool feat_obs(const double* R, const double* p, const double z[4], double pfx, double pfy,
                                         double pfz, const MsckfOpts& op, double (*Gm)[3], double* rs)
{
    constexpr int RPO = STEREO ? 4 : 2;
    const double dx = pfx - p[0], dy = pfy - p[1], dz = pfz - p[2];
    double q[3], Rt[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        q[i] = R[i] * dx + R[3 + i] * dy + R[6 + i] * dz;                 // R^T (p_f - p), :448
        Rt[3 * i] = R[i]; Rt[3 * i + 1] = R[3 + i]; Rt[3 * i + 2] = R[6 + i];
    }
    // reciprocal by v_rcp_f64 + Newton (~1 ulp); q_z == 0 keeps the IEEE result so the guard below sees inf/NaN as the reference does
    const double iz = q[2] == 0.0 ? __builtin_copysign(__builtin_inf(), q[2]) : fast_rcp(q[2]);
    const double hp02 = -q[0] * (iz * iz), hp12 = -q[1] * (iz * iz);   // :452-456
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        Gm[0][m] = iz * Rt[m] + hp02 * Rt[6 + m];
        Gm[1][m] = iz * Rt[3 + m] + hp12 * Rt[6 + m];
    }
    rs[0] = z[0] - q[0] * iz;
    rs[1] = z[1] - q[1] * iz;
    bool nan = (iz != iz) || (hp02 != hp02) || (hp12 != hp12);                 // :486
#pragma unroll
    for (int i = 0; i < 9; ++i) nan |= (R[i] != R[i]);
    nan |= (pfx != pfx) || (pfy != pfy) || (pfz != pfz);
    if (STEREO) {
        double qr[3], M[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            qr[i] = op.R_lr[3 * i] * q[0] + op.R_lr[3 * i + 1] * q[1] + op.R_lr[3 * i + 2] * q[2] + op.t_lr[i];   // :450
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m)
                M[3 * i + m] = op.R_lr[3 * i] * Rt[m] + op.R_lr[3 * i + 1] * Rt[3 + m] + op.R_lr[3 * i + 2] * Rt[6 + m];
        const double izr = qr[2] == 0.0 ? __builtin_copysign(__builtin_inf(), qr[2]) : fast_rcp(qr[2]);
        const double h02 = -qr[0] * (izr * izr), h12 = -qr[1] * (izr * izr);     // :458-462
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            Gm[RPO - 2][m] = izr * M[m] + h02 * M[6 + m];
            Gm[RPO - 1][m] = izr * M[3 + m] + h12 * M[6 + m];
        }
        rs[RPO - 2] = z[2] - qr[0] * izr;
        rs[RPO - 1] = z[3] - qr[1] * izr;                                               // :503
    }
    return !nan;
}

// phase 1 (wave 0, one lane per window slot): q = R^T(p_f - p), projection Jacobians, residuals.
// Fills sh.G (= Pi~ R^T, also the rows of Hf), sh.GX (= G [p_f]x), sh.res, sh.slot, sh.nobs.
// RemoveLostUpdate.cpp:435-506.  Ends with a workgroup barrier.  Returns rows = RPO * nobs.
template <int CMAX, bool STEREO, bool LEAN = false>
__device__ __forceinline__ int feat_phase1(const FrameView& fv, const MsckfOpts& op, int b, int j, int C,
                                           FeatShared<CMAX, STEREO, LEAN>& sh)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RPO = Cfg::RPO;
    const int tid = threadIdx.x;
    const double* pf = fv.pf + ((size_t)b * fv.fmax + j) * 3;
    const double pfx = pf[0], pfy = pf[1], pfz = pf[2];
    const unsigned long long mask = fv.obs_mask[(size_t)b * fv.fmax + j];
    // ---- phase 1: one lane per window slot (wave 0) --------------------------------------
    if (tid < WAVE) {
        const int s = tid;
        bool valid = false;
        double Gm[RPO][3], rs[RPO];
        if (s < C && ((mask >> s) & 1ULL)) {
            const double* R = fv.clone_R + ((size_t)b * fv.cmax + s) * 9;
            const double* p = fv.clone_p + ((size_t)b * fv.cmax + s) * 3;
            const double* z = fv.uv + (((size_t)b * fv.fmax + j) * fv.cmax + s) * 4;
            double zz[4] = { z[0], z[1], 0.0, 0.0 };
            if (STEREO) { zz[2] = z[2]; zz[3] = z[3]; }
            valid = feat_obs<STEREO>(R, p, zz, pfx, pfy, pfz, op, Gm, rs);
        }
        const unsigned long long vm = __ballot(valid);
        if (valid) {
            const int od = __popcll(vm & ((1ULL << s) - 1ULL));
            sh.slot[od] = s;
#pragma unroll
            for (int t = 0; t < RPO; ++t) {
#pragma unroll
                for (int m = 0; m < 3; ++m) sh.G[od][t][m] = Gm[t][m];
                if constexpr (!LEAN) {
                    sh.GX[od][t][0] = Gm[t][1] * pfz - Gm[t][2] * pfy;       // G * skew(p_f)
                    sh.GX[od][t][1] = -Gm[t][0] * pfz + Gm[t][2] * pfx;
                    sh.GX[od][t][2] = Gm[t][0] * pfy - Gm[t][1] * pfx;
                }
                sh.res[od][t] = rs[t];
            }
        }
        if (tid == 0) sh.nobs = __popcll(vm);
    }
    __syncthreads();
    return RPO * sh.nobs;
}

// phase 2 (wave 0): Householder QR of Hf (rows x 3) in registers -> the three reflectors
// sh.V[k][.], sh.tau[k] whose product's last rows-3 columns span the left nullspace
// (RemoveLostUpdate.cpp:518-522 uses JacobiSVD's full U; any orthonormal basis is equivalent).
// Ends with a workgroup barrier.
template <int CMAX, bool STEREO>
__device__ __forceinline__ void feat_phase2(FeatShared<CMAX, STEREO>& sh, int rows)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RPO = Cfg::RPO, RR = Cfg::RR;
    const int tid = threadIdx.x;
    // ---- phase 2: QR of Hf (rows x 3) in wave 0's registers -> 3 reflectors ---------------
    if (tid < WAVE) {
        constexpr int PER = (RR + WAVE - 1) / WAVE;
        double hf[3][PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * WAVE;
#pragma unroll
            for (int m = 0; m < 3; ++m) hf[m][u] = (i < rows) ? sh.G[i / RPO][i % RPO][m] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double part = 0.0;
#pragma unroll
            for (int u = 0; u < PER; ++u) { const int i = tid + u * WAVE; if (i >= k) part += hf[k][u] * hf[k][u]; }
            const double nrm = sqrt(wave_sum(part));
            const double x0 = __shfl(hf[k][0], k, WAVE);
            double tau = 0.0, v0 = 1.0;
            if (nrm > 0.0) {
                const double alpha = x0 >= 0.0 ? -nrm : nrm;
                v0 = x0 - alpha;
                tau = -v0 / alpha;
            }
            double v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = tid + u * WAVE;
                v[u] = (nrm > 0.0) ? (i == k ? 1.0 : (i > k ? hf[k][u] / v0 : 0.0)) : 0.0;
                if (i < RR) sh.V[k][i] = v[u];
            }
#pragma unroll
            for (int c2 = k + 1; c2 < 3; ++c2) {
                double w = 0.0;
#pragma unroll
                for (int u = 0; u < PER; ++u) w += v[u] * hf[c2][u];
                w = wave_sum(w) * tau;
#pragma unroll
                for (int u = 0; u < PER; ++u) hf[c2][u] -= w * v[u];
            }
            if (tid == 0) sh.tau[k] = tau;
        }
    }
    __syncthreads();

}

template <int CMAX, bool STEREO, bool LEAN = false>
__device__ __forceinline__ void load_gidx(const FrameView& fv, int b, int C, FeatShared<CMAX, STEREO, LEAN>& sh)
{
    if constexpr (LEAN) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) sh.gidx[c] = fv.clone_idx[(size_t)b * fv.cmax + c];
    } else {
        for (int c = threadIdx.x; c < 6 * C; c += blockDim.x)
            sh.gidx[c] = fv.clone_idx[(size_t)b * fv.cmax + c / 6] + c % 6;
    }
}

