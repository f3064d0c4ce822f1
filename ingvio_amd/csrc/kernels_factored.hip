// kernels_factored.hip — the structure-exploiting MSCKF path (same posterior as the dense
// K3..K11 path of kernels_msckf.hip / kernels_ekf.hip to FP64 rounding, ~7x fewer FLOPs).
//
// Per feature the stacked Jacobian factors as  Hx = Gblk * D  (RemoveLostUpdate.cpp:474-501):
//   Gblk = blockdiag(G_o), G_o = Pi~_o R_o^T  (rows-per-obs x 3; stack(G_o) is also Hf)
//   D    = [3*nobs x 6C] sparse: row block o has [p_f]x at theta_o, -I at p_o, -[p_f]x at theta_anchor
// so, with V the left-nullspace basis of Hf (H_j = V^T Hx, r_j = V^T r):
//   K5   S_j = V^T (Gblk Su Gblk^T + s^2 I) V,  Su = D Pcc D^T  (3nobs x 3nobs), and by the
//        projector identity  V (V^T S V)^-1 V^T = S^-1 - S^-1 Hf (Hf^T S^-1 Hf)^-1 Hf^T S^-1 :
//        gamma = r^T S^-1 r - b^T (Hf^T S^-1 Hf)^-1 b,  b = Hf^T S^-1 r   (one bordered elimination)
//   K7   H_j^T H_j = D^T W D,  W = Gblk^T (I - U U^T) Gblk,  U = orthonormal basis of range(Hf);
//        the stacked-QR factor R only enters the update through A = R^T R = sum_j H_j^T H_j and
//        z-term b = sum_j H_j^T r_j, accumulated here as 6x6 slot-pair blocks in registers.
//   K8-K11  K H = Pc (A Pcc + s^2 I)^-1 A  (push-through identity; A may be singular, rank n-6):
//        P <- P - (Pc M) Pc^T,  dx = Pc (A Pcc + s^2 I)^-1 b,  Pc = P[:, clone cols].
// gfx950 only.
#include "feat_build.h"
#include "launch_factored.h"

__device__ __forceinline__ void cross3(double ax, double ay, double az, const double v[3], double out[3])
{
    out[0] = ay * v[2] - az * v[1];
    out[1] = az * v[0] - ax * v[2];
    out[2] = ax * v[1] - ay * v[0];
}
// (M X)[r][q] and (X^T M)[q][c] for X = skew(p), M row-major 3x3
__device__ __forceinline__ void mulX(const double M[9], double x, double y, double z, double out[9])
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        out[3 * r + 0] = M[3 * r + 1] * z - M[3 * r + 2] * y;
        out[3 * r + 1] = -M[3 * r + 0] * z + M[3 * r + 2] * x;
        out[3 * r + 2] = M[3 * r + 0] * y - M[3 * r + 1] * x;
    }
}
__device__ __forceinline__ void mulXt(const double M[9], double x, double y, double z, double out[9])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        out[0 + c] = z * M[3 + c] - y * M[6 + c];
        out[3 + c] = -z * M[0 + c] + x * M[6 + c];
        out[6 + c] = y * M[0 + c] - x * M[3 + c];
    }
}

// ---------------------------------------------------------------------------------------------
// K3 + K5, one workgroup per (feature, filter).
// ---------------------------------------------------------------------------------------------
#define GATE2_NT 128

template <int CMAX, bool STEREO>
struct Gate2Shared {
    using Cfg = FeatCfg<CMAX, STEREO>;
    static constexpr int NU = 3 * CMAX;
    static constexpr int NB = Cfg::RR + 4;                 // S bordered by [r | Hf]
    static constexpr int NBB = (NB + 3) / 4;               // 4x4 register blocks per side
    static constexpr int NBLK = NBB * (NBB + 1) / 2;
    static constexpr int BPT = (NBLK + GATE2_NT - 1) / GATE2_NT;
    FeatShared<CMAX, STEREO> f;
    int cna[CMAX];                                 // obs slot != anchor
    int pfl[CMAX];                                 // obs keeps its -I block (false only under Q10)
    union {
        double PD[Cfg::NCOLMAX][NU];               // Pcc D^T
        double T2[NU][Cfg::RR + 4];                // Su Gblk^T
    } u;
    double Su[NU][NU + 1];
    double col[2][4 * NBB];                        // published pivot column, double buffered
    double W4[16];
};

template <int CMAX, bool STEREO>
__global__ __launch_bounds__(GATE2_NT) void k_feat_gate2(
    CovView cv, FrameView fv, MsckfOpts op, int b0, double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    using SH = Gate2Shared<CMAX, STEREO>;
    constexpr int RPO = Cfg::RPO, NT = GATE2_NT, BPT = SH::BPT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SH& sh = *reinterpret_cast<SH*>(smem_raw);
    const int b = b0 + blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    if (j >= fv.n_feat[b]) return;
    const int C = fv.n_clones[b], ncol = 6 * C, ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    const size_t oidx = (size_t)b * fv.fmax + j;
    const int a = fv.anchor[oidx];
    const double* pf = fv.pf + oidx * 3;
    const double px = pf[0], py = pf[1], pz = pf[2];
    load_gidx<CMAX, STEREO>(fv, b, C, sh.f);
    const int rows = feat_phase1<CMAX, STEREO>(fv, op, b, j, C, sh.f);
    const int nobs = sh.f.nobs, nu = 3 * nobs, rho = rows - 3;
    if (rho <= 0) {
        if (tid == 0) { gamma_out[oidx] = __builtin_nan(""); accept_out[oidx] = 0; }
        return;
    }
    if (tid < nobs) {
        const int so = sh.f.slot[tid];
        sh.cna[tid] = so != a;
        sh.pfl[tid] = !(op.selected_variant && so == a);
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;               // 32 x 4 item grid for the small products
    // PD[r][3o..3o+2] = Pcc[r,:] D_o^T  (rows of P read through the symmetric counterpart: coalesced)
    const int ga0 = sh.f.gidx[6 * a];
    for (int o = 0; o < nobs; ++o) {
        const int gc = sh.f.gidx[6 * sh.f.slot[o]];
        const bool cn = sh.cna[o], pl = sh.pfl[o];
        for (int r = tid; r < ncol; r += NT) {
            const int gr = sh.f.gidx[r];
            double dth[3] = { 0.0, 0.0, 0.0 }, pp[3] = { 0.0, 0.0, 0.0 }, cr[3];
            if (cn) {
#pragma unroll
                for (int q = 0; q < 3; ++q) dth[q] = P[gr + (size_t)(gc + q) * ld] - P[gr + (size_t)(ga0 + q) * ld];
            }
            if (pl) {
#pragma unroll
                for (int m = 0; m < 3; ++m) pp[m] = P[gr + (size_t)(gc + 3 + m) * ld];
            }
            cross3(px, py, pz, dth, cr);
#pragma unroll
            for (int m = 0; m < 3; ++m) sh.u.PD[r][3 * o + m] = cr[m] - pp[m];
        }
    }
    __syncthreads();
    // Su = D (Pcc D^T)
    for (int o = ty; o < nobs; o += 4) {
        const int rc = 6 * sh.f.slot[o], ra = 6 * a;
        const bool cn = sh.cna[o], pl = sh.pfl[o];
        for (int jj = tx; jj < nu; jj += 32) {
            double dth[3] = { 0.0, 0.0, 0.0 }, pp[3] = { 0.0, 0.0, 0.0 }, cr[3];
            if (cn) {
#pragma unroll
                for (int q = 0; q < 3; ++q) dth[q] = sh.u.PD[rc + q][jj] - sh.u.PD[ra + q][jj];
            }
            if (pl) {
#pragma unroll
                for (int m = 0; m < 3; ++m) pp[m] = sh.u.PD[rc + 3 + m][jj];
            }
            cross3(px, py, pz, dth, cr);
#pragma unroll
            for (int m = 0; m < 3; ++m) sh.Su[3 * o + m][jj] = cr[m] - pp[m];
        }
    }
    __syncthreads();
    // T2 = Su Gblk^T (aliases PD, which is dead now)
    for (int o2 = ty; o2 < nobs; o2 += 4) {
        for (int u = tx; u < nu; u += 32) {
            const double s0 = sh.Su[u][3 * o2], s1 = sh.Su[u][3 * o2 + 1], s2 = sh.Su[u][3 * o2 + 2];
#pragma unroll
            for (int t = 0; t < RPO; ++t)
                sh.u.T2[u][RPO * o2 + t] = s0 * sh.f.G[o2][t][0] + s1 * sh.f.G[o2][t][1] + s2 * sh.f.G[o2][t][2];
        }
    }
    __syncthreads();
    // S = Gblk T2 + s^2 I, bordered by [r | Hf], held as 4x4 register blocks of the lower triangle;
    // right-looking elimination of the first `rows` pivots leaves -Y^T S^-1 Y in the border block.
    const int nb = rows + 4, nbb = (nb + 3) >> 2, nblk = nbb * (nbb + 1) / 2;
    double val[BPT][4][4];
    int bi_[BPT], bk_[BPT];
#pragma unroll
    for (int s = 0; s < BPT; ++s) {
        const int bq = tid + s * NT;
        int bi = -1, bk = -1;
        if (bq < nblk) {
            bi = (int)((sqrtf(8.0f * bq + 1.0f) - 1.0f) * 0.5f);
            while ((bi + 1) * (bi + 2) / 2 <= bq) ++bi;
            while (bi * (bi + 1) / 2 > bq) --bi;
            bk = bq - bi * (bi + 1) / 2;
        }
        bi_[s] = bi; bk_[s] = bk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * bi + r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = 4 * bk + c;
                double v = 0.0;
                if (bi >= 0 && i < nb && k <= i) {
                    if (i < rows) {
                        const int o = i / RPO, t = i % RPO;
                        v = sh.f.G[o][t][0] * sh.u.T2[3 * o][k] + sh.f.G[o][t][1] * sh.u.T2[3 * o + 1][k] +
                            sh.f.G[o][t][2] * sh.u.T2[3 * o + 2][k] + (i == k ? op.var : 0.0);
                    } else if (k < rows) {
                        const int kb = i - rows;
                        v = (kb == 0) ? sh.f.res[k / RPO][k % RPO] : sh.f.G[k / RPO][k % RPO][kb - 1];
                    }
                }
                val[s][r][c] = v;
            }
        }
    }
    for (int jj = 0; jj < rows; ++jj) {
        const int buf = jj & 1, bj = jj >> 2, rj = jj & 3;
#pragma unroll
        for (int s = 0; s < BPT; ++s) {
            if (bk_[s] == bj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = rj == 0 ? val[s][r][0] : (rj == 1 ? val[s][r][1] : (rj == 2 ? val[s][r][2] : val[s][r][3]));
                    sh.col[buf][4 * bi_[s] + r] = v;
                }
            }
        }
        __syncthreads();
        const double inv = 1.0 / sh.col[buf][jj];
#pragma unroll
        for (int s = 0; s < BPT; ++s) {
            if (bk_[s] >= bj) {                   // block still has columns > jj (or is the pivot block)
                double ci[4], ck[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { ci[r] = sh.col[buf][4 * bi_[s] + r] * inv; ck[r] = sh.col[buf][4 * bk_[s] + r]; }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) val[s][r][c] -= ci[r] * ck[c];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < BPT; ++s) {
        if (bi_[s] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = 4 * bi_[s] + r, k = 4 * bk_[s] + c;
                if (i >= rows && i < nb && k >= rows && k <= i) sh.W4[(i - rows) * 4 + (k - rows)] = val[s][r][c];
            }
    }
    __syncthreads();
    if (tid == 0) {
        // -border = Y^T S^-1 Y, Y = [r | Hf];  gamma = W00 - f^T Wff^-1 f
        double W[4][4];
        for (int p = 0; p < 4; ++p) for (int q = 0; q <= p; ++q) { W[p][q] = -sh.W4[p * 4 + q]; W[q][p] = W[p][q]; }
        const double l00 = sqrt(W[1][1]);
        const double l10 = W[2][1] / l00, l20 = W[3][1] / l00;
        const double l11 = sqrt(W[2][2] - l10 * l10);
        const double l21 = (W[3][2] - l20 * l10) / l11;
        const double l22 = sqrt(W[3][3] - l20 * l20 - l21 * l21);
        const double y0 = W[1][0] / l00;
        const double y1 = (W[2][0] - l10 * y0) / l11;
        const double y2 = (W[3][0] - l20 * y0 - l21 * y1) / l22;
        const double g = W[0][0] - (y0 * y0 + y1 * y1 + y2 * y2);
        const int dof = fv.dof[oidx];
        const bool ok = dof >= 1 && dof < op.chi2_len && g < op.chi2[dof];      // Update.cpp:120
        gamma_out[oidx] = g;
        accept_out[oidx] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// K4 + K6/K7 in Gram form.  grid = (G chunks, nb); chunk g accumulates A_g = sum H_j^T H_j and
// b_g = sum H_j^T r_j over its used features j = g, g+G, ...; lane (c, c') owns the 6x6 block of
// window-slot pair (c, c') in registers for the whole chunk (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
template <int CMAX, bool STEREO>
struct GramShared {
    using Cfg = FeatCfg<CMAX, STEREO>;
    static constexpr int NU = 3 * CMAX;
    static constexpr int NT = ((CMAX * CMAX + 63) / 64) * 64;
    FeatShared<CMAX, STEREO> f;
    int cna[CMAX], pfl[CMAX], obs_of_slot[CMAX];
    double U[Cfg::RR][3];
    double rp[Cfg::RR];
    double Z[NU][3];
    double g[NU];
    double W[NU][NU + 1];
    double Wa[3][NU + 1];
    double Waa[9];
    double ga[3];
};

template <int CMAX, bool STEREO>
__global__ __launch_bounds__((GramShared<CMAX, STEREO>::NT)) void k_feat_gram(
    FrameView fv, MsckfOpts op, int b0, const int* __restrict__ accept_in, int* __restrict__ used_out,
    double* __restrict__ Apart, int* __restrict__ chunk_used, int G, int rstride)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    using SH = GramShared<CMAX, STEREO>;
    constexpr int RPO = Cfg::RPO, RR = Cfg::RR, NT = SH::NT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    SH& sh = *reinterpret_cast<SH*>(smem_raw);
    int* sUse = reinterpret_cast<int*>(smem_raw + ((sizeof(SH) + 15) / 16) * 16);
    const int bl = blockIdx.y, b = b0 + bl, g = blockIdx.x, tid = threadIdx.x;
    const int F = fv.n_feat[b], C = fv.n_clones[b], ncol = 6 * C;

    for (int j = tid; j < F; j += NT) {                    // RemoveLostUpdate.cpp:357-359
        int use = accept_in[(size_t)b * fv.fmax + j];
        if (use && op.max_accept > 0) {
            int rank = 0;
            for (int q = 0; q < j; ++q) rank += accept_in[(size_t)b * fv.fmax + q];
            if (rank >= op.max_accept) use = 0;
        }
        sUse[j] = use;
        if (g == 0) used_out[(size_t)b * fv.fmax + j] = use;
    }
    __syncthreads();

    const int c = tid / C, c2 = tid - c * C;
    const bool pair = tid < C * C;
    double acc[36], bacc[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) bacc[i] = 0.0;
    int nused = 0;

    for (int j = g; j < F; j += G) {
        if (!sUse[j]) continue;
        const size_t oidx = (size_t)b * fv.fmax + j;
        const int a = fv.anchor[oidx];
        const double* pf = fv.pf + oidx * 3;
        const double px = pf[0], py = pf[1], pz = pf[2];
        const int rows = feat_phase1<CMAX, STEREO>(fv, op, b, j, C, sh.f);
        const int nobs = sh.f.nobs, nu = 3 * nobs;
        feat_phase2<CMAX, STEREO>(sh.f, rows);
        if (tid < C) sh.obs_of_slot[tid] = -1;
        __syncthreads();
        if (tid < nobs) {
            const int so = sh.f.slot[tid];
            sh.obs_of_slot[so] = tid;
            sh.cna[tid] = so != a;
            sh.pfl[tid] = !(op.selected_variant && so == a);
        }
        // U = Q[:, 0:3] = H1 H2 H3 e_k ;  rp = (I - U U^T) r      (wave 0, registers)
        if (tid < WAVE) {
            constexpr int PER = (RR + WAVE - 1) / WAVE;
            double ucol[3][PER], rr[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = tid + u * WAVE;
                rr[u] = (i < rows) ? sh.f.res[i / RPO][i % RPO] : 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) ucol[k][u] = (i == k) ? 1.0 : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int q = 2; q >= 0; --q) {
                    double w = 0.0;
#pragma unroll
                    for (int u = 0; u < PER; ++u) { const int i = tid + u * WAVE; w += (i < RR ? sh.f.V[q][i] : 0.0) * ucol[k][u]; }
                    w = wave_sum(w) * sh.f.tau[q];
#pragma unroll
                    for (int u = 0; u < PER; ++u) { const int i = tid + u * WAVE; ucol[k][u] -= w * (i < RR ? sh.f.V[q][i] : 0.0); }
                }
            }
            double c3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double w = 0.0;
#pragma unroll
                for (int u = 0; u < PER; ++u) w += ucol[k][u] * rr[u];
                c3[k] = wave_sum(w);
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int i = tid + u * WAVE;
                if (i < RR) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) sh.U[i][k] = (i < rows) ? ucol[k][u] : 0.0;
                    sh.rp[i] = rr[u] - (ucol[0][u] * c3[0] + ucol[1][u] * c3[1] + ucol[2][u] * c3[2]);
                }
            }
        }
        __syncthreads();
        // Z = Gblk^T U, g = Gblk^T rp
        for (int it = tid; it < nu * 4; it += NT) {
            const int u = it >> 2, k = it & 3, o = u / 3, m = u - 3 * o;
            double s = 0.0;
            if (k < 3) {
#pragma unroll
                for (int t = 0; t < RPO; ++t) s += sh.f.G[o][t][m] * sh.U[RPO * o + t][k];
                sh.Z[u][k] = s;
            } else {
#pragma unroll
                for (int t = 0; t < RPO; ++t) s += sh.f.G[o][t][m] * sh.rp[RPO * o + t];
                sh.g[u] = s;
            }
        }
        __syncthreads();
        // W = blockdiag(G_o^T G_o) - Z Z^T
        for (int u = tid >> 5; u < nu; u += NT / 32) {
            for (int u2 = tid & 31; u2 < nu; u2 += 32) {
                double s = -(sh.Z[u][0] * sh.Z[u2][0] + sh.Z[u][1] * sh.Z[u2][1] + sh.Z[u][2] * sh.Z[u2][2]);
                const int o = u / 3, o2 = u2 / 3;
                if (o == o2) {
                    const int m = u - 3 * o, m2 = u2 - 3 * o2;
#pragma unroll
                    for (int t = 0; t < RPO; ++t) s += sh.f.G[o][t][m] * sh.f.G[o][t][m2];
                }
                sh.W[u][u2] = s;
            }
        }
        __syncthreads();
        // anchor sums: Wa = sum_{o != anchor} W[o-block, :],  ga likewise
        for (int u2 = tid; u2 < nu + 1; u2 += NT) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                double s = 0.0;
                if (u2 < nu) { for (int o = 0; o < nobs; ++o) if (sh.cna[o]) s += sh.W[3 * o + m][u2]; sh.Wa[m][u2] = s; }
                else { for (int o = 0; o < nobs; ++o) if (sh.cna[o]) s += sh.g[3 * o + m]; sh.ga[m] = s; }
            }
        }
        __syncthreads();
        if (tid < 9) {
            const int m = tid / 3, m2 = tid - 3 * m;
            double s = 0.0;
            for (int o = 0; o < nobs; ++o) if (sh.cna[o]) s += sh.Wa[m][3 * o + m2];
            sh.Waa[tid] = s;
        }
        __syncthreads();
        // slot-pair blocks
        if (pair) {
            const int o = sh.obs_of_slot[c], o2 = sh.obs_of_slot[c2];
            if (o >= 0 && o2 >= 0) {
                double Wb[9], WX[9], XtW[9], XtWX[9];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int m2 = 0; m2 < 3; ++m2) Wb[3 * m + m2] = sh.W[3 * o + m][3 * o2 + m2];
                mulX(Wb, px, py, pz, WX);
                mulXt(Wb, px, py, pz, XtW);
                mulXt(WX, px, py, pz, XtWX);
                const double stt = (sh.cna[o] && sh.cna[o2]) ? 1.0 : 0.0;
                const double stp = (sh.cna[o] && sh.pfl[o2]) ? -1.0 : 0.0;
                const double spt = (sh.pfl[o] && sh.cna[o2]) ? -1.0 : 0.0;
                const double spp = (sh.pfl[o] && sh.pfl[o2]) ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int q2 = 0; q2 < 3; ++q2) {
                        acc[6 * q + q2] += stt * XtWX[3 * q + q2];
                        acc[6 * q + 3 + q2] += stp * XtW[3 * q + q2];
                        acc[6 * (3 + q) + q2] += spt * WX[3 * q + q2];
                        acc[6 * (3 + q) + 3 + q2] += spp * Wb[3 * q + q2];
                    }
            }
            if (c == a && o2 >= 0) {              // rows theta_anchor: -X^T Wa[:, o2] [cna X, -pfl I]
                double M[9], XtM[9], XtMX[9];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int m2 = 0; m2 < 3; ++m2) M[3 * m + m2] = sh.Wa[m][3 * o2 + m2];
                mulXt(M, px, py, pz, XtM);
                mulX(XtM, px, py, pz, XtMX);
                const double s1 = sh.cna[o2] ? -1.0 : 0.0, s2 = sh.pfl[o2] ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int q2 = 0; q2 < 3; ++q2) { acc[6 * q + q2] += s1 * XtMX[3 * q + q2]; acc[6 * q + 3 + q2] += s2 * XtM[3 * q + q2]; }
            }
            if (c2 == a && o >= 0) {              // cols theta_anchor: [cna X, -pfl I]^T Wa[:, o]^T (-X)
                double M[9], MX[9], XtMX[9];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int m2 = 0; m2 < 3; ++m2) M[3 * m + m2] = sh.Wa[m2][3 * o + m];
                mulX(M, px, py, pz, MX);
                mulXt(MX, px, py, pz, XtMX);
                const double s1 = sh.cna[o] ? -1.0 : 0.0, s2 = sh.pfl[o] ? 1.0 : 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int q2 = 0; q2 < 3; ++q2) { acc[6 * q + q2] += s1 * XtMX[3 * q + q2]; acc[6 * (3 + q) + q2] += s2 * MX[3 * q + q2]; }
            }
            if (c == a && c2 == a) {
                double MX[9], XtMX[9];
                mulX(sh.Waa, px, py, pz, MX);
                mulXt(MX, px, py, pz, XtMX);
#pragma unroll
                for (int q = 0; q < 9; ++q) acc[6 * (q / 3) + q % 3] += XtMX[q];
            }
            if (c == c2) {
                if (o >= 0) {
                    const double g0 = sh.g[3 * o], g1 = sh.g[3 * o + 1], g2 = sh.g[3 * o + 2];
                    if (sh.cna[o]) { bacc[0] += pz * g1 - py * g2; bacc[1] += -pz * g0 + px * g2; bacc[2] += py * g0 - px * g1; }
                    if (sh.pfl[o]) { bacc[3] -= g0; bacc[4] -= g1; bacc[5] -= g2; }
                }
                if (c == a) {
                    const double g0 = sh.ga[0], g1 = sh.ga[1], g2 = sh.ga[2];
                    bacc[0] -= pz * g1 - py * g2; bacc[1] -= -pz * g0 + px * g2; bacc[2] -= py * g0 - px * g1;
                }
            }
        }
        ++nused;
        __syncthreads();
    }
    double* out = Apart + ((size_t)bl * G + g) * rstride;      // [ncol][ncol+1] row-major, b in the last column
    if (pair) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int q2 = 0; q2 < 6; ++q2) out[(size_t)(6 * c + q) * (ncol + 1) + 6 * c2 + q2] = acc[6 * q + q2];
            if (c == c2) out[(size_t)(6 * c + q) * (ncol + 1) + ncol] = bacc[q];
        }
    }
    if (tid == 0) chunk_used[bl * G + g] = nused;
}

// ---------------------------------------------------------------------------------------------
// K8/K9/K11 in information form, one workgroup per filter:
//   [M | t] = (A Pcc + s^2 I)^-1 [A | b]  by Gauss-Jordan with partial pivoting in LDS,
//   T = Pc M (-> Tout), Pc copy (-> Pcout), dx = Pc t.   P <- P - T Pc^T is done by k_downdate.
// ---------------------------------------------------------------------------------------------
#define INFO_NT 512
__global__ __launch_bounds__(INFO_NT) void k_info_update(
    CovView cv, FrameView fv, int b0, const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
    const double* __restrict__ noise_all, double* __restrict__ Tall, double* __restrict__ Pcall, int ystride,
    double* __restrict__ dx_all, int* __restrict__ m_out, int* __restrict__ nc_out, int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* aug = reinterpret_cast<double*>(smem_raw);
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int C = fv.n_clones[b], ncol = 6 * C, n = cv.n[b], ld = cv.ldp;
    const int LA = 2 * ncol + 1;
    int* sCol = reinterpret_cast<int*>(aug + (size_t)ncol * LA);
    __shared__ int sPiv;
    double* dx = dx_all + (size_t)b * ld;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        for (int r = tid; r < n; r += INFO_NT) dx[r] = 0.0;
        if (tid == 0) { m_out[bl] = 0; nc_out[bl] = ncol; }
        return;
    }
    const double* P = cov_ptr(cv, b);
    const double var = noise_all[bl];
    for (int c = tid; c < ncol; c += INFO_NT) sCol[c] = fv.clone_idx[(size_t)b * fv.cmax + c / 6] + c % 6;
    const int tx = tid & 63, ty = tid >> 6;               // 64 x 8 thread grid: no runtime div/mod in the loops
    for (int i = ty; i < ncol; i += INFO_NT / 64) {
        for (int j = tx; j <= ncol; j += 64) {
            const size_t e = (size_t)i * (ncol + 1) + j;
            double s = 0.0;
            for (int g = 0; g < G; ++g)
                if (chunk_used[bl * G + g]) s += Apart[((size_t)bl * G + g) * rstride + e];
            aug[i * LA + ncol + j] = s;
        }
    }
    __syncthreads();
    for (int i = ty; i < ncol; i += INFO_NT / 64) {
        for (int j = tx; j < ncol; j += 64) {              // consecutive lanes: consecutive j -> coalesced P rows
            double s = (i == j) ? var : 0.0;
            const int gj = sCol[j];
            for (int k = 0; k < ncol; ++k) s += aug[i * LA + ncol + k] * P[gj + (size_t)sCol[k] * ld];   // Pcc[k][j] = P[gj, gk]
            aug[i * LA + j] = s;
        }
    }
    __syncthreads();
    for (int k = 0; k < ncol; ++k) {
        if (tid < WAVE) {                                   // partial pivoting
            double best = -1.0; int bi = k;
            for (int i = k + tid; i < ncol; i += WAVE) { const double v = fabs(aug[i * LA + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off, WAVE); const int oi = __shfl_xor(bi, off, WAVE);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) { sPiv = bi; if (!(best > 0.0)) atomicOr(&status[b], 4); }
        }
        __syncthreads();
        const int p = sPiv;
        if (p != k) {
            for (int j = k + tid; j < LA; j += INFO_NT) { const double t = aug[k * LA + j]; aug[k * LA + j] = aug[p * LA + j]; aug[p * LA + j] = t; }
            __syncthreads();
        }
        const double inv = 1.0 / aug[k * LA + k];
        for (int i = ty; i < ncol; i += INFO_NT / 64) {
            if (i == k) continue;
            const double f = aug[i * LA + k] * inv;
            for (int j = k + 1 + tx; j < LA; j += 64) aug[i * LA + j] -= f * aug[k * LA + j];
        }
        __syncthreads();
    }
    for (int i = ty; i < ncol; i += INFO_NT / 64) {
        const double d = 1.0 / aug[i * LA + i];
        for (int j = tx; j <= ncol; j += 64) aug[i * LA + ncol + j] *= d;
    }
    __syncthreads();
    double* T = Tall + (size_t)bl * ystride;
    double* Pc = Pcall + (size_t)bl * ystride;
    const int mp = (ncol + 3) & ~3;
    for (int r = tid; r < n; r += INFO_NT) {
        double d = 0.0;
        for (int jb = 0; jb < ncol; jb += 16) {
            double acc[16];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) acc[jj] = 0.0;
            for (int k = 0; k < ncol; ++k) {
                const double p = P[r + (size_t)sCol[k] * ld];
                if (jb == 0) { Pc[r + (size_t)k * ld] = p; d += p * aug[k * LA + 2 * ncol]; }
                const double* mrow = aug + k * LA + ncol + jb;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) if (jb + jj < ncol) acc[jj] += p * mrow[jj];
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) if (jb + jj < ncol) T[r + (size_t)(jb + jj) * ld] = acc[jj];
        }
        dx[r] = d;
        for (int j = ncol; j < mp; ++j) { T[r + (size_t)j * ld] = 0.0; Pc[r + (size_t)j * ld] = 0.0; }
    }
    if (tid == 0) { m_out[bl] = ncol; nc_out[bl] = ncol; }
}

// ---------------------------------------------------------------------------------------------
template <int CMAX, bool STEREO>
static void launch_ft(const FactoredLaunch& L, hipStream_t st)
{
    if (L.stage == 0) {
        const size_t sm = sizeof(Gate2Shared<CMAX, STEREO>);
        hipFuncSetAttribute((const void*)k_feat_gate2<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL((k_feat_gate2<CMAX, STEREO>), dim3(L.fmax_used, L.nb), dim3(GATE2_NT), sm, st,
                           L.cv, L.fv, L.op, L.b0, L.gamma, L.accept);
    } else {
        const size_t sm = ((sizeof(GramShared<CMAX, STEREO>) + 15) / 16) * 16 + sizeof(int) * (size_t)L.fv.fmax;
        hipFuncSetAttribute((const void*)k_feat_gram<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL((k_feat_gram<CMAX, STEREO>), dim3(L.G, L.nb), dim3(GramShared<CMAX, STEREO>::NT), sm, st,
                           L.fv, L.op, L.b0, L.accept, L.used, L.Apart, L.chunk_used, L.G, L.rstride);
    }
}

int launch_factored(const FactoredLaunch& L, hipStream_t st)
{
    if (L.stage == 2) {
        const int ncm = 6 * L.fv.cmax;
        const size_t sm = sizeof(double) * (size_t)ncm * (2 * ncm + 1) + sizeof(int) * (size_t)ncm + 16;
        hipFuncSetAttribute((const void*)k_info_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL(k_info_update, dim3(L.nb), dim3(INFO_NT), sm, st, L.cv, L.fv, L.b0, L.Apart, L.chunk_used, L.G, L.rstride,
                           L.noise, L.T, L.Pc, L.ystride, L.dx, L.m_out, L.nc_out, L.status);
        return 0;
    }
    const int cm = L.fv.cmax;
    const int cls = cm <= 6 ? 6 : (cm <= 11 ? 11 : (cm <= 16 ? 16 : -1));
    if (cls < 0) return -1;
#define DISPATCH(CM)                                                         \
    if (cls == CM) { if (L.stereo) launch_ft<CM, true>(L, st); else launch_ft<CM, false>(L, st); return 0; }
    DISPATCH(6)
    DISPATCH(11)
    DISPATCH(16)
#undef DISPATCH
    return -1;
}
