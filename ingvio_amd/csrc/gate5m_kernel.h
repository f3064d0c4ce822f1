// gate5m_kernel.h — K3 + K5 for MONO windows of up to 11 clones on the machinery of gate5_kernel.h (round 6): FOUR features of one
// filter per wave, the front on 16 lanes per feature, every slot pair's block of P loaded once for the four features, four interleaved
// blocked LDL^T eliminations with the register finish.
//
// A mono observation has two rows, G_o = Pi~_o R_o^T is 2 x 3 and N_o = G_o^T G_o is singular: the Woodbury / difference-coordinate
// reduction of the stereo gate (gate_kernel.h) does not exist.  The gate value in the measurement space (RemoveLostUpdate.cpp:169-273,
// Update.cpp:36-56,104-124) is
//     gamma = r^T V (V^T S V)^-1 V^T r,     S = Gblk Su Gblk^T + s^2 I   (2 nobs square),   Su = D Pcc D^T,   V = left null space of Hf
// and for any S > 0 and Hf of full column rank
//     r^T V (V^T S V)^-1 V^T r = a - b^T C^-1 b,     a = r^T S^-1 r,   b = Hf^T S^-1 r,   C = Hf^T S^-1 Hf
// which is minus the last corner of the LDL^T of the quasi-definite bordered matrix
//     [ S     Hf   r ]      pivots: the 2 nobs rows of S (positive), then the three columns of Hf - by then they hold -C (negative
//     [ Hf^T  0    0 ]      definite) -, and the corner ends as -a + b^T C^-1 b = -gamma.  No pivoting is needed for a quasi-definite
//     [ r^T   0    0 ]      matrix; a feature whose Hf has no rank 3 (no parallax) is as undefined here as in the reference's SVD.
// 22 + 3 + 1 = 26 rows for an 11-clone window: TWO 16-row tile rows (the first-generation gate, gate3_body, pads K to 36 and adds four
// border rows: three tile rows, and its 4 x 4 border block is finished by one lane).  S is padded with unit pivots up to row 27, the Hf
// columns sit at rows 28..30, r at row 31 = BR: the panels eliminate the first tile column, the register finish of gate5 the second -
// 15 pivots, three of them negative, and the border.
//
// Su in 3 x 3 blocks as in gate3_body: with F_c = [cn_c X | -pl_c I] on clone c's six columns (X = [p_f]x; cn_c = 0 for the anchor's own
// observation, whose two rotation terms cancel, RemoveLostUpdate.cpp:476-482; pl_c = 0 only for it in the Selected-timestamp variants,
// quirk Q10) and the anchor's rotation block entering every other observation with -X,
//     Su[c][c'] = T_cc' - cn_c' R_c - cn_c R_c'^T + cn_c cn_c' Q,
//     T_cc' = F_c P(c, c') F_c'^T  (the slot pair's block of P),   R_c = F_c P(c, th_a) X^T  (per observation),   Q = X P(th_a, th_a) X^T.
// The front lanes (one per window slot and feature) build G_c, r_c, R_c, Q and the DIAGONAL 2 x 2 block of S; the 55 pair lanes
// (slots c > c', enumerated over window slots so that a lane's block serves the four features) the off-diagonal ones.
#pragma once
#include "gate5_kernel.h"

template <int CMAX>
struct Gate5mShared {
    static constexpr int NF = 4;
    static constexpr int NPMAX = 2 * CMAX, NTL = 2, HR = 16 * NTL - 4, KPK = NPMAX * (NPMAX + 1) / 2;      // HR: first border row (Hf columns HR..HR+2, r at HR+3)
    static_assert(NPMAX <= HR, "S and its border do not fit two tile rows");
    struct Feat {
        double G[CMAX][6];           // G_c (2 x 3, row-major) by window slot
        double Rb[CMAX][9];          // R_c by window slot
        double Sd[CMAX][3];          // diagonal block of S by window slot: (0,0), (1,0), (1,1)
        double Q[9];                 // X P(th_a, th_a) X^T
        double hb[4][NPMAX + 2];     // border rows by packed row index 2 rank(c) + t: the three columns of Hf, then r
        double thr;                  // the feature's chi^2 threshold (-inf: no valid degrees of freedom), fetched by the front
    } f[NF];
    static constexpr int KPS = (KPK + 16 + 1) & ~1;
    union alignas(16) {
        double kp[2][KPS];           // S of one feature, packed lower triangle by rows; two buffers as in gate5
        double pan[NF][16 * NTL][4]; // panel exchange of the four eliminations
        double fin[NF][16][18];      // the last 16 x 16 blocks, one row per lane of the feature's group
    };
    alignas(16) double lf[NF][20];   // per panel and feature: W = L^-1 of the 4 x 4 diagonal block | r0 r1 r2 r3
};

// F_c P(c, c2) F_c2^T with F_c = [X1 | -pl I], F_c2 = [X2 | -pl2 I], X1 = skew(x1, y1, z1), X2 = skew(x2, y2, z2): gate5_fpf with a
// skew vector per side (cn p_f: zero for the anchor's own observation)
__device__ __forceinline__ void gate5m_fpf(const double Att[9], const double Atp[9], const double Apt[9], const double App[9], double pl, double pl2,
                                           double x1, double y1, double z1, double x2, double y2, double z2, double out[9])
{
    double Ut[9], Up[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Ut[0 + c] = fma(y1, Att[6 + c], fma(-z1, Att[3 + c], -pl * Apt[0 + c]));
        Ut[3 + c] = fma(-x1, Att[6 + c], fma(z1, Att[0 + c], -pl * Apt[3 + c]));
        Ut[6 + c] = fma(x1, Att[3 + c], fma(-y1, Att[0 + c], -pl * Apt[6 + c]));
        Up[0 + c] = fma(y1, Atp[6 + c], fma(-z1, Atp[3 + c], -pl * App[0 + c]));
        Up[3 + c] = fma(-x1, Atp[6 + c], fma(z1, Atp[0 + c], -pl * App[3 + c]));
        Up[6 + c] = fma(x1, Atp[3 + c], fma(-y1, Atp[0 + c], -pl * App[6 + c]));
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        out[3 * r + 0] = fma(y2, Ut[3 * r + 2], fma(-z2, Ut[3 * r + 1], -pl2 * Up[3 * r + 0]));
        out[3 * r + 1] = fma(-x2, Ut[3 * r + 2], fma(z2, Ut[3 * r + 0], -pl2 * Up[3 * r + 1]));
        out[3 * r + 2] = fma(x2, Ut[3 * r + 1], fma(-y2, Ut[3 * r + 0], -pl2 * Up[3 * r + 2]));
    }
}

template <int CMAX>
__device__ __forceinline__ void gate5m_body(CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
                                            int* __restrict__ accept_out)
{
    using SH = Gate5mShared<CMAX>;
    constexpr int NF = SH::NF, NTL = SH::NTL, NLT = NTL * (NTL + 1) / 2, NPMAX = SH::NPMAX, HR = SH::HR;
    static_assert(CMAX <= 16 && CMAX * (CMAX - 1) / 2 <= WAVE, "window class");
    __shared__ SH sh;
    const int ngrp = (fmax_used + NF - 1) / NF;
    const int wg = blockIdx.x, xcd = wg & 7, t = wg >> 3;
    const int bl = xcd + 8 * (t / ngrp), j0 = NF * (t % ngrp);
    if (bl >= nb) return;
    const int b = b0 + bl, lane = threadIdx.x & (WAVE - 1);
    const int F = fv.n_feat[b];
    if (j0 >= F) return;
    const int C = fv.n_clones[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    // ================= front: lane = (feature g, window slot sl) =================
    const int g = lane >> 4, sl = lane & 15, gbase = lane & 48;
    const bool jok = j0 + g < F;
    const size_t oidx = (size_t)b * fv.fmax + (jok ? j0 + g : j0);
    const int a = fv.anchor[oidx];
    const double* pf = fv.pf + oidx * 3;
    const double px = pf[0], py = pf[1], pz = pf[2];
    const unsigned long long mask = jok ? fv.obs_mask[oidx] : 0ULL;
    const int cidx = sl < C ? fv.clone_idx[(size_t)b * fv.cmax + sl] : 0;
    // ================= pair lane = window-slot pair (c, c2), 0 <= c2 < c < C: its block of P, loaded once - requested here, before the
    // front's projections, for every slot pair of the window (see gate5_body) =================
    const int npair = C * (C - 1) / 2;
    const bool pact = lane < npair;
    int pi = 0, pi2 = 0;
    {
        const int q = pact ? lane : 0;
        pi = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
        while ((pi + 1) * (pi + 2) / 2 <= q) ++pi;
        while (pi * (pi + 1) / 2 > q) --pi;
        pi2 = q - pi * (pi + 1) / 2;
    }
    const int pc = pi + 1, pc2 = pi2;                                       // window slots of the pair, pc > pc2
    const int gc = __shfl(cidx, pc, WAVE), gc2 = __shfl(cidx, pc2, WAVE);   // lanes 0..15 hold clone_idx of slots 0..15 (feature 0's group)
    double Att[9], Atp[9], Apt[9], App[9];
    if (pact) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                Att[3 * m + q] = P[(gc + m) + (size_t)(gc2 + q) * ld];
                Atp[3 * m + q] = P[(gc + m) + (size_t)(gc2 + 3 + q) * ld];
                Apt[3 * m + q] = P[(gc + 3 + m) + (size_t)(gc2 + q) * ld];
                App[3 * m + q] = P[(gc + 3 + m) + (size_t)(gc2 + 3 + q) * ld];
            }
    }
    bool valid = false;
    double Gm[2][3], rs[2];
    if (sl < C && ((mask >> sl) & 1ULL)) {
        const double* R = fv.clone_R + ((size_t)b * fv.cmax + sl) * 9;
        const double* pp = fv.clone_p + ((size_t)b * fv.cmax + sl) * 3;
        const double* z = fv.uv + (oidx * fv.cmax + sl) * 4;
        const double zz[4] = { z[0], z[1], 0.0, 0.0 };
        valid = feat_obs<false>(R, pp, zz, px, py, pz, op, Gm, rs);           // RemoveLostUpdate.cpp:435-506; false: NaN guard (:486)
    }
    const unsigned long long vm = __ballot(valid);
    const unsigned gm = (unsigned)(vm >> gbase) & 0xFFFFu;                    // the valid slots of this lane's feature
    const int od = __popc(gm & ((1u << sl) - 1u));                            // rank of this slot among the feature's observations
    const int ga = __shfl(cidx, gbase + (a >= 0 && a < 16 ? a : 0), WAVE);    // first state column of the anchor clone
    if (sl == 0) {                                                            // the gate's threshold, staged now (see gate5_body)
        const int dof = fv.dof[oidx];
        sh.f[g].thr = (dof >= 1 && dof < op.chi2_len) ? op.chi2[dof] : -__builtin_inf();      // Update.cpp:120
    }
    if (valid) {
        typename SH::Feat& fg = sh.f[g];
        const double cn = sl != a ? 1.0 : 0.0, pl = !(op.selected_variant && sl == a) ? 1.0 : 0.0;
        const double cx = cn * px, cy = cn * py, cz = cn * pz;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int m = 0; m < 3; ++m) { fg.G[sl][3 * q + m] = Gm[q][m]; fg.hb[m][2 * od + q] = Gm[q][m]; }
        fg.hb[3][2 * od] = rs[0]; fg.hb[3][2 * od + 1] = rs[1];
        // R_c = F_c P(c, th_a) X^T = cn X U X^T + pl V X,  U = P(th_c, th_a), V = P(p_c, th_a)      (X^T = -X)
        double U[9], V[9], Paa[9], T1[9], T2[9], Rb[9], Q[9];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                U[3 * m + q] = P[(cidx + m) + (size_t)(ga + q) * ld];
                V[3 * m + q] = P[(cidx + 3 + m) + (size_t)(ga + q) * ld];
                Paa[3 * m + q] = P[(ga + m) + (size_t)(ga + q) * ld];
            }
        mulXt(U, px, py, pz, T1);
        mulX(T1, px, py, pz, T2);                 // -X U X = X U X^T ... (X^T M)(X) with X^T = -X twice: the sign pair cancels
        mulX(V, px, py, pz, T1);                  // V X
#pragma unroll
        for (int i = 0; i < 9; ++i) { Rb[i] = cn * T2[i] + pl * T1[i]; fg.Rb[sl][i] = Rb[i]; }
        mulXt(Paa, px, py, pz, T1);
        mulX(T1, px, py, pz, Q);
        if (od == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) fg.Q[i] = Q[i];
        }
        // diagonal block: Su[c][c] = T_cc - cn (R_c + R_c^T) + cn Q,  S_cc = G_c Su[c][c] G_c^T + s^2 I
        double Su[9];
        {
            double Att[9], Atp[9], Apt[9], App[9];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    Att[3 * m + q] = P[(cidx + m) + (size_t)(cidx + q) * ld];
                    Atp[3 * m + q] = P[(cidx + m) + (size_t)(cidx + 3 + q) * ld];
                    Apt[3 * m + q] = P[(cidx + 3 + m) + (size_t)(cidx + q) * ld];
                    App[3 * m + q] = P[(cidx + 3 + m) + (size_t)(cidx + 3 + q) * ld];
                }
            gate5m_fpf(Att, Atp, Apt, App, pl, pl, cx, cy, cz, cx, cy, cz, Su);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int k = 0; k < 3; ++k) Su[3 * m + k] += cn * (Q[3 * m + k] - Rb[3 * m + k] - Rb[3 * k + m]);
        double GS[2][3];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 3; ++k) GS[q][k] = Gm[q][0] * Su[k] + Gm[q][1] * Su[3 + k] + Gm[q][2] * Su[6 + k];
        fg.Sd[sl][0] = GS[0][0] * Gm[0][0] + GS[0][1] * Gm[0][1] + GS[0][2] * Gm[0][2] + op.var;
        fg.Sd[sl][1] = GS[1][0] * Gm[0][0] + GS[1][1] * Gm[0][1] + GS[1][2] * Gm[0][2];
        fg.Sd[sl][2] = GS[1][0] * Gm[1][0] + GS[1][1] * Gm[1][1] + GS[1][2] * Gm[1][2] + op.var;
    }
    // per-feature scalars (wave-uniform)
    unsigned vmg[NF];
    int np_g[NF];
    bool fok_g[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        vmg[q] = (unsigned)(vm >> (16 * q)) & 0xFFFFu;
        const int nobs = __popc(vmg[q]);
        fok_g[q] = 2 * nobs - 3 > 0;
        np_g[q] = 2 * nobs;
    }
    wave_sync();
    // ================= pair lanes: which of the blocks requested before the front are needed at all =================
    bool need = false;
#pragma unroll
    for (int q = 0; q < NF; ++q) need |= ((vmg[q] >> pc) & 1u) && ((vmg[q] >> pc2) & 1u);
    need = need && pact;
    const int kq = lane >> 4, l15 = lane & 15;
    double qx[NF], qy[NF], qz[NF];
    int aq[NF];
#pragma unroll
    for (int fq = 0; fq < NF; ++fq) {
        const size_t oq = (size_t)b * fv.fmax + (j0 + fq < F ? j0 + fq : j0);
        qx[fq] = fv.pf[oq * 3]; qy[fq] = fv.pf[oq * 3 + 1]; qz[fq] = fv.pf[oq * 3 + 2];
        aq[fq] = fv.anchor[oq];
    }
    // S blocks of feature fq into triangle fq & 1: the off-diagonal ones by the pair lanes, the diagonal ones (from the front) by lane = slot
    auto pair_blocks = [&](int fq) {
        const unsigned vq = vmg[fq];
        double* kp = sh.kp[fq & 1];
        const typename SH::Feat& fg = sh.f[fq];
        if (need && ((vq >> pc) & 1u) && ((vq >> pc2) & 1u)) {
            const double cn = pc != aq[fq] ? 1.0 : 0.0, cn2 = pc2 != aq[fq] ? 1.0 : 0.0;
            const double pl = !(op.selected_variant && pc == aq[fq]) ? 1.0 : 0.0, pl2 = !(op.selected_variant && pc2 == aq[fq]) ? 1.0 : 0.0;
            double Su[9];
            gate5m_fpf(Att, Atp, Apt, App, pl, pl2, cn * qx[fq], cn * qy[fq], cn * qz[fq], cn2 * qx[fq], cn2 * qy[fq], cn2 * qz[fq], Su);
            const double cc = cn * cn2;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int k = 0; k < 3; ++k) Su[3 * m + k] += cc * fg.Q[3 * m + k] - cn2 * fg.Rb[pc][3 * m + k] - cn * fg.Rb[pc2][3 * k + m];
            double G1[6], G2[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { G1[i] = fg.G[pc][i]; G2[i] = fg.G[pc2][i]; }
            const int i = __popc(vq & ((1u << pc) - 1u)), i2 = __popc(vq & ((1u << pc2) - 1u));      // ranks, i > i2
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double gs[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) gs[k] = G1[3 * q] * Su[k] + G1[3 * q + 1] * Su[3 + k] + G1[3 * q + 2] * Su[6 + k];
                const int tri = (2 * i + q) * (2 * i + q + 1) / 2 + 2 * i2;
                kp[tri] = gs[0] * G2[0] + gs[1] * G2[1] + gs[2] * G2[2];
                kp[tri + 1] = gs[0] * G2[3] + gs[1] * G2[4] + gs[2] * G2[5];
            }
        }
        if (lane < C && ((vq >> lane) & 1u)) {                               // diagonal block of slot = lane
            const int i = __popc(vq & ((1u << lane) - 1u));
            const int tri = (2 * i) * (2 * i + 1) / 2 + 2 * i;
            kp[tri] = fg.Sd[lane][0];
            kp[tri + 2 * i + 1] = fg.Sd[lane][1];
            kp[tri + 2 * i + 2] = fg.Sd[lane][2];
        }
    };
    double4_f T[NF][NLT];
    // lane constants of the tile fill (see gate5_body): packed-triangle index of every element this lane holds, clamped into the buffer
    int fidx[NLT][4];
    bool fdiag[NLT][4];
#pragma unroll
    for (int ti = 0; ti < NTL; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r, jc = 16 * tj + l15;
                const int hi = i > jc ? i : jc, lo = i > jc ? jc : i;
                const int e = hi * (hi + 1) / 2 + lo;
                fidx[ti * (ti + 1) / 2 + tj][r] = e < SH::KPK + 16 ? e : SH::KPK + 15;
                fdiag[ti * (ti + 1) / 2 + tj][r] = i == jc;
            }
    pair_blocks(0);
    wave_sync();
#pragma unroll
    for (int fq = 0; fq < NF; ++fq) {
        if (fq + 1 < NF) pair_blocks(fq + 1);
        const int np = np_g[fq];
        const double* kp = sh.kp[fq & 1];
        const typename SH::Feat& fg = sh.f[fq];
        bool jreal[NTL];
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) jreal[tj] = 16 * tj + l15 < np;
        // all reads of the feature's tiles first (unconditional, clamped indices), one PIN4 per tile in front of the selects (gate5_body)
#pragma unroll
        for (int t2 = 0; t2 < NLT; ++t2)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[fq][t2][r] = kp[fidx[t2][r]];
        double hrow[NTL], hcol[3];            // border ROW elements of rows HR + kq (r == 3 of the last tile row); border COLUMN elements of rows 16 + kq + 4 r
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) { const int col = 16 * tj + l15; hrow[tj] = fg.hb[kq][col < NPMAX ? col : 0]; }
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int row = 16 * (NTL - 1) + kq + 4 * r, col = 16 * (NTL - 1) + l15; hcol[r] = fg.hb[col >= HR ? col - HR : 0][row < NPMAX ? row : 0]; }
#pragma unroll
        for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int t2 = ti * (ti + 1) / 2 + tj;
                double b0 = T[fq][t2][0], b1 = T[fq][t2][1], b2 = T[fq][t2][2], b3 = T[fq][t2][3];
                PIN4(b0, b1, b2, b3);
                const double bvs[4] = { b0, b1, b2, b3 };
                const int col = 16 * tj + l15;
                const bool bcol = col >= HR;                                          // border columns (Hf, r)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ti + kq + 4 * r;
                    const bool ireal = row < np, brow = row >= HR;                    // border rows: zero pivots, no padding
                    const double padv = (fdiag[t2][r] && !ireal && !brow) ? 1.0 : 0.0; // unit pivots on the padding rows np..HR-1
                    double v = (ireal && jreal[tj]) ? bvs[r] : padv;
                    if (ti == NTL - 1) {
                        // the border, filled symmetrically (the diagonal tile keeps both triangles: the register finish reads whole rows)
                        if (r == 3) v = jreal[tj] ? hrow[tj] : 0.0;                   // rows HR + kq: column col of [Hf | r]^T
                        else if (tj == NTL - 1) v = bcol ? (ireal ? hcol[r] : 0.0) : v;   // columns HR..HR+3 of the rows above the border
                    }
                    T[fq][t2][r] = v;
                }
            }
        }
        wave_sync();                          // triangle fq & 1 is free again; triangle (fq + 1) & 1 is complete
    }
    // ================= four blocked LDL^T eliminations, interleaved: the panels of the first tile column (gate5_body) =================
    int npmax = 0;
#pragma unroll
    for (int q = 0; q < NF; ++q) npmax = np_g[q] > npmax ? np_g[q] : npmax;
    const int npan = (npmax + 3) >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < npan) {
            const int cb = 4 * k;
            if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
                for (int fq = 0; fq < NF; ++fq)
#pragma unroll
                    for (int ti = 0; ti < NTL; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.pan[fq][16 * ti + kq + 4 * r][l15 - cb] = T[fq][ti * (ti + 1) / 2][r];
            }
            wave_sync();
            {
                double a4[4][4];
#pragma unroll
                for (int ra = 0; ra < 4; ++ra) {
                    const double2* pr = reinterpret_cast<const double2*>(sh.pan[g][4 * k + ra]);
                    const double2 u0 = pr[0], u1 = pr[1];
                    a4[ra][0] = u0.x; a4[ra][1] = u0.y; a4[ra][2] = u1.x; a4[ra][3] = u1.y;
                }
                const double r0 = G5RCP(a4[0][0]);
                const double l10 = a4[1][0] * r0, l20 = a4[2][0] * r0, l30 = a4[3][0] * r0;
                const double r1 = G5RCP(a4[1][1] - l10 * a4[1][0]);
                const double t21 = a4[2][1] - l20 * a4[1][0], t31 = a4[3][1] - l30 * a4[1][0];
                const double l21 = t21 * r1, l31 = t31 * r1;
                const double r2 = G5RCP(a4[2][2] - l20 * a4[2][0] - l21 * t21);
                const double t32 = a4[3][2] - l30 * a4[2][0] - l31 * t21;
                const double l32 = t32 * r2;
                const double r3 = G5RCP(a4[3][3] - l30 * a4[3][0] - l31 * t31 - l32 * t32);
                const double w20 = fma(l21, l10, -l20), w31 = fma(l32, l21, -l31);
                const double w30 = fma(-l32, w20, fma(l31, l10, -l30));
                if (sl < 4) {                               // lane sl of the group writes row sl of W = L^-1 (and r_sl)
                    const double c0 = sl == 0 ? 1.0 : (sl == 1 ? -l10 : (sl == 2 ? w20 : w30));
                    const double c1 = sl == 0 ? 0.0 : (sl == 1 ? 1.0 : (sl == 2 ? -l21 : w31));
                    const double c2 = sl <= 1 ? 0.0 : (sl == 2 ? 1.0 : -l32);
                    const double c3 = sl == 3 ? 1.0 : 0.0;
                    double2* o = reinterpret_cast<double2*>(sh.lf[g] + 4 * sl);
                    o[0] = make_double2(c0, c1); o[1] = make_double2(c2, c3);
                    sh.lf[g][16 + sl] = sl == 0 ? r0 : (sl == 1 ? r1 : (sl == 2 ? r2 : r3));
                }
            }
            wave_sync();
#pragma unroll
            for (int fq = 0; fq < NF; ++fq) {
                const double2* lp = reinterpret_cast<const double2*>(sh.lf[fq] + 4 * kq);      // row kq of W = L^-1
                const double2 w01 = lp[0], w23 = lp[1];
                const double dsel = sh.lf[fq][16 + kq];                                         // this lane's pivot reciprocal
                double A[NTL], B[NTL];
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) {
                    const double2* pr = reinterpret_cast<const double2*>(sh.pan[fq][16 * tt + l15]);
                    const double2 u0 = pr[0], u1 = pr[1];
                    double xs = fma(w23.y, u1.y, fma(w23.x, u1.x, fma(w01.y, u0.y, w01.x * u0.x)));
                    if (16 * tt + l15 <= 4 * k + 3) xs = 0.0;                 // pivot rows and everything above: finished
                    A[tt] = xs;
                    B[tt] = -xs * dsel;
                }
#pragma unroll
                for (int ti = 0; ti < NTL; ++ti)
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj)
                        T[fq][ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti], B[tj], T[fq][ti * (ti + 1) / 2 + tj], 0, 0, 0);
            }
            wave_sync();
        }
    }
    // ================= the 16 x 16 block (1, 1) in registers: every lane of group g takes ONE ROW of feature g's block (gate5_body's
    // finish).  Always runs: the three Hf pivots live here whatever the feature's size. =================
    {
#pragma unroll
        for (int fq = 0; fq < NF; ++fq)
#pragma unroll
            for (int r = 0; r < 4; ++r) sh.fin[fq][kq + 4 * r][l15] = T[fq][NLT - 1][r];      // C/D layout: column l15, rows kq + 4 r
        wave_sync();
        double row[16];
        {
            const double2* pr = reinterpret_cast<const double2*>(sh.fin[g][sl]);
#pragma unroll
            for (int q = 0; q < 8; ++q) { const double2 u = pr[q]; row[2 * q] = u.x; row[2 * q + 1] = u.y; }
        }
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const double rinv = G5RCP(row_bcast_f64(row[i], i));      // pivot (i, i): lane i's diagonal element (negative for the Hf columns)
            const double l = row[i] * rinv;
#pragma unroll
            for (int q = i + 1; q < 16; ++q) row[q] = fma(-l, row_bcast_f64(row[q], i), row[q]);
        }
#pragma unroll
        for (int fq = 0; fq < NF; ++fq) T[fq][NLT - 1][3] = __shfl(row[15], 16 * fq + 15, WAVE);
    }
    if (lane == WAVE - 1) {                       // lane (kq = 3, l15 = 15) holds element (BR, BR) = -gamma of every feature
#pragma unroll
        for (int fq = 0; fq < NF; ++fq) {
            if (j0 + fq < F) {
                const size_t oq = (size_t)b * fv.fmax + j0 + fq;
                if (!fok_g[fq]) { gamma_out[oq] = __builtin_nan(""); accept_out[oq] = 0; }
                else {
                    const double gval = -T[fq][NLT - 1][3];
                    const bool ok = gval < sh.f[fq].thr;                                        // Update.cpp:120 (threshold staged by the front)
                    gamma_out[oq] = gval;
                    accept_out[oq] = ok ? 1 : 0;
                }
            }
        }
    }
}
