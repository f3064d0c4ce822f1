// gate_kernel.h — shared device code of the factored MSCKF path: small 3x3 helpers, the per-feature record layout and
// the body of the gate kernel (K3 + K5).  Included by kernels_factored.hip (window classes 6 / 11 / 16, tuned for
// occupancy) and kernels_bigwin.hip (windows up to 36 clones, one wave per SIMD).  gfx950 only.
#pragma once
#include "feat_build.h"

typedef double double4_f __attribute__((ext_vector_type(4)));

// The per-lane selects of the elimination ("this lane's row of the 4 x 4 block": kq == 0 ? r0 : kq == 1 ? r1 : ...) must stay
// v_cndmask chains.  Left alone, the compiler SINKS each operand into its own exec-masked branch (the last reciprocal is computed
// under `kq == 3` only): the same instructions issue anyway, plus four s_cbranch_exec* per panel, and the basic-block boundaries stop
// the scheduler from interleaving independent chains (seen in the ISA of gate4 / gate5, round 4).  Pinning the operands as
// outputs of an empty volatile asm keeps them in the straight-line block.
#define PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

__device__ __forceinline__ void inv3sym(const double N[9], double out[9])
{
    const double a = N[0], b = N[1], c = N[2], d = N[4], e = N[5], f = N[8];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02, id = fast_rcp(det);
    out[0] = c00 * id; out[1] = c01 * id; out[2] = c02 * id;
    out[3] = out[1]; out[4] = (a * f - c * c) * id; out[5] = (b * c - a * e) * id;
    out[6] = out[2]; out[7] = out[5]; out[8] = (a * d - b * b) * id;
}
__device__ __forceinline__ void mul33(const double A[9], const double B[9], double C[9])
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) C[3 * i + k] = A[3 * i] * B[k] + A[3 * i + 1] * B[3 + k] + A[3 * i + 2] * B[6 + k];
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
// Per-feature record written by the gate kernel and consumed by k_feat_gram:
//   {nobs, anchor slot, p_f(3), window-slot mask, Ns^-1 (9), hs (3), Nsa (9)} then per observation (ascending slot);
//   Ns = sum_o N_o, hs = sum_o h_o, Nsa = sum over the observations whose clone is not the anchor {slot, cna, pfl, N_o = G_o^T G_o (9), h_o = G_o^T r_o (3)}
// ---------------------------------------------------------------------------------------------
#define REC_HDR 27
#define REC_OBS 15          // slot, cna, pfl, N(9), h(3)

__host__ __device__ constexpr int rec_size(int cmax) { return REC_HDR + REC_OBS * cmax; }

// ---------------------------------------------------------------------------------------------
// K3 + K5, one WAVE per (feature, filter): blocked LDL^T in the reduced observation space.
//
// Stereo (G_o is 4x3, full column rank): by Woodbury on S = s^2 I + Gblk Su Gblk^T,
//     S^-1 = s^-2 (I - Gblk N^-1 Gblk^T) + Gblk N^-1 K^-1 N^-1 Gblk^T,   K = Su + s^2 N^-1,  N = blockdiag(G_o^T G_o)
// so  Y^T S^-1 Y = W^T K^-1 W + s^-2 |r_perp|^2 e0 e0^T  with  W = N^-1 Gblk^T [r | Hf] = [u | 1],
//     u_o = N_o^-1 G_o^T r_o,  |r_perp|^2 = sum_o (|r_o|^2 - h_o^T u_o)   (Hf = Gblk 1, so its residual part is 0):
// a (3 nobs)-dimensional SPD system instead of the (4 nobs)-dimensional S, and no G_o products on Su.
// Mono (G_o is 2x3): K = s^2 I + Gblk Su Gblk^T itself (2 nobs), W = [r | Hf].
//
// The bordered matrix [[K, W], [W^T, 0]] (K padded with unit pivots to a multiple of 16 minus 12, then the four rows of W^T) is
// held as 16x16 lower tiles in the MFMA C/D layout and eliminated in panels of 4 pivots on the matrix cores (see the
// "blocked LDL^T" block in gate3_body); the 4x4 border block ends up holding -W^T K^-1 W.
// The 3x3 (2x2) blocks of K are built one observation pair per lane from four 3x3 blocks of P and collected in a packed lower
// triangle in LDS, from which the tiles are filled.
// ---------------------------------------------------------------------------------------------
template <int CMAX, bool STEREO, bool WREC = true>
struct Gate3Shared {
    static constexpr int D = STEREO ? 3 : 2;
    static constexpr int NPAIR = CMAX * (CMAX + 1) / 2;
    static constexpr int KPK = (D * CMAX) * (D * CMAX + 1) / 2;
    FeatShared<CMAX, STEREO, true> f;
    int cna[CMAX];
    int pfl[CMAX];
    double recbuf[WREC ? REC_HDR + REC_OBS * CMAX : 8];      // the feature record, staged: stored once, coalesced, at the very end (WREC)
    double Ninv[CMAX][9];
    double u[CMAX][3];
    double rperp[CMAX];
    static constexpr int NTL = (D * CMAX + 12 + 15) / 16;      // 16x16 tile rows of the bordered matrix (MFMA back end)
    static constexpr int KP = 16 * NTL - 12;                   // K padded with unit pivots to KP, border rows KP..KP+3
    union alignas(16) {
        double kp[KPK + 16];          // K, packed lower triangle by rows: element (i, j <= i) at i (i + 1) / 2 + j; +16: unclamped reads of padding columns
        double nh[CMAX * 12 + 24];    // before they are built: N_o | h_o per observation and the record's 21 sums
        double pan[16 * NTL][4];      // after the tiles are built: panel exchange of the MFMA elimination
        double bz[64];                // finally: the last diagonal tile's leading 8x8 (left-over pivots 0..2 | pad | the 4x4 border block)
    };
    double Rb[CMAX][9];               // per-observation part of Su: R_o = cn X P(th_o,th_a) X^T + pl P(p_o,th_a) X
    double Qb[9];                     // X P(th_a,th_a) X^T
    int oa;                           // observation index of the anchor clone itself, or -1
};

__device__ __forceinline__ void wave_sync()      // LDS hand-over between the lanes of ONE wave
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// FPW = features per workgroup (= waves per workgroup).  The FRONT of the kernel (projection Jacobians, the record,
// the per-observation terms) only has one lane of work per window slot / observation; with FPW = 4 wave 0 runs it for
// four features at once, 16 lanes each (window classes up to 16 clones), instead of four waves issuing the same
// instructions for 11 lanes each.  The BACK (pair blocks, tile fill, blocked LDL^T) is one wave per feature.
// WREC: also write the per-feature record for a gram kernel that reads it (large-window path); the 6 / 11 / 16-clone gram kernel
// recomputes the few per-observation quantities it needs from the frame inputs instead (k_feat_gram2), so the small-window
// gate neither builds nor stores a record.
template <int CMAX, bool STEREO, int FPW, bool WREC = true, int RSTRIDE = REC_HDR + REC_OBS * CMAX>      // RSTRIDE: doubles between the records of two features
__device__ __forceinline__ void gate3_body(
    CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
    int* __restrict__ accept_out, double* __restrict__ rec_out)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    using SH = Gate3Shared<CMAX, STEREO, WREC>;
    constexpr int RPO = Cfg::RPO, D = SH::D, GS = WAVE / FPW;
    static_assert(FPW == 1 || CMAX <= GS, "a lane group must cover the window");
    __shared__ SH sh4[FPW];
    // XCD-aware mapping: consecutive workgroups go round-robin to the 8 XCDs, so give every XCD whole filters
    // (a filter's P blocks then live in one L2 instead of eight)
    const int nblk = (fmax_used + FPW - 1) / FPW;
    const int w = blockIdx.x, xcd = w & 7, t = w >> 3;
    const int bl = xcd + 8 * (t / nblk), jbase = (t % nblk) * FPW;
    if (bl >= nb) return;
    const int b = b0 + bl, wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int F = fv.n_feat[b];
    if (jbase >= F) return;
    const int C = fv.n_clones[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    auto ldblk = [&](int r0, int c0, double M[9]) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) M[3 * m + q] = P[(r0 + m) + (size_t)(c0 + q) * ld];
    };
    auto finish_pair = [&](SH& sh, int o, int o2, double Su[9]) {        // o >= o2: K block (stereo: + s^2 N^-1 on the diagonal)
        const int q = o * (o + 1) / 2 + o2;
        if (STEREO) {
            if (o == o2) {
#pragma unroll
                for (int i = 0; i < 9; ++i) Su[i] += op.var * sh.Ninv[o][i];
            }
            // rows 3 o + a of the packed triangle, columns 3 o2 + b (the diagonal blocks only keep b <= a)
            int tri = (3 * o) * (3 * o + 1) / 2 + 3 * o2;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2)
                    if (o != o2 || b2 <= a) sh.kp[tri + b2] = Su[3 * a + b2];
                tri += 3 * o + a + 1;
            }
        } else {
            double GSm[2][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2)
                    GSm[r][m2] = sh.f.G[o][r][0] * Su[m2] + sh.f.G[o][r][1] * Su[3 + m2] + sh.f.G[o][r][2] * Su[6 + m2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    double v = GSm[r][0] * sh.f.G[o2][r2][0] + GSm[r][1] * sh.f.G[o2][r2][1] + GSm[r][2] * sh.f.G[o2][r2][2];
                    if (o == o2 && r == r2) v += op.var;
                    const int ii = 2 * o + r, jj = 2 * o2 + r2;
                    if (jj <= ii) sh.kp[ii * (ii + 1) / 2 + jj] = v;
                }
        }
    };

    dbg_stamp(5);
    // ================= FRONT: wave 0, lane group f = lane / GS handles feature jbase + f =================
    if (wave == 0) {
        const int f = lane / GS, sl = lane - GS * f;               // feature of this lane group, lane within the group
        const unsigned long long gmask = GS == 64 ? ~0ULL : ((1ULL << GS) - 1ULL);
        SH& sh = sh4[f];
        const int j = jbase + f;
        const bool jok = j < F;
        const size_t oidx = (size_t)b * fv.fmax + (jok ? j : 0);
        const int a = fv.anchor[oidx];
        const double* pf = fv.pf + oidx * 3;
        const double px = pf[0], py = pf[1], pz = pf[2];
        const unsigned long long mask = jok ? fv.obs_mask[oidx] : 0ULL;
        if (sl < C) sh.f.gidx[sl] = fv.clone_idx[(size_t)b * fv.cmax + sl];
        // ---- per window slot: q = R^T(p_f - p), projection Jacobians, residuals (RemoveLostUpdate.cpp:435-506) --------
        bool valid = false;
        double Gm[RPO][3], rs[RPO];
        if (sl < C && ((mask >> sl) & 1ULL)) {
            const double* R = fv.clone_R + ((size_t)b * fv.cmax + sl) * 9;
            const double* pp = fv.clone_p + ((size_t)b * fv.cmax + sl) * 3;
            const double* z = fv.uv + (oidx * fv.cmax + sl) * 4;
            double zz[4] = { z[0], z[1], 0.0, 0.0 };
            if (STEREO) { zz[2] = z[2]; zz[3] = z[3]; }
            valid = feat_obs<STEREO>(R, pp, zz, px, py, pz, op, Gm, rs);
        }
        const unsigned long long vm = (__ballot(valid) >> (GS * f)) & gmask;
        const int nobs = __popcll(vm);
        if (valid) {
            const int od = __popcll(vm & ((1ULL << sl) - 1ULL));
            sh.f.slot[od] = sl;
#pragma unroll
            for (int q = 0; q < RPO; ++q) {
#pragma unroll
                for (int m = 0; m < 3; ++m) sh.f.G[od][q][m] = Gm[q][m];
                sh.f.res[od][q] = rs[q];
            }
        }
        const bool fok = jok && RPO * nobs - 3 > 0;
        double* const rec = sh.recbuf;            // global stores wait in vmcnt with the loads (gfx9): the record is staged
        if (sl == 0) {
            sh.f.nobs = fok ? nobs : 0;
            if (jok && !fok) { gamma_out[oidx] = __builtin_nan(""); accept_out[oidx] = 0; if (WREC) rec_out[oidx * RSTRIDE] = 0.0; }
        }
        wave_sync();
        double* const Nh = sh.nh;
        double* const sums = Nh + CMAX * 12;
        if (fok && sl < nobs) {
            const int so = sh.f.slot[sl];
            const bool cn = so != a, pl = !(op.selected_variant && so == a);
            sh.cna[sl] = cn;
            sh.pfl[sl] = pl;
            double* ro = rec + REC_HDR + REC_OBS * sl;
            if (WREC) { ro[0] = so; ro[1] = cn ? 1.0 : 0.0; ro[2] = pl ? 1.0 : 0.0; }
            double N[9], h[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2) {
                    double sN = 0.0;
#pragma unroll
                    for (int q = 0; q < RPO; ++q) sN += sh.f.G[sl][q][m] * sh.f.G[sl][q][m2];
                    N[3 * m + m2] = sN;
                    if (WREC) ro[3 + 3 * m + m2] = sN;
                }
                double hh = 0.0;
#pragma unroll
                for (int q = 0; q < RPO; ++q) hh += sh.f.G[sl][q][m] * sh.f.res[sl][q];
                h[m] = hh;
                if (WREC) { ro[12 + m] = hh; Nh[sl * 12 + 9 + m] = hh; }
            }
            if (WREC) {
#pragma unroll
                for (int i = 0; i < 9; ++i) Nh[sl * 12 + i] = N[i];
            }
            if (STEREO) {
                double Ni[9];
                inv3sym(N, Ni);
                double rr = 0.0;
#pragma unroll
                for (int q = 0; q < RPO; ++q) rr += sh.f.res[sl][q] * sh.f.res[sl][q];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const double um = Ni[3 * m] * h[0] + Ni[3 * m + 1] * h[1] + Ni[3 * m + 2] * h[2];
                    sh.u[sl][m] = um;
                    rr -= h[m] * um;
#pragma unroll
                    for (int m2 = 0; m2 < 3; ++m2) sh.Ninv[sl][3 * m + m2] = Ni[3 * m + m2];
                }
                sh.rperp[sl] = rr;
            }
        }
        if (fok && sl == 0) {
            rec[2] = px; rec[3] = py; rec[4] = pz;                       // the back end reads p_f from here
            if (WREC) {
                unsigned long long sm = 0ULL;
                for (int o = 0; o < nobs; ++o) sm |= 1ULL << sh.f.slot[o];
                rec[0] = nobs; rec[1] = a; rec[5] = (double)sm;
            }
        }
        wave_sync();
        // sums for the large-window gram kernel
        if (WREC && fok) {
            for (int cmp = sl; cmp < 21; cmp += GS) {
                const int anch = cmp >= 12, comp = anch ? cmp - 12 : cmp;
                double tt[CMAX];
#pragma unroll
                for (int o = 0; o < CMAX; ++o) tt[o] = (o < nobs && (!anch || sh.cna[o])) ? Nh[o * 12 + comp] : 0.0;      // all loads in flight
                double sacc = 0.0;
#pragma unroll
                for (int o = 0; o < CMAX; ++o) sacc += tt[o];
                sums[cmp] = sacc;
            }
        }
        if (WREC) wave_sync();
        if (WREC && fok) {
            for (int cmp = sl; cmp < 21; cmp += GS) {
                double v = sums[cmp];
                if (cmp < 9) { double Nsi[9]; inv3sym(sums, Nsi); v = Nsi[cmp]; }
                rec[6 + cmp] = v;
            }
        }
        if (WREC) wave_sync();
        // ---- Su = D Pcc D^T in 3x3 blocks.  With U_o = P(th_o, th_a), V_o = P(p_o, th_a):
        //        Su[o][o'] = T_oo' - cn' R_o - cn R_o'^T + cn cn' Q
        //        T_oo' = cn cn' X P(th_o,th_o') X^T + cn pl' X^T P(th_o,p_o') + pl cn' P(p_o,th_o') X + pl pl' P(p_o,p_o')
        //        R_o = cn X U_o X^T + pl V_o X   (per observation),   Q = X P(th_a,th_a) X^T   (per feature)
        //      so a pair needs 4 blocks of P instead of 9.  Pairs with the anchor's own observation (cn' = 0) only keep
        //      the p-column terms; they are built here, in the per-observation pass, which leaves nobs-1 choose 2 (+diag)
        //      <= 55 generic pairs for the back end: ONE round of the wave for an 11-clone window instead of two.
        const bool olane = fok && sl < nobs;
        const unsigned long long amask = (__ballot(olane && sh.f.slot[olane ? sl : 0] == a) >> (GS * f)) & gmask;
        const int oa = amask ? __ffsll((long long)amask) - 1 : -1;          // the anchor clone's own observation, if any
        if (sl == 0) sh.oa = oa;
        if (olane) {
            const int ga = sh.f.gidx[a];
            const int o = sl, gc = sh.f.gidx[sh.f.slot[o]];
            const double cn = sh.cna[o] ? 1.0 : 0.0, pl = sh.pfl[o] ? 1.0 : 0.0;
            double U[9], V[9], Paa[9], T1[9], T2[9];
            ldblk(gc, ga, U);
            ldblk(gc + 3, ga, V);
            ldblk(ga, ga, Paa);
            double B1[9], B2[9], Bp[9];
            if (oa >= 0) { ldblk(gc, ga + 3, B1); ldblk(ga, ga + 3, B2); ldblk(gc + 3, ga + 3, Bp); }
            mulXt(U, px, py, pz, T1);
            mulX(T1, px, py, pz, T2);                 // X U X^T
            mulX(V, px, py, pz, T1);                  // V X
#pragma unroll
            for (int i = 0; i < 9; ++i) sh.Rb[o][i] = cn * T2[i] + pl * T1[i];
            mulXt(Paa, px, py, pz, T1);
            mulX(T1, px, py, pz, T2);
            if (sl == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) sh.Qb[i] = T2[i];
            }
            if (oa >= 0) {                            // pair (o, anchor obs): cn' = 0
                const double pl2 = sh.pfl[oa] ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) B1[i] -= B2[i];
                mulXt(B1, px, py, pz, T1);            // X^T (P(th_o,p_a) - P(th_a,p_a))
                double Su[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Su[i] = cn * pl2 * T1[i] + pl * pl2 * Bp[i];
                if (o >= oa) finish_pair(sh, o, oa, Su);
                else {
                    double St[9];
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int q = 0; q < 3; ++q) St[3 * m + q] = Su[3 * q + m];
                    finish_pair(sh, oa, o, St);
                }
            }
        }
    }
    dbg_stamp(6);
    __syncthreads();
    dbg_stamp(7);

    // ================= BACK: wave w finishes feature jbase + w =================
    SH& sh = sh4[wave];
    const int tid = lane;
    const int j = jbase + wave;
    if (j >= F) return;
    const int nobs = sh.f.nobs;
    if (nobs == 0) return;                            // nothing to gate (reported by the front)
    const size_t oidx = (size_t)b * fv.fmax + j;
    double* const rec = sh.recbuf;
    double* const rec_g = rec_out + oidx * RSTRIDE;
    const double px = rec[2], py = rec[3], pz = rec[4];
    const int oa = sh.oa;
    {
        const int nred = oa >= 0 ? nobs - 1 : nobs, npair = nred * (nred + 1) / 2;
        for (int q = tid; q < npair; q += WAVE) {
            int i = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while ((i + 1) * (i + 2) / 2 <= q) ++i;
            while (i * (i + 1) / 2 > q) --i;
            const int i2 = q - i * (i + 1) / 2;
            const int o = i + ((oa >= 0 && i >= oa) ? 1 : 0), o2 = i2 + ((oa >= 0 && i2 >= oa) ? 1 : 0);
            const int gc = sh.f.gidx[sh.f.slot[o]], gc2 = sh.f.gidx[sh.f.slot[o2]];
            const double cn = sh.cna[o] ? 1.0 : 0.0, cn2 = sh.cna[o2] ? 1.0 : 0.0;
            const double pl = sh.pfl[o] ? 1.0 : 0.0, pl2 = sh.pfl[o2] ? 1.0 : 0.0;
            double Att[9], Atp[9], Apt[9], App[9], T1[9], T2[9], Su[9];
            ldblk(gc, gc2, Att);
            ldblk(gc, gc2 + 3, Atp);
            ldblk(gc + 3, gc2, Apt);
            ldblk(gc + 3, gc2 + 3, App);
            mulXt(Att, px, py, pz, T1);
            mulX(T1, px, py, pz, T2);             // X Ptt' X^T
#pragma unroll
            for (int k = 0; k < 9; ++k) Su[k] = cn * cn2 * (T2[k] + sh.Qb[k]) + pl * pl2 * App[k];
            mulXt(Atp, px, py, pz, T1);
            mulX(Apt, px, py, pz, T2);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    Su[3 * m + k] += cn * pl2 * T1[3 * m + k] + pl * cn2 * T2[3 * m + k] - cn2 * sh.Rb[o][3 * m + k] - cn * sh.Rb[o2][3 * k + m];
            finish_pair(sh, o, o2, Su);
        }
    }
    wave_sync();
    dbg_stamp(8);
    {
        // ---- blocked LDL^T on the matrix cores --------------------------------------------------------
        // The bordered matrix (K padded with unit pivots to KP rows, then the 4 rows of W^T) is held as 16x16
        // lower tiles in the MFMA C/D layout: lane (kq = lane>>4, l15 = lane&15), register r of tile (ti,tj)
        // holds element (16 ti + kq + 4r, 16 tj + l15).  A panel of 4 pivots: its columns go through LDS once
        // ([row][4] layout), every lane reads the 4x4 diagonal block (uniform) and the 4 panel entries of "its"
        // rows, forms X = R L^-T for them and feeds X (A operand) and -X D^-1 (B operand) to one
        // v_mfma_f64_16x16x4 per trailing tile: no per-element broadcasts at all.
        constexpr int NTL = SH::NTL, KP = SH::KP, NLT = NTL * (NTL + 1) / 2;
        // np mod 4 left-over pivots that share the LAST tile row with the border (np = 33 for 11 stereo observations) do not get a
        // panel of their own (one LDS round trip, a diagonal 4x4 factorisation and an MFMA for one real pivot): lane 0 eliminates
        // them in the scalar tail below, together with the border block.
        const int np = D * nobs;
        const bool tail = (np & 3) != 0 && 4 * (np >> 2) >= 16 * (SH::NTL - 1);
        const int npan = tail ? np >> 2 : (np + 3) >> 2, rem = tail ? np & 3 : 0;
        const int kq = tid >> 4, l15 = tid & 15;
        const double mk0 = kq == 0 ? 1.0 : 0.0, mk1 = kq == 1 ? 1.0 : 0.0, mk2 = kq == 2 ? 1.0 : 0.0, mk3 = kq == 3 ? 1.0 : 0.0;      // row selectors of the 4 x 4 block (see PIN4)
        // Tile fill, branch-free: one (clamped) LDS read + selects per element.  Row classes are compile-time:
        // i = 16 ti + kq + 4 r is a border row exactly for (ti, r) = (NTL-1, 1) (then i - KP = kq), rows past the
        // border (ti = NTL-1, r >= 2) are zero; whether a K row/column is real (< np) or a unit pad is a lane select.
        double4_f T[NLT];
        int jo[NTL], jc[NTL];
        bool jreal[NTL];
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) {
            const int j = 16 * tj + l15;
            jreal[tj] = j < np;
            const int jj = jreal[tj] ? j : 0;
            jo[tj] = jj / D; jc[tj] = jj - D * jo[tj];
        }
#pragma unroll
        for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r;
                if (ti == NTL - 1 && r >= 2) {
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) T[ti * (ti + 1) / 2 + tj][r] = 0.0;
                } else if (ti == NTL - 1 && r == 1) {                 // border row kq: W[j][kq]
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) {
                        double v;
                        if (STEREO) {
                            const double uv = sh.u[jo[tj]][jc[tj]];
                            v = kq == 0 ? uv : (kq - 1 == jc[tj] ? 1.0 : 0.0);
                        } else {
                            const double rv = sh.f.res[jo[tj]][jc[tj]], gv = sh.f.G[jo[tj]][jc[tj]][(kq + 2) % 3];
                            v = kq == 0 ? rv : gv;
                        }
                        T[ti * (ti + 1) / 2 + tj][r] = jreal[tj] ? v : 0.0;
                    }
                } else {
                    const bool ireal = i < np;
                    const int ii = ireal ? i : 0, tri_i = ii * (ii + 1) / 2;
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) {
                        // strictly-below-diagonal tiles: j < i, one read at (row base + lane) + an immediate; diagonal tiles pick
                        // the stored half.  Padding rows/columns read something valid and are overridden by the selects.
                        double bv;
                        if (tj < ti) bv = sh.kp[tri_i + 16 * tj + l15];
                        else {
                            const int jj = jreal[tj] ? 16 * tj + l15 : 0;
                            bv = sh.kp[ii >= jj ? tri_i + jj : jj * (jj + 1) / 2 + ii];
                        }
                        const double idv = (i == 16 * tj + l15) ? 1.0 : 0.0;      // unit pivots on the padding rows
                        T[ti * (ti + 1) / 2 + tj][r] = (ireal && jreal[tj]) ? bv : ((!ireal && !jreal[tj]) ? idv : 0.0);
                    }
                }
            }
        }
        wave_sync();                              // kp is dead from here on: its LDS becomes the panel buffer
#pragma unroll
        for (int k = 0; k < KP / 4; ++k) {
            if (k < npan) {
                const int tj0 = k >> 2, cb = 4 * (k & 3);
                if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
                    for (int ti = tj0; ti < NTL; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.pan[16 * ti + kq + 4 * r][l15 - cb] = T[ti * (ti + 1) / 2 + tj0][r];
                }
                wave_sync();
                double a[4][4];
#pragma unroll
                for (int ra = 0; ra < 4; ++ra) {
                    const double2* pr = reinterpret_cast<const double2*>(sh.pan[4 * k + ra]);
                    const double2 u0 = pr[0], u1 = pr[1];
                    a[ra][0] = u0.x; a[ra][1] = u0.y; a[ra][2] = u1.x; a[ra][3] = u1.y;
                }
                double m[NTL][4];
#pragma unroll
                for (int t = tj0; t < NTL; ++t) {
                    const double2* pr = reinterpret_cast<const double2*>(sh.pan[16 * t + l15]);
                    const double2 u0 = pr[0], u1 = pr[1];
                    m[t][0] = u0.x; m[t][1] = u0.y; m[t][2] = u1.x; m[t][3] = u1.y;
                }
                // 4x4 LDL^T of the diagonal block (every lane, uniform data)
                const double r0 = fast_rcp(a[0][0]);
                const double l10 = a[1][0] * r0, l20 = a[2][0] * r0, l30 = a[3][0] * r0;
                const double r1 = fast_rcp(a[1][1] - l10 * a[1][0]);
                const double t21 = a[2][1] - l20 * a[1][0], t31 = a[3][1] - l30 * a[1][0];
                const double l21 = t21 * r1, l31 = t31 * r1;
                const double r2 = fast_rcp(a[2][2] - l20 * a[2][0] - l21 * t21);
                const double t32 = a[3][2] - l30 * a[2][0] - l31 * t21;
                const double l32 = t32 * r2;
                const double r3 = fast_rcp(a[3][3] - l30 * a[3][0] - l31 * t31 - l32 * t32);
                const double dsel = kq == 0 ? r0 : (kq == 1 ? r1 : (kq == 2 ? r2 : r3));
                double A[NTL], B[NTL];
#pragma unroll
                for (int t = tj0; t < NTL; ++t) {
                    const double x0 = m[t][0];
                    const double x1 = m[t][1] - l10 * x0;
                    const double x2 = m[t][2] - l20 * x0 - l21 * x1;
                    const double x3 = m[t][3] - l30 * x0 - l31 * x1 - l32 * x2;
                    double xs = fma(mk3, x3, fma(mk2, x2, fma(mk1, x1, mk0 * x0)));
                    if (16 * t + l15 <= 4 * k + 3) xs = 0.0;              // pivot rows and everything above: finished
                    A[t] = xs;
                    B[t] = -xs * dsel;
                }
#pragma unroll
                for (int ti = tj0; ti < NTL; ++ti)
#pragma unroll
                    for (int tj = tj0; tj <= ti; ++tj)
                        T[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti], B[tj], T[ti * (ti + 1) / 2 + tj], 0, 0, 0);
                wave_sync();
            }
        }
        dbg_stamp(9);
        // last diagonal tile (NTL-1, NTL-1): local rows 0..2 = left-over pivots (r = 0, kq = 0..2; pads are unit rows), local rows 4..7 =
        // the border rows KP..KP+3 (r = 1, kq = 0..3); columns alike
        if (l15 < 8) { sh.bz[kq * 8 + l15] = T[NLT - 1][0]; sh.bz[(4 + kq) * 8 + l15] = T[NLT - 1][1]; }
        wave_sync();
        if (tid == 0) {
            double Bd[4][4];
            for (int p = 0; p < 4; ++p) for (int q = 0; q <= p; ++q) Bd[p][q] = sh.bz[(4 + p) * 8 + 4 + q];
            if (rem > 0) {
                double Lk[3][3], Cx[4][3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int q = 0; q <= i; ++q) Lk[i][q] = sh.bz[i * 8 + q];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) Cx[p][q] = sh.bz[(4 + p) * 8 + q];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if (p < rem) {
                        const double rp = fast_rcp(Lk[p][p]);
#pragma unroll
                        for (int i = p + 1; i < 3; ++i) {
                            const double l = Lk[i][p] * rp;
#pragma unroll
                            for (int q = p + 1; q <= i; ++q) Lk[i][q] -= l * Lk[q][p];
                        }
#pragma unroll
                        for (int bq = 0; bq < 4; ++bq) {
                            const double l = Cx[bq][p] * rp;
#pragma unroll
                            for (int q = p + 1; q < 3; ++q) Cx[bq][q] -= l * Lk[q][p];
#pragma unroll
                            for (int q = 0; q <= bq; ++q) Bd[bq][q] -= l * Cx[q][p];
                        }
                    }
                }
            }
            double W[4][4];
            for (int p = 0; p < 4; ++p) for (int q = 0; q <= p; ++q) W[p][q] = -Bd[p][q];
            const double r1 = fast_rcp(W[1][1]);
            const double l21 = W[2][1] * r1, l31 = W[3][1] * r1;
            const double r2 = fast_rcp(W[2][2] - l21 * W[2][1]);
            const double t32 = W[3][2] - l31 * W[2][1];
            const double l32 = t32 * r2;
            const double r3 = fast_rcp(W[3][3] - l31 * W[3][1] - l32 * t32);
            const double y1 = W[1][0], y2 = W[2][0] - l21 * y1, y3 = W[3][0] - l31 * y1 - l32 * y2;
            double g = W[0][0] - (y1 * y1 * r1 + y2 * y2 * r2 + y3 * y3 * r3);
            if (STEREO) {
                double rp = 0.0;
                for (int o = 0; o < nobs; ++o) rp += sh.rperp[o];
                g += rp / op.var;
            }
            const int dof = fv.dof[oidx];
            const bool ok = dof >= 1 && dof < op.chi2_len && g < op.chi2[dof];      // Update.cpp:120
            gamma_out[oidx] = g;
            accept_out[oidx] = ok ? 1 : 0;
        }
        if (WREC) { for (int e = tid; e < REC_HDR + REC_OBS * nobs; e += WAVE) rec_g[e] = rec[e]; }
        dbg_stamp(10);
        return;
    }
}


// =============================================================================================
// K3 + K5, stereo, second generation (round 3): the chi^2 gate in DIFFERENCE coordinates of the observations.
//
// With W = [u | E], E = 1 (x) I_3 (Hf = Gblk E) the gate value is  gamma = u^T [K^-1 - K^-1 E (E^T K^-1 E)^-1 E^T K^-1] u + |r_perp|^2 / s^2
// (see gate3_body), and for any basis Z of null(E^T) the bracket equals Z (Z^T K Z)^-1 Z^T.  null(E^T) = {x : sum_o x_o = 0}; with
// Z^T x = (x_o - x_b)_{o != b} for a reference observation b (the first one)
//     gamma = w^T Kr^-1 w + |r_perp|^2 / s^2,      w_o = u_o - u_b,      Kr = Z^T K Z   (3 (nobs - 1) square, SPD).
// K = Su + s^2 N^-1 with Su[x][y] = D_x Pcc D_y^T and D_x u = X (th_x - th_a) - pl_x p_x for EVERY observation x (for the anchor's
// own observation the two rotation terms cancel, RemoveLostUpdate.cpp:476-482), so the anchor drops out of the differences:
//     (D_o - D_b) u = X (th_o - th_b) - (pl_o p_o - pl_b p_b)      =: F_o u_o - F_b u_b,      F_c = [X | -pl_c I]  on clone c's 6 columns
//     Kr[o][o'] = T_oo' - R_o - R_o'^T + Q + delta_oo' s^2 N_o^-1,
//     T_oo' = F_o P_oo' F_o'^T   (4 blocks of P per pair),   R_o = F_o P_ob F_b^T   (per observation),   Q = F_b P_bb F_b^T + s^2 N_b^-1.
// The bordered matrix [[Kr, w], [w^T, 0]] has 3 (nobs - 1) + 1 rows: 31 for an 11-clone window = TWO 16-row tile rows with nothing
// wasted (first generation: K (33) + 4 border rows -> three tile rows, 21 of its 37 MFMAs in a last tile row holding 5 real rows),
// 15 MFMAs instead of 37, 8 panels with 2 tile rows instead of 8 with 3 + a scalar tail, no anchor pairs, no G / residuals in LDS.
// The border row is the LAST row of the last tile (BR); K is padded with unit pivots up to it; after the last panel element (BR, BR)
// holds -w^T Kr^-1 w.  Same tile / panel machinery as gate3_body.  pl_c = 0 only for the anchor's own observation in the
// Selected-timestamp variants (quirk Q10), which this form covers as well.
// =============================================================================================
template <int CMAX>
struct Gate4Shared {
    static constexpr int NR = CMAX - 1;                       // observations after the reference one
    static constexpr int NPMAX = 3 * NR;
    static constexpr int NTL = (NPMAX + 1 + 15) / 16;         // tile rows of the bordered matrix
    static constexpr int BR = 16 * NTL - 1;                   // the border row
    static constexpr int KPK = NPMAX * (NPMAX + 1) / 2;
    int gidx[CMAX];                  // state index of the first column of observation o's clone
    int pl[CMAX];
    int nobs;
    double rpsum;                    // sum_o |r_perp,o|^2
    double pf[3];
    double vNinv[CMAX][9];           // s^2 N_o^-1
    double w[3 * CMAX];              // w_o = u_o - u_b at 3 (o - 1)
    double Rb[CMAX][9];              // R_o; Rb[0] = Q
    union alignas(16) {
        double kp[KPK + 16];         // Kr, packed lower triangle by rows (+16: unclamped reads of padding columns)
        double pan[16 * NTL][4];     // panel exchange of the elimination
    };
};

// F_c P(c, c2) F_c2^T from the block's four 3 x 3 parts held in registers, F = [X | -pl I]:
//   U_t = X Ptt - pl Ppt,  U_p = X Ptp - pl Ppp,  out = U_t X^T - pl2 U_p     (81 operations; term by term as gate4_pairblock: 99)
__device__ __forceinline__ void gate5_fpf(const double Att[9], const double Atp[9], const double Apt[9], const double App[9], double pl, double pl2,
                                          double px, double py, double pz, double out[9])
{
    // (X M)[r][c] with X = skew(p): row 0 = (0, -z, y), row 1 = (z, 0, -x), row 2 = (-y, x, 0)
    double Ut[9], Up[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Ut[0 + c] = fma(py, Att[6 + c], fma(-pz, Att[3 + c], -pl * Apt[0 + c]));
        Ut[3 + c] = fma(-px, Att[6 + c], fma(pz, Att[0 + c], -pl * Apt[3 + c]));
        Ut[6 + c] = fma(px, Att[3 + c], fma(-py, Att[0 + c], -pl * Apt[6 + c]));
        Up[0 + c] = fma(py, Atp[6 + c], fma(-pz, Atp[3 + c], -pl * App[0 + c]));
        Up[3 + c] = fma(-px, Atp[6 + c], fma(pz, Atp[0 + c], -pl * App[3 + c]));
        Up[6 + c] = fma(px, Atp[3 + c], fma(-py, Atp[0 + c], -pl * App[6 + c]));
    }
    // (U X^T)[r][c] = sum_k U[r][k] X[c][k]:  X^T columns: c = 0: (0, -z, y), c = 1: (z, 0, -x), c = 2: (-y, x, 0)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        out[3 * r + 0] = fma(py, Ut[3 * r + 2], fma(-pz, Ut[3 * r + 1], -pl2 * Up[3 * r + 0]));
        out[3 * r + 1] = fma(-px, Ut[3 * r + 2], fma(pz, Ut[3 * r + 0], -pl2 * Up[3 * r + 1]));
        out[3 * r + 2] = fma(px, Ut[3 * r + 1], fma(-py, Ut[3 * r + 0], -pl2 * Up[3 * r + 2]));
    }
}

// F_c P(c, c2) F_c2^T for the clones whose first state columns are g and g2 (X = [p_f]x, X^T = -X)
__device__ __forceinline__ void gate4_pairblock(const double* __restrict__ P, int ld, int g, int g2, double pl, double pl2, double px, double py,
                                                double pz, double out[9])
{
    double Att[9], Atp[9], Apt[9], App[9];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            Att[3 * m + q] = P[(g + m) + (size_t)(g2 + q) * ld];
            Atp[3 * m + q] = P[(g + m) + (size_t)(g2 + 3 + q) * ld];
            Apt[3 * m + q] = P[(g + 3 + m) + (size_t)(g2 + q) * ld];
            App[3 * m + q] = P[(g + 3 + m) + (size_t)(g2 + 3 + q) * ld];
        }
    gate5_fpf(Att, Atp, Apt, App, pl, pl2, px, py, pz, out);
}

template <int CMAX>
__device__ __forceinline__ void gate4_body(CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
                                           int* __restrict__ accept_out)
{
    using SH = Gate4Shared<CMAX>;
    constexpr int NTL = SH::NTL, NLT = NTL * (NTL + 1) / 2, BR = SH::BR, NPMAX = SH::NPMAX;
    static_assert(CMAX <= 64 && NPMAX < 16 * NTL, "window class");
    __shared__ SH sh;
    // XCD-aware mapping: consecutive workgroups go round-robin to the 8 XCDs, so give every XCD whole filters
    const int wg = blockIdx.x, xcd = wg & 7, t = wg >> 3;
    const int bl = xcd + 8 * (t / fmax_used), j = t % fmax_used;
    if (bl >= nb) return;
    const int b = b0 + bl, lane = threadIdx.x & (WAVE - 1);
    const int F = fv.n_feat[b];
    if (j >= F) return;
    const int C = fv.n_clones[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    const size_t oidx = (size_t)b * fv.fmax + j;
    dbg_stamp(5);
    // ================= per window slot (lane = slot): projection, N_o, u_o, the block against the reference observation =================
    const int a = fv.anchor[oidx];
    const double* pf = fv.pf + oidx * 3;
    const double px = pf[0], py = pf[1], pz = pf[2];
    const unsigned long long mask = fv.obs_mask[oidx];
    const int sl = lane;
    const int cidx = sl < C ? fv.clone_idx[(size_t)b * fv.cmax + sl] : 0;
    bool valid = false;
    double Gm[4][3], rs[4];
    if (sl < C && ((mask >> sl) & 1ULL)) {
        const double* R = fv.clone_R + ((size_t)b * fv.cmax + sl) * 9;
        const double* pp = fv.clone_p + ((size_t)b * fv.cmax + sl) * 3;
        const double* z = fv.uv + (oidx * fv.cmax + sl) * 4;
        const double zz[4] = { z[0], z[1], z[2], z[3] };
        valid = feat_obs<true>(R, pp, zz, px, py, pz, op, Gm, rs);          // RemoveLostUpdate.cpp:435-506; false: NaN guard (:486)
    }
    const unsigned long long vm = __ballot(valid);
    const int nobs = __popcll(vm);
    const bool fok = 4 * nobs - 3 > 0;
    if (!fok) {
        if (lane == 0) { gamma_out[oidx] = __builtin_nan(""); accept_out[oidx] = 0; }
        return;
    }
    const int bslot = __ffsll((long long)vm) - 1;                      // the reference observation: the first one (wave-uniform)
    const int od = __popcll(vm & ((1ULL << sl) - 1ULL));
    const double plo = (valid && !(op.selected_variant && sl == a)) ? 1.0 : 0.0;
    const double plb = !(op.selected_variant && bslot == a) ? 1.0 : 0.0;
    const int gb = __builtin_amdgcn_readlane(cidx, bslot);
    double u[3] = { 0.0, 0.0, 0.0 }, Ni[9], rr = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) Ni[i] = 0.0;
    if (valid) {
        double N[9], h[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
#pragma unroll
            for (int m2 = m; m2 < 3; ++m2) {
                double sN = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) sN += Gm[q][m] * Gm[q][m2];
                N[3 * m + m2] = sN; N[3 * m2 + m] = sN;
            }
            double hh = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) hh += Gm[q][m] * rs[q];
            h[m] = hh;
        }
        inv3sym(N, Ni);
#pragma unroll
        for (int q = 0; q < 4; ++q) rr += rs[q] * rs[q];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            u[m] = Ni[3 * m] * h[0] + Ni[3 * m + 1] * h[1] + Ni[3 * m + 2] * h[2];
            rr -= h[m] * u[m];
        }
    }
    const double rsum = wave_sum(rr);
    double ub[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(u[m]), bslot), hi = __builtin_amdgcn_readlane(__double2hiint(u[m]), bslot);
        ub[m] = __hiloint2double(hi, lo);
    }
    if (lane == 0) { sh.nobs = nobs; sh.rpsum = rsum; sh.pf[0] = px; sh.pf[1] = py; sh.pf[2] = pz; }
    if (valid) {
        sh.gidx[od] = cidx;
        sh.pl[od] = plo != 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) sh.vNinv[od][i] = op.var * Ni[i];
        if (od > 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m) sh.w[3 * (od - 1) + m] = u[m] - ub[m];
        }
        double Rb[9];
        gate4_pairblock(P, ld, cidx, gb, plo, plb, px, py, pz, Rb);          // R_o = F_o P_ob F_b^T  (o = b: F_b P_bb F_b^T)
#pragma unroll
        for (int i = 0; i < 9; ++i) sh.Rb[od][i] = od == 0 ? Rb[i] + op.var * Ni[i] : Rb[i];      // Rb[0] = Q
    }
    wave_sync();
    dbg_stamp(7);
#if defined(GATE4_STOP_AFTER) && GATE4_STOP_AFTER == 1      // ablation probe (tools/gpu_gate_ablation.sh): time of the front alone
    if (lane == 0) { gamma_out[oidx] = sh.Rb[0][0] + sh.w[0] + sh.vNinv[nobs - 1][8]; accept_out[oidx] = 0; }
    return;
#endif
    // ================= Kr blocks, one observation pair per lane =================
    const int tid = lane;
    const int nred = nobs - 1, np = 3 * nred;
    {
        const int npair = nred * (nred + 1) / 2;
        double Q[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Q[i] = sh.Rb[0][i];
        for (int q = tid; q < npair; q += WAVE) {
            int i = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while ((i + 1) * (i + 2) / 2 <= q) ++i;
            while (i * (i + 1) / 2 > q) --i;
            const int i2 = q - i * (i + 1) / 2;
            const int o = i + 1, o2 = i2 + 1;
            double Su[9];
            gate4_pairblock(P, ld, sh.gidx[o], sh.gidx[o2], sh.pl[o] ? 1.0 : 0.0, sh.pl[o2] ? 1.0 : 0.0, px, py, pz, Su);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int k = 0; k < 3; ++k) Su[3 * m + k] += Q[3 * m + k] - sh.Rb[o][3 * m + k] - sh.Rb[o2][3 * k + m];
            if (i == i2) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Su[k] += sh.vNinv[o][k];
            }
            // rows 3 i + a of the packed triangle, columns 3 i2 + c (the diagonal blocks only keep c <= a)
            int tri = (3 * i) * (3 * i + 1) / 2 + 3 * i2;
#pragma unroll
            for (int a2 = 0; a2 < 3; ++a2) {
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2)
                    if (i != i2 || c2 <= a2) sh.kp[tri + c2] = Su[3 * a2 + c2];
                tri += 3 * i + a2 + 1;
            }
        }
    }
    wave_sync();
    dbg_stamp(8);
#if defined(GATE4_STOP_AFTER) && GATE4_STOP_AFTER == 2      // front + pair blocks
    if (lane == 0) { gamma_out[oidx] = sh.kp[0] + sh.kp[np * (np + 1) / 2 - 1]; accept_out[oidx] = 0; }
    return;
#endif
    // ================= blocked LDL^T on the matrix cores (see gate3_body) =================
    const int kq = tid >> 4, l15 = tid & 15;
    const double mk0 = kq == 0 ? 1.0 : 0.0, mk1 = kq == 1 ? 1.0 : 0.0, mk2 = kq == 2 ? 1.0 : 0.0, mk3 = kq == 3 ? 1.0 : 0.0;      // row selectors of the 4 x 4 block (see PIN4)
    const int npan = (np + 3) >> 2;
    double4_f T[NLT];
    bool jreal[NTL];
#pragma unroll
    for (int tj = 0; tj < NTL; ++tj) jreal[tj] = 16 * tj + l15 < np;
#pragma unroll
    for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + kq + 4 * r;
            const bool ireal = i < np;
            const int ii = ireal ? i : 0, tri_i = ii * (ii + 1) / 2;
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int jcol = 16 * tj + l15;
                // strictly-below-diagonal tiles: j < i, one read at (row base + lane); diagonal tiles pick the stored half.  Padding rows /
                // columns read something valid and are overridden by the selects.
                double bv;
                if (tj < ti) bv = sh.kp[tri_i + jcol];
                else {
                    const int jj = jreal[tj] ? jcol : 0;
                    bv = sh.kp[ii >= jj ? tri_i + jj : jj * (jj + 1) / 2 + ii];
                }
                const double idv = (i == jcol) ? 1.0 : 0.0;                       // unit pivots on the padding rows
                double v = (ireal && jreal[tj]) ? bv : ((!ireal && !jreal[tj]) ? idv : 0.0);
                if (ti == NTL - 1 && r == 3) {                                    // i == BR for kq == 3: the border row w^T, corner 0
                    const double wv = sh.w[jreal[tj] ? jcol : 0];
                    v = kq == 3 ? (jreal[tj] ? wv : 0.0) : v;
                }
                T[ti * (ti + 1) / 2 + tj][r] = v;
            }
        }
    }
    wave_sync();                              // kp is dead from here on: its LDS becomes the panel buffer
#if defined(GATE4_STOP_AFTER) && GATE4_STOP_AFTER == 3      // front + pair blocks + tile fill
    {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NLT; ++q) acc += T[q][0] + T[q][1] + T[q][2] + T[q][3];
        acc = wave_sum(acc);
        if (lane == 0) { gamma_out[oidx] = acc; accept_out[oidx] = 0; }
        return;
    }
#endif
    constexpr int KPAN = (NPMAX + 3) / 4;
#pragma unroll
    for (int k = 0; k < KPAN; ++k) {
        if (k < npan) {
            const int tj0 = k >> 2, cb = 4 * (k & 3);
            if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
                for (int ti = tj0; ti < NTL; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh.pan[16 * ti + kq + 4 * r][l15 - cb] = T[ti * (ti + 1) / 2 + tj0][r];
            }
            wave_sync();
            double a4[4][4];
#pragma unroll
            for (int ra = 0; ra < 4; ++ra) {
                const double2* pr = reinterpret_cast<const double2*>(sh.pan[4 * k + ra]);
                const double2 u0 = pr[0], u1 = pr[1];
                a4[ra][0] = u0.x; a4[ra][1] = u0.y; a4[ra][2] = u1.x; a4[ra][3] = u1.y;
            }
            double m[NTL][4];
#pragma unroll
            for (int tt = tj0; tt < NTL; ++tt) {
                const double2* pr = reinterpret_cast<const double2*>(sh.pan[16 * tt + l15]);
                const double2 u0 = pr[0], u1 = pr[1];
                m[tt][0] = u0.x; m[tt][1] = u0.y; m[tt][2] = u1.x; m[tt][3] = u1.y;
            }
            // 4x4 LDL^T of the diagonal block (every lane, uniform data)
            double r0 = fast_rcp(a4[0][0]);
            const double l10 = a4[1][0] * r0, l20 = a4[2][0] * r0, l30 = a4[3][0] * r0;
            double r1 = fast_rcp(a4[1][1] - l10 * a4[1][0]);
            const double t21 = a4[2][1] - l20 * a4[1][0], t31 = a4[3][1] - l30 * a4[1][0];
            const double l21 = t21 * r1, l31 = t31 * r1;
            double r2 = fast_rcp(a4[2][2] - l20 * a4[2][0] - l21 * t21);
            const double t32 = a4[3][2] - l30 * a4[2][0] - l31 * t21;
            const double l32 = t32 * r2;
            double r3 = fast_rcp(a4[3][3] - l30 * a4[3][0] - l31 * t31 - l32 * t32);
            PIN4(r0, r1, r2, r3);
            // The last panel of the tile grid (k = 4 NTL - 1) ends ON the border row: BR is its fourth row but not a pivot - it stays
            // active (finished-row threshold one lower) and its column takes no part in the update (zero pivot reciprocal).
            const bool bord = (k == 4 * NTL - 1);
            const double dsel = bord ? fma(mk2, r2, fma(mk1, r1, mk0 * r0)) : fma(mk3, r3, fma(mk2, r2, fma(mk1, r1, mk0 * r0)));      // this lane's pivot reciprocal (MASKSEL)
            double A[NTL], B[NTL];
#pragma unroll
            for (int tt = tj0; tt < NTL; ++tt) {
                double x0 = m[tt][0];
                double x1 = m[tt][1] - l10 * x0;
                double x2 = m[tt][2] - l20 * x0 - l21 * x1;
                double x3 = m[tt][3] - l30 * x0 - l31 * x1 - l32 * x2;
                PIN4(x0, x1, x2, x3);
                double xs = fma(mk3, x3, fma(mk2, x2, fma(mk1, x1, mk0 * x0)));
                if (16 * tt + l15 <= 4 * k + (bord ? 2 : 3)) xs = 0.0;   // pivot rows and everything above: finished
                A[tt] = xs;
                B[tt] = -xs * dsel;
            }
#pragma unroll
            for (int ti = tj0; ti < NTL; ++ti)
#pragma unroll
                for (int tj = tj0; tj <= ti; ++tj)
                    T[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti], B[tj], T[ti * (ti + 1) / 2 + tj], 0, 0, 0);
            wave_sync();
        }
    }
    dbg_stamp(9);
    if (tid == WAVE - 1) {                        // lane (kq = 3, l15 = 15) holds element (BR, BR) = -w^T Kr^-1 w
        const double g = -T[NLT - 1][3] + sh.rpsum / op.var;
        const int dof = fv.dof[oidx];
        const bool ok = dof >= 1 && dof < op.chi2_len && g < op.chi2[dof];      // Update.cpp:120
        gamma_out[oidx] = g;
        accept_out[oidx] = ok ? 1 : 0;
    }
    dbg_stamp(10);
}

// =============================================================================================
// The same gate (gate4_body: difference coordinates of the observations) for LARGE windows (17..36 clones), TWO waves per feature.
// The first generation ran one wave per feature with 28 (32-clone class) lower tiles in VGPRs + AGPRs and 48 KB of LDS: three
// waves per CU, every LDS round trip and every MFMA chain fully exposed (config 5: 0.60 ms per 32 filters, 0.19 of the FP64 peak).
// Here the 3 (nobs - 1) + 1 bordered system has one tile row less (6 instead of 7 for 30 clones: 21 tiles), the tile rows are dealt
// to the two waves at compile time (each holds ~half of the tiles, in VGPRs only), the pair blocks are built by 128 lanes, and the
// panel exchange is double-buffered: ONE workgroup barrier per panel.  Both waves transform the panel rows of every live tile row
// (the B operands of the other wave's tiles) - a few VALU instructions - and issue MFMAs for their own tiles only.  Wave 0 runs
// the per-observation front and stages the record k_feat_gram_big reads (same layout as gate3_body's, rec_size(BIG_CMAX) stride).
// =============================================================================================
template <int CMAX>
struct Gate4BigShared {
    static constexpr int NR = CMAX - 1, NPMAX = 3 * NR, NTL = (NPMAX + 1 + 15) / 16, BR = 16 * NTL - 1, KPK = NPMAX * (NPMAX + 1) / 2;
    int gidx[CMAX];
    int pl[CMAX], cna[CMAX], slot[CMAX];
    int nobs;
    double rpsum;
    double pf[3];
    double vNinv[CMAX][9];
    double w[3 * CMAX];
    double Rb[CMAX][9];
    double recbuf[REC_HDR + REC_OBS * CMAX];
    union alignas(16) {
        double kp[KPK + 16];
        double nh[CMAX * 12 + 24];          // front: N_o | h_o per observation and the record's 21 sums
        double pan[2][16 * NTL][4];         // double-buffered panel exchange
    };
};

// tile row ti of the bordered matrix -> owning wave (0 / 1): rows taken from the largest down, each to the lighter wave
template <int NTL>
__host__ __device__ constexpr int gate4_row_owner(int ti)
{
    int load0 = 0, load1 = 0, own = 0;
    for (int r = NTL - 1; r >= 0; --r) {
        const int o = load0 <= load1 ? 0 : 1;
        if (o == 0) load0 += r + 1; else load1 += r + 1;
        if (r == ti) own = o;
    }
    return own;
}

template <int CMAX, int W>
__device__ __forceinline__ void gate4_big_back(Gate4BigShared<CMAX>& sh, const MsckfOpts& op, const FrameView& fv, size_t oidx, int lane,
                                               double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    using SH = Gate4BigShared<CMAX>;
    constexpr int NTL = SH::NTL, NLT = NTL * (NTL + 1) / 2, NPMAX = SH::NPMAX;
    const int kq = lane >> 4, l15 = lane & 15;
    const double mk0 = kq == 0 ? 1.0 : 0.0, mk1 = kq == 1 ? 1.0 : 0.0, mk2 = kq == 2 ? 1.0 : 0.0, mk3 = kq == 3 ? 1.0 : 0.0;      // row selectors of the 4 x 4 block (see PIN4)
    const int np = 3 * (sh.nobs - 1), npan = (np + 3) >> 2;
    double4_f T[NLT];                         // only the tiles of this wave's rows are ever touched (the others fold away)
    bool jreal[NTL];
#pragma unroll
    for (int tj = 0; tj < NTL; ++tj) jreal[tj] = 16 * tj + l15 < np;
#pragma unroll
    for (int ti = 0; ti < NTL; ++ti) {
        if (gate4_row_owner<NTL>(ti) != W) continue;
        // Round 6: every read of the tile row is UNCONDITIONAL, from an index clamped into the buffer (padding elements read something
        // valid and are overridden), ALL of them are issued before the first select (one PIN4 per tile keeps them out of the selects'
        // branches without a wait per element).  As `ireal ? kp[..] : ..` every element was an exec-masked branch with its own LDS
        // round trip behind s_waitcnt lgkmcnt(0): ~100 serial round trips per wave, 10.8 k of the kernel's 73 k cycles (shader-clock
        // stamps, 32 filters x 300 features x 30 clones).
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r, jcol = 16 * tj + l15;
                const int hi = tj < ti ? i : (i > jcol ? i : jcol), lo = tj < ti ? jcol : (i > jcol ? jcol : i);
                const int e = hi * (hi + 1) / 2 + lo;
                T[ti * (ti + 1) / 2 + tj][r] = sh.kp[e < SH::KPK + 16 ? e : SH::KPK + 15];
            }
        double wv[NTL];
        if (ti == NTL - 1) {
#pragma unroll
            for (int tj = 0; tj < NTL; ++tj) wv[tj] = sh.w[jreal[tj] ? 16 * tj + l15 : 0];
        }
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
            double4_f& t4 = T[ti * (ti + 1) / 2 + tj];
            double b0 = t4[0], b1 = t4[1], b2 = t4[2], b3 = t4[3];
            PIN4(b0, b1, b2, b3);
            const double bvs[4] = { b0, b1, b2, b3 };
            const int jcol = 16 * tj + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r;
                const bool ireal = i < np;
                const double idv = (i == jcol) ? 1.0 : 0.0;
                double v = (ireal && jreal[tj]) ? bvs[r] : ((!ireal && !jreal[tj]) ? idv : 0.0);
                if (ti == NTL - 1 && r == 3) v = kq == 3 ? (jreal[tj] ? wv[tj] : 0.0) : v;
                t4[r] = v;
            }
        }
    }
    __syncthreads();                          // kp is dead: its LDS becomes the panel buffers
    dbg_stamp(9);
    constexpr int KPAN = (NPMAX + 3) / 4;
#pragma unroll
    for (int k = 0; k < KPAN; ++k) {
        if (k < npan) {
            const int tj0 = k >> 2, cb = 4 * (k & 3), buf = k & 1;
            if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
                for (int ti = tj0; ti < NTL; ++ti) {
                    if (gate4_row_owner<NTL>(ti) != W) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh.pan[buf][16 * ti + kq + 4 * r][l15 - cb] = T[ti * (ti + 1) / 2 + tj0][r];
                }
            }
            __syncthreads();
            double a4[4][4];
#pragma unroll
            for (int ra = 0; ra < 4; ++ra) {
                const double2* pr = reinterpret_cast<const double2*>(sh.pan[buf][4 * k + ra]);
                const double2 u0 = pr[0], u1 = pr[1];
                a4[ra][0] = u0.x; a4[ra][1] = u0.y; a4[ra][2] = u1.x; a4[ra][3] = u1.y;
            }
            double r0 = fast_rcp(a4[0][0]);
            const double l10 = a4[1][0] * r0, l20 = a4[2][0] * r0, l30 = a4[3][0] * r0;
            double r1 = fast_rcp(a4[1][1] - l10 * a4[1][0]);
            const double t21 = a4[2][1] - l20 * a4[1][0], t31 = a4[3][1] - l30 * a4[1][0];
            const double l21 = t21 * r1, l31 = t31 * r1;
            double r2 = fast_rcp(a4[2][2] - l20 * a4[2][0] - l21 * t21);
            const double t32 = a4[3][2] - l30 * a4[2][0] - l31 * t21;
            const double l32 = t32 * r2;
            double r3 = fast_rcp(a4[3][3] - l30 * a4[3][0] - l31 * t31 - l32 * t32);
            PIN4(r0, r1, r2, r3);
            const bool bord = (k == 4 * NTL - 1);            // the last panel of the grid ends ON the border row (see gate4_body)
            const double dsel = bord ? fma(mk2, r2, fma(mk1, r1, mk0 * r0)) : fma(mk3, r3, fma(mk2, r2, fma(mk1, r1, mk0 * r0)));      // this lane's pivot reciprocal (MASKSEL)
            double A[NTL], B[NTL];
#pragma unroll
            for (int tt = tj0; tt < NTL; ++tt) {
                const double2* pr = reinterpret_cast<const double2*>(sh.pan[buf][16 * tt + l15]);
                const double2 u0 = pr[0], u1 = pr[1];
                const double x0 = u0.x;
                const double x1 = u0.y - l10 * x0;
                const double x2 = u1.x - l20 * x0 - l21 * x1;
                const double x3 = u1.y - l30 * x0 - l31 * x1 - l32 * x2;
                double xs = fma(mk3, x3, fma(mk2, x2, fma(mk1, x1, mk0 * x0)));
                if (16 * tt + l15 <= 4 * k + (bord ? 2 : 3)) xs = 0.0;
                A[tt] = xs;
                B[tt] = -xs * dsel;
            }
#pragma unroll
            for (int ti = tj0; ti < NTL; ++ti) {
                if (gate4_row_owner<NTL>(ti) != W) continue;
#pragma unroll
                for (int tj = tj0; tj <= ti; ++tj)
                    T[ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti], B[tj], T[ti * (ti + 1) / 2 + tj], 0, 0, 0);
            }
        }
    }
    if (gate4_row_owner<NTL>(NTL - 1) == W && lane == WAVE - 1) {      // element (BR, BR) = -w^T Kr^-1 w
        const double g = -T[NLT - 1][3] + sh.rpsum / op.var;
        const int dof = fv.dof[oidx];
        const bool ok = dof >= 1 && dof < op.chi2_len && g < op.chi2[dof];      // Update.cpp:120
        gamma_out[oidx] = g;
        accept_out[oidx] = ok ? 1 : 0;
    }
}

template <int CMAX, int RSTRIDE>
__device__ __forceinline__ void gate4_big_body(CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
                                               int* __restrict__ accept_out, double* __restrict__ rec_out)
{
    using SH = Gate4BigShared<CMAX>;
    static_assert(CMAX <= 64, "one lane per window slot");
    extern __shared__ __attribute__((aligned(16))) char gate4_smem[];
    SH& sh = *reinterpret_cast<SH*>(gate4_smem);
    const int wg = blockIdx.x, xcd = wg & 7, t = wg >> 3;
    const int bl = xcd + 8 * (t / fmax_used), j = t % fmax_used;
    if (bl >= nb) return;
    const int b = b0 + bl, tid = threadIdx.x, wave = tid >> 6, lane = tid & (WAVE - 1);
    const int F = fv.n_feat[b];
    if (j >= F) return;
    const int C = fv.n_clones[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    const size_t oidx = (size_t)b * fv.fmax + j;
    const int a = fv.anchor[oidx];
    const double* pf = fv.pf + oidx * 3;
    const double px = pf[0], py = pf[1], pz = pf[2];
    double* const rec = sh.recbuf;
    dbg_stamp(5);
    // ================= front, lane = window slot, BOTH waves: each evaluates the projections (validity and ordering of the observations
    // must be known to both); wave 0 goes on with N_o, u_o, the gate's per-observation terms and the record, wave 1 with the blocks
    // R_o = F_o P_ob F_b^T against the reference observation (36 loads and two cross-product passes per lane) =================
    {
        const unsigned long long mask = fv.obs_mask[oidx];
        const int sl = lane;
        const int cidx = sl < C ? fv.clone_idx[(size_t)b * fv.cmax + sl] : 0;
        bool valid = false;
        double Gm[4][3], rs[4];
        if (sl < C && ((mask >> sl) & 1ULL)) {
            const double* R = fv.clone_R + ((size_t)b * fv.cmax + sl) * 9;
            const double* pp = fv.clone_p + ((size_t)b * fv.cmax + sl) * 3;
            const double* z = fv.uv + (oidx * fv.cmax + sl) * 4;
            const double zz[4] = { z[0], z[1], z[2], z[3] };
            valid = feat_obs<true>(R, pp, zz, px, py, pz, op, Gm, rs);
        }
        const unsigned long long vm = __ballot(valid);
        const int nobs = __popcll(vm);
        const bool fok = 4 * nobs - 3 > 0;
        if (wave == 0 && lane == 0) {
            sh.nobs = fok ? nobs : 0;
            if (!fok) { gamma_out[oidx] = __builtin_nan(""); accept_out[oidx] = 0; rec_out[oidx * RSTRIDE] = 0.0; }
        }
        if (fok) {
            const int bslot = __ffsll((long long)vm) - 1;
            const int od = __popcll(vm & ((1ULL << sl) - 1ULL));
            const bool cn = sl != a, plf = !(op.selected_variant && sl == a);
            if (wave == 1) {
                const double plo = (valid && plf) ? 1.0 : 0.0;
                const double plb = !(op.selected_variant && bslot == a) ? 1.0 : 0.0;
                const int gb = __builtin_amdgcn_readlane(cidx, bslot);
                if (valid) {
                    double Rb[9];
                    gate4_pairblock(P, ld, cidx, gb, plo, plb, px, py, pz, Rb);
#pragma unroll
                    for (int i = 0; i < 9; ++i) sh.Rb[od][i] = Rb[i];              // Rb[0] = F_b P_bb F_b^T: Q = Rb[0] + s^2 N_b^-1 is formed by the pair lanes
                }
            } else {
                double u[3] = { 0.0, 0.0, 0.0 }, Ni[9], N[9], h[3] = { 0.0, 0.0, 0.0 }, rr = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) { Ni[i] = 0.0; N[i] = 0.0; }
                if (valid) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
#pragma unroll
                        for (int m2 = m; m2 < 3; ++m2) {
                            double sN = 0.0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) sN += Gm[q][m] * Gm[q][m2];
                            N[3 * m + m2] = sN; N[3 * m2 + m] = sN;
                        }
                        double hh = 0.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) hh += Gm[q][m] * rs[q];
                        h[m] = hh;
                    }
                    inv3sym(N, Ni);
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr += rs[q] * rs[q];
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        u[m] = Ni[3 * m] * h[0] + Ni[3 * m + 1] * h[1] + Ni[3 * m + 2] * h[2];
                        rr -= h[m] * u[m];
                    }
                }
                const double rsum = wave_sum(rr);
                double ub[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(u[m]), bslot), hi = __builtin_amdgcn_readlane(__double2hiint(u[m]), bslot);
                    ub[m] = __hiloint2double(hi, lo);
                }
                if (lane == 0) {
                    sh.rpsum = rsum; sh.pf[0] = px; sh.pf[1] = py; sh.pf[2] = pz;
                    rec[0] = nobs; rec[1] = a; rec[2] = px; rec[3] = py; rec[4] = pz; rec[5] = (double)vm;      // the valid observations' slot mask
                }
                double* const Nh = sh.nh;
                if (valid) {
                    sh.gidx[od] = cidx; sh.pl[od] = plf; sh.cna[od] = cn; sh.slot[od] = sl;
#pragma unroll
                    for (int i = 0; i < 9; ++i) sh.vNinv[od][i] = op.var * Ni[i];
                    if (od > 0) {
#pragma unroll
                        for (int m = 0; m < 3; ++m) sh.w[3 * (od - 1) + m] = u[m] - ub[m];
                    }
                    // the record of k_feat_gram_big: per observation {slot, cna, pfl, N_o (9), h_o (3)}
                    double* ro = rec + REC_HDR + REC_OBS * od;
                    ro[0] = sl; ro[1] = cn ? 1.0 : 0.0; ro[2] = plf ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < 9; ++i) { ro[3 + i] = N[i]; Nh[od * 12 + i] = N[i]; }
#pragma unroll
                    for (int i = 0; i < 3; ++i) { ro[12 + i] = h[i]; Nh[od * 12 + 9 + i] = h[i]; }
                }
                wave_sync();
                // the record's sums: Ns = sum N_o (-> Ns^-1), hs = sum h_o, Nsa = sum over the observations whose clone is not the anchor
                double* const sums = Nh + CMAX * 12;
                if (lane < 21) {
                    // Round 6: all CMAX reads are issued together and selected afterwards, added in the same order - as a loop over nobs
                    // with the read under its condition this was a chain of up to 36 dependent LDS round trips on 21 lanes
                    const int anch = lane >= 12, comp = anch ? lane - 12 : lane;
                    double xs[CMAX];
                    int cs[CMAX];
#pragma unroll
                    for (int o = 0; o < CMAX; ++o) { xs[o] = Nh[o * 12 + comp]; cs[o] = sh.cna[o]; }
#pragma unroll
                    for (int o = 0; o + 3 < CMAX; o += 4) PIN4(xs[o], xs[o + 1], xs[o + 2], xs[o + 3]);
                    double sacc = 0.0;
#pragma unroll
                    for (int o = 0; o < CMAX; ++o) sacc += (o < nobs && (!anch || cs[o])) ? xs[o] : 0.0;
                    sums[lane] = sacc;
                }
                wave_sync();
                if (lane < 21) {
                    double v = sums[lane];
                    if (lane < 9) { double Nsi[9]; inv3sym(sums, Nsi); v = Nsi[lane]; }
                    rec[6 + lane] = v;
                }
            }
        }
    }
    dbg_stamp(6);
    __syncthreads();
    dbg_stamp(7);
    const int nobs = sh.nobs;
    if (nobs == 0) return;
    // ================= Kr blocks, one observation pair per lane (128 lanes).  The 36 covariance loads of a lane's NEXT pair are issued
    // before the arithmetic of the current one (three to four rounds per feature at 30 clones: the rounds used to pay one L2
    // round trip each) =================
    {
        const int nred = nobs - 1, npair = nred * (nred + 1) / 2;
        double Q[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Q[i] = sh.Rb[0][i] + sh.vNinv[0][i];
        auto pair_of = [&](int q, int& i, int& i2) {
            i = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while ((i + 1) * (i + 2) / 2 <= q) ++i;
            while (i * (i + 1) / 2 > q) --i;
            i2 = q - i * (i + 1) / 2;
        };
        double Att[9], Atp[9], Apt[9], App[9];
        auto load4 = [&](int g, int g2) {
#ifdef GATE4_ABL_SAMEBLK      // ablation probe (results invalid): every lane reads the same block - what the pair stage costs without its gathers
            g = sh.gidx[1]; g2 = sh.gidx[1];
#endif
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Att[3 * m + c] = P[(g + m) + (size_t)(g2 + c) * ld];
                    Atp[3 * m + c] = P[(g + m) + (size_t)(g2 + 3 + c) * ld];
                    Apt[3 * m + c] = P[(g + 3 + m) + (size_t)(g2 + c) * ld];
                    App[3 * m + c] = P[(g + 3 + m) + (size_t)(g2 + 3 + c) * ld];
                }
        };
        int i = 0, i2 = 0;
        if (tid < npair) { pair_of(tid, i, i2); load4(sh.gidx[i + 1], sh.gidx[i2 + 1]); }
        for (int q = tid; q < npair; q += 2 * WAVE) {
            const int o = i + 1, o2 = i2 + 1, ci = i, ci2 = i2;
            const double pl = sh.pl[o] ? 1.0 : 0.0, pl2 = sh.pl[o2] ? 1.0 : 0.0;
            double T1[9], T2[9], Su[9], Bt[9], Bp[9];
            mulXt(Att, px, py, pz, T1);
            mulX(T1, px, py, pz, T2);                 // X Ptt' X^T
#pragma unroll
            for (int k = 0; k < 9; ++k) { Su[k] = T2[k] + (pl * pl2) * App[k]; Bt[k] = Atp[k]; Bp[k] = Apt[k]; }
            if (q + 2 * WAVE < npair) { pair_of(q + 2 * WAVE, i, i2); load4(sh.gidx[i + 1], sh.gidx[i2 + 1]); }      // next pair's loads in flight
            mulXt(Bt, px, py, pz, T1);
            mulX(Bp, px, py, pz, T2);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    Su[3 * m + k] += pl2 * T1[3 * m + k] + pl * T2[3 * m + k] + Q[3 * m + k] - sh.Rb[o][3 * m + k] - sh.Rb[o2][3 * k + m];
            if (ci == ci2) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Su[k] += sh.vNinv[o][k];
            }
            int tri = (3 * ci) * (3 * ci + 1) / 2 + 3 * ci2;
#pragma unroll
            for (int a2 = 0; a2 < 3; ++a2) {
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2)
                    if (ci != ci2 || c2 <= a2) sh.kp[tri + c2] = Su[3 * a2 + c2];
                tri += 3 * ci + a2 + 1;
            }
        }
    }
    __syncthreads();
    dbg_stamp(8);
    // the record leaves now (its stores complete under the elimination)
    {
        double* const rec_g = rec_out + oidx * RSTRIDE;
        for (int e = tid; e < REC_HDR + REC_OBS * nobs; e += 2 * WAVE) rec_g[e] = rec[e];
    }
    if (wave == 0) gate4_big_back<CMAX, 0>(sh, op, fv, oidx, lane, gamma_out, accept_out);
    else gate4_big_back<CMAX, 1>(sh, op, fv, oidx, lane, gamma_out, accept_out);
    dbg_stamp(10);
}
