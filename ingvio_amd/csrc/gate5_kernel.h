// gate5_kernel.h — K3 + K5 for stereo windows of up to 11 clones, third generation (round 4): FOUR features of one filter per wave.
//
// Same mathematics as gate4_body (gate_kernel.h: the chi^2 gate in difference coordinates of the observations, a 3 (nobs - 1) + 1
// bordered system eliminated as a blocked LDL^T on the matrix cores); what changes is who does the work.  Ablation of gate4 on
// MI355X (512 filters x 150 features, tools/gpu_variant_times.sh with GATE4_STOP_AFTER): front 0.081 ms, pair blocks 0.112 ms,
// tile fill ~0, LDL^T + gate 0.090 ms.  The front issues every instruction for 11 live lanes of 64, and the pair stage is bound by
// the vector-memory path, not by arithmetic: each of its 55 lanes gathers a 6 x 6 block of P with 36 eight-byte loads - and the
// 150 features of a filter gather the SAME blocks (only X = [p_f]x differs between them).  Here
//   * the front runs for four features at once, 16 lanes (window slots) each;
//   * a pair lane loads its block P(c, c') ONCE and forms F P F^T for the four features from registers (pairs are enumerated over
//     window SLOTS 1..C-1, so they name the same clones for every feature; a feature skips the pairs its mask does not hold);
//   * Kr of one feature at a time goes through ONE packed triangle in LDS into tile registers (4 x 3 tiles = 96 VGPRs), then the
//     four eliminations run interleaved - four independent LDS -> VALU -> MFMA chains per wave instead of one.
// LDS 11.9 KB per wave (gate4: 5.8 KB per feature), registers cap the kernel at 2 waves per SIMD = 8 features in flight per
// SIMD (gate4: 4).  Window classes 6 and 11 (C (C - 1) / 2 <= 64 slot pairs: one round); the 16-clone class keeps gate4.
#pragma once
#include "gate_kernel.h"

// reciprocal of a pivot in the eliminations: v_rcp_f64 + TWO Newton steps (1.1e-16 relative, dev_common.h) by default;
// -DGATE5_RCP1: one step (2.2e-15) - a shorter dependent chain per panel (measured: see DESIGN 4.1)
#ifdef GATE5_RCP1
__device__ __forceinline__ double g5_rcp1(double x) { double r = __builtin_amdgcn_rcp(x); return fma(fma(-x, r, 1.0), r, r); }
#define G5RCP(x) g5_rcp1(x)
#else
#define G5RCP(x) fast_rcp(x)
#endif

template <int CMAX>
struct Gate5Shared {
    static constexpr int NF = 4;
    static constexpr int NR = CMAX - 1, NPMAX = 3 * NR, NTL = (NPMAX + 1 + 15) / 16, BR = 16 * NTL - 1, KPK = NPMAX * (NPMAX + 1) / 2;
    struct Feat {
        double vNinv[CMAX][9];       // s^2 N_c^-1 by window slot
        double Rb[CMAX][9];          // R_c = F_c P(c, b) F_b^T by window slot (b = the feature's reference slot; Rb[b] unused)
        double Q[9];                 // F_b P(b, b) F_b^T + s^2 N_b^-1
        double w[3 * CMAX];          // w_o = u_o - u_b at 3 (rank(o) - 1)
        double rpsum;                // sum_o |r_perp,o|^2
        double thr;                  // the feature's chi^2 threshold (-inf: no valid degrees of freedom), fetched by the front
    } f[NF];
    static constexpr int KPS = (KPK + 16 + 1) & ~1;
#ifndef GATE5_DB
#define GATE5_DB 1               // two triangles: feature f + 1's pair blocks are formed while feature f's tiles are filled
#endif
    union alignas(16) {
        double kp[GATE5_DB ? 2 : 1][KPS];           // Kr of one feature, packed lower triangle by rows (+16: unclamped reads of padding columns); two
                                     // buffers: feature f + 1's blocks are formed while feature f's tiles are filled
        double pan[NF][16 * NTL][4]; // panel exchange of the four eliminations
        double fin[NF][16][18];      // the last 16 x 16 blocks, one row per lane of the feature's group (see the finish of the eliminations)
    };
    alignas(16) double lf[NF][20];   // per panel and feature: W = L^-1 of the 4 x 4 diagonal block (row-major, 16) | r0 r1 r2 r3
};

template <int CMAX>
__device__ __forceinline__ void gate5_body(CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
                                           int* __restrict__ accept_out)
{
    using SH = Gate5Shared<CMAX>;
    constexpr int NF = SH::NF, NTL = SH::NTL, NLT = NTL * (NTL + 1) / 2, NPMAX = SH::NPMAX;
    static_assert(CMAX <= 16 && CMAX * (CMAX - 1) / 2 <= WAVE && NPMAX < 16 * NTL, "window class");
    __shared__ SH sh;
    // XCD-aware mapping: consecutive workgroups go round-robin to the 8 XCDs, so give every XCD whole filters
    const int ngrp = (fmax_used + NF - 1) / NF;
    const int wg = blockIdx.x, xcd = wg & 7, t = wg >> 3;
    const int bl = xcd + 8 * (t / ngrp), j0 = NF * (t % ngrp);
    if (bl >= nb) return;
    const int b = b0 + bl, lane = threadIdx.x & (WAVE - 1);
    const int F = fv.n_feat[b];
    if (j0 >= F) return;
    const int C = fv.n_clones[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    dbg_stamp(5);
    // ================= front: lane = (feature g, window slot sl) =================
    const int g = lane >> 4, sl = lane & 15, gbase = lane & 48;
    const bool jok = j0 + g < F;
    const size_t oidx = (size_t)b * fv.fmax + (jok ? j0 + g : j0);
    const int a = fv.anchor[oidx];
    const double* pf = fv.pf + oidx * 3;
    const double px = pf[0], py = pf[1], pz = pf[2];
    const unsigned long long mask = jok ? fv.obs_mask[oidx] : 0ULL;
    const int cidx = sl < C ? fv.clone_idx[(size_t)b * fv.cmax + sl] : 0;
    // ================= pair lane = window-slot pair (c, c2), 1 <= c2 <= c < C: its block of P, loaded once.  Round 6: requested HERE,
    // before the front's projections, for every slot pair of the window (whether any of the four features needs the pair is only known
    // after the front): the block's 36 loads used to start after the front - a full memory round trip in front of the first pair block =====
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
    const int pc = pi + 1, pc2 = pi2 + 1;                                   // window slots of the pair
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
    double Gm[4][3], rs[4];
    if (sl < C && ((mask >> sl) & 1ULL)) {
        const double* R = fv.clone_R + ((size_t)b * fv.cmax + sl) * 9;
        const double* pp = fv.clone_p + ((size_t)b * fv.cmax + sl) * 3;
        const double* z = fv.uv + (oidx * fv.cmax + sl) * 4;
        const double zz[4] = { z[0], z[1], z[2], z[3] };
        valid = feat_obs<true>(R, pp, zz, px, py, pz, op, Gm, rs);          // RemoveLostUpdate.cpp:435-506; false: NaN guard (:486)
    }
    const unsigned long long vm = __ballot(valid);
    const unsigned gm = (unsigned)(vm >> gbase) & 0xFFFFu;                  // the valid slots of this lane's feature
    const int bslot = gm ? __ffs(gm) - 1 : 0;                               // reference observation: the first one
    const int od = __popc(gm & ((1u << sl) - 1u));
    const double plo = (valid && !(op.selected_variant && sl == a)) ? 1.0 : 0.0;
    const double plb = !(op.selected_variant && bslot == a) ? 1.0 : 0.0;
    const int gb = __shfl(cidx, gbase + bslot, WAVE);
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
    // sum over the 16 lanes of the feature, in the order wave_sum takes within a 16-lane group
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) rr += __shfl_xor(rr, off, WAVE);
    double ub[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) ub[m] = __shfl(u[m], gbase + bslot, WAVE);
    if (sl == 0) {
        sh.f[g].rpsum = rr;
        // the gate's threshold, fetched HERE (round 6): read at the very end - dof, then chi2[dof], two dependent global round trips of the
        // one lane that writes the results - it was 3.2 k of a group's 37 k cycles (shader-clock stamps of a contended workgroup)
        const int dof = fv.dof[oidx];
        sh.f[g].thr = (dof >= 1 && dof < op.chi2_len) ? op.chi2[dof] : -__builtin_inf();      // Update.cpp:120
    }
    if (valid) {
        typename SH::Feat& fg = sh.f[g];
#pragma unroll
        for (int i = 0; i < 9; ++i) fg.vNinv[sl][i] = op.var * Ni[i];
        if (od > 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m) fg.w[3 * (od - 1) + m] = u[m] - ub[m];
        }
        double Rb[9];
        gate4_pairblock(P, ld, cidx, gb, plo, plb, px, py, pz, Rb);          // R_c = F_c P(c, b) F_b^T  (c = b: F_b P_bb F_b^T)
        if (od == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) fg.Q[i] = Rb[i] + op.var * Ni[i];
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) fg.Rb[sl][i] = Rb[i];
        }
    }
    // per-feature scalars (wave-uniform)
    unsigned vmg[NF];
    int np_g[NF], bs_g[NF];
    bool fok_g[NF];
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        vmg[q] = (unsigned)(vm >> (16 * q)) & 0xFFFFu;
        const int nobs = __popc(vmg[q]);
        fok_g[q] = nobs > 0;                                               // 4 nobs - 3 > 0
        np_g[q] = nobs > 0 ? 3 * (nobs - 1) : 0;
        bs_g[q] = vmg[q] ? __ffs(vmg[q]) - 1 : 0;
    }
    wave_sync();
    dbg_stamp(7);
#if defined(GATE5_STOP_AFTER) && GATE5_STOP_AFTER == 1      // ablation probe (tools/gpu_variant_times.sh): the front alone
    if (lane < NF && j0 + lane < F) { gamma_out[(size_t)b * fv.fmax + j0 + lane] = sh.f[lane].Q[0] + sh.f[lane].w[0] + sh.f[lane].rpsum; accept_out[(size_t)b * fv.fmax + j0 + lane] = 0; }
    return;
#endif
    // ================= pair lanes: which of the blocks requested above are needed at all =================
    bool need = false;
#pragma unroll
    for (int q = 0; q < NF; ++q) need |= ((vmg[q] >> pc) & 1u) && ((vmg[q] >> pc2) & 1u) && pc2 != bs_g[q];
    need = need && pact;
    const int kq = lane >> 4, l15 = lane & 15;
    const double mk0 = kq == 0 ? 1.0 : 0.0, mk1 = kq == 1 ? 1.0 : 0.0, mk2 = kq == 2 ? 1.0 : 0.0, mk3 = kq == 3 ? 1.0 : 0.0;      // row selectors of the 4 x 4 block (see PIN4)
    // per-feature inputs of the pair stage (wave-uniform: scalar loads, all requested up front)
    double qx[NF], qy[NF], qz[NF];
    int aq[NF];
#pragma unroll
    for (int fq = 0; fq < NF; ++fq) {
        const size_t oq = (size_t)b * fv.fmax + (j0 + fq < F ? j0 + fq : j0);
        qx[fq] = fv.pf[oq * 3]; qy[fq] = fv.pf[oq * 3 + 1]; qz[fq] = fv.pf[oq * 3 + 2];
        aq[fq] = fv.anchor[oq];
    }
    // Kr blocks of feature fq into triangle fq & 1
    auto pair_blocks = [&](int fq) {
        const unsigned vq = vmg[fq];
        if (need && ((vq >> pc) & 1u) && ((vq >> pc2) & 1u) && pc2 != bs_g[fq]) {
            const double pl = !(op.selected_variant && pc == aq[fq]) ? 1.0 : 0.0, pl2 = !(op.selected_variant && pc2 == aq[fq]) ? 1.0 : 0.0;
            double Su[9];
            gate5_fpf(Att, Atp, Apt, App, pl, pl2, qx[fq], qy[fq], qz[fq], Su);
            const typename SH::Feat& fg = sh.f[fq];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int k = 0; k < 3; ++k) Su[3 * m + k] += fg.Q[3 * m + k] - fg.Rb[pc][3 * m + k] - fg.Rb[pc2][3 * k + m];
            if (pc == pc2) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Su[k] += fg.vNinv[pc][k];
            }
            // rank of the slots among the feature's observations after the reference one
            const int i = __popc(vq & ((1u << pc) - 1u)) - 1, i2 = __popc(vq & ((1u << pc2) - 1u)) - 1;
            double* kp = sh.kp[GATE5_DB ? (fq & 1) : 0];
            int tri = (3 * i) * (3 * i + 1) / 2 + 3 * i2;
#pragma unroll
            for (int a2 = 0; a2 < 3; ++a2) {
#pragma unroll
                for (int c2 = 0; c2 < 3; ++c2)
                    if (i != i2 || c2 <= a2) kp[tri + c2] = Su[3 * a2 + c2];
                tri += 3 * i + a2 + 1;
            }
        }
    };
    double4_f T[NF][NLT];
    // lane constants of the tile fill: element (i, j) of the bordered matrix sits at tri(max) + min in the packed triangle; indices
    // of padding elements are clamped into the buffer (what they read is discarded)
    int fidx[NLT][4], wcol[NTL];
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
#pragma unroll
    for (int tj = 0; tj < NTL; ++tj) wcol[tj] = 16 * tj + l15 < 3 * CMAX ? 16 * tj + l15 : 0;
    if (GATE5_DB) { pair_blocks(0); wave_sync(); }
#pragma unroll
    for (int fq = 0; fq < NF; ++fq) {
        if (GATE5_DB) { if (fq + 1 < NF) pair_blocks(fq + 1); }      // independent of the fill below: the scheduler interleaves the two
        else { pair_blocks(fq); wave_sync(); }
        const int np = np_g[fq];
        // ---- tile fill of feature fq: the packed-triangle index of every element this lane holds is a lane constant (fidx, set up
        //      once for the four features); the read is unconditional (padding elements read something valid and are overridden),
        //      only the selects depend on the feature's size ----
        const double* kp = sh.kp[GATE5_DB ? (fq & 1) : 0];
        const double* wq = sh.f[fq].w;
        bool jreal[NTL];
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) jreal[tj] = 16 * tj + l15 < np;
#ifndef GATE5_FILL_PIN1      // round 6: ALL reads of the feature's tiles are issued before the first select (one PIN4 per tile keeps them out
                             // of the selects' branches); -DGATE5_FILL_PIN1: pinned one by one (rounds 4-5) - a wait for LDS per element
#pragma unroll
        for (int t = 0; t < NLT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[fq][t][r] = kp[fidx[t][r]];
        double wvv[NTL];
#pragma unroll
        for (int tj = 0; tj < NTL; ++tj) wvv[tj] = wq[wcol[tj]];
#pragma unroll
        for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
                const int t = ti * (ti + 1) / 2 + tj;
                double b0 = T[fq][t][0], b1 = T[fq][t][1], b2 = T[fq][t][2], b3 = T[fq][t][3];
                PIN4(b0, b1, b2, b3);
                const double bvs[4] = { b0, b1, b2, b3 };
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ireal = 16 * ti + kq + 4 * r < np;
                    const double padv = (fdiag[t][r] && !ireal) ? 1.0 : 0.0;          // unit pivots on the padding rows
                    double v = (ireal && jreal[tj]) ? bvs[r] : padv;
                    if (ti == NTL - 1 && r == 3) v = kq == 3 ? (jreal[tj] ? wvv[tj] : 0.0) : v;      // row BR for kq == 3: the border row w^T, corner 0
                    T[fq][t][r] = v;
                }
            }
        }
#else
#pragma unroll
        for (int ti = 0; ti < NTL; ++ti) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ireal = 16 * ti + kq + 4 * r < np;
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj) {
                    const int t = ti * (ti + 1) / 2 + tj;
                    double bv = kp[fidx[t][r]];
                    asm volatile("" : "+v"(bv));                                      // keep the read out of the selects' branches
                    const double padv = (fdiag[t][r] && !ireal) ? 1.0 : 0.0;          // unit pivots on the padding rows
                    double v = (ireal && jreal[tj]) ? bv : padv;
                    if (ti == NTL - 1 && r == 3) {                                    // row BR for kq == 3: the border row w^T, corner 0
                        double wv = wq[wcol[tj]];
                        asm volatile("" : "+v"(wv));
                        v = kq == 3 ? (jreal[tj] ? wv : 0.0) : v;
                    }
                    T[fq][t][r] = v;
                }
            }
        }
#endif
        wave_sync();                          // triangle fq & 1 is free again; triangle (fq + 1) & 1 is complete
    }
    dbg_stamp(8);
#if defined(GATE5_STOP_AFTER) && GATE5_STOP_AFTER == 2      // front + pair blocks + tile fill
    {
        double acc[NF];
#pragma unroll
        for (int fq = 0; fq < NF; ++fq) {
            acc[fq] = 0.0;
#pragma unroll
            for (int q = 0; q < NLT; ++q) acc[fq] += T[fq][q][0] + T[fq][q][1] + T[fq][q][2] + T[fq][q][3];
            acc[fq] = wave_sum(acc[fq]);
            if (lane == 0 && j0 + fq < F) { gamma_out[(size_t)b * fv.fmax + j0 + fq] = acc[fq]; accept_out[(size_t)b * fv.fmax + j0 + fq] = 0; }
        }
        return;
    }
#endif
    // ================= four blocked LDL^T eliminations, interleaved (gate4_body's panel step per feature) =================
    // Padding rows carry unit pivots, so a feature with fewer observations simply eliminates identity panels: every feature runs
    // the panels of the largest one and no panel step branches on the feature.
    int npmax = 0;
#pragma unroll
    for (int q = 0; q < NF; ++q) npmax = np_g[q] > npmax ? np_g[q] : npmax;
    const int npan = (npmax + 3) >> 2;
    constexpr int KPAN = (NPMAX + 3) / 4;
    // Two tile rows (7 ... 11 clones): once the first tile column is eliminated only the 16 x 16 block (1, 1) is left - 14 pivot rows,
    // a padding row and the border row.  Taken through four more panels it costs four times the panel's fixed price (two LDS round
    // trips, the 4 x 4 factorisation, an MFMA) for a quarter of the first panels' work.  Instead every lane of group g takes ONE ROW of
    // feature g's block and the 15 pivots are eliminated in registers: the pivot row is broadcast inside the 16-lane row by DPP
    // (row_newbcast), no LDS, no barrier, one reciprocal per pivot.
#ifndef GATE5_FINISH
#define GATE5_FINISH 1
#endif
    constexpr bool FIN = GATE5_FINISH && NTL == 2;
#pragma unroll
    for (int k = 0; k < KPAN; ++k) {
        if (k < npan && !(FIN && k >= 4)) {
            const int tj0 = k >> 2, cb = 4 * (k & 3);
            if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
                for (int fq = 0; fq < NF; ++fq)
#pragma unroll
                    for (int ti = tj0; ti < NTL; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sh.pan[fq][16 * ti + kq + 4 * r][l15 - cb] = T[fq][ti * (ti + 1) / 2 + tj0][r];
            }
            wave_sync();
            const bool bord = (k == 4 * NTL - 1);
            // 4x4 LDL^T of the diagonal blocks: the 16 lanes of group g factorise FEATURE g's block (uniform within the group) - one
            // pass for the four features instead of one redundant pass per feature on all 64 lanes - and hand L^-1, r over through LDS
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
                // the last panel of the tile grid ends ON the border row: BR is its fourth row but not a pivot (see gate4_body)
                const double r3 = bord ? 0.0 : G5RCP(a4[3][3] - l30 * a4[3][0] - l31 * t31 - l32 * t32);
                // every lane only needs ITS row of the transformed panel, x_kq = (L^-1 m)_kq: hand over the rows of W = L^-1 (unit
                // lower; row kq = 4 coefficients) and the pivot reciprocals instead of L - a lane then forms its entry with 4
                // multiply-adds on what it reads, instead of the whole forward substitution (6) plus a 4-way selection (4)
                const double w20 = fma(l21, l10, -l20), w31 = fma(l32, l21, -l31);
                const double w30 = fma(-l32, w20, fma(l31, l10, -l30));
                if (sl < 4) {                               // lane sl of the group writes row sl of W (and r_sl)
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
                const double dsel = sh.lf[fq][16 + kq];                                         // this lane's pivot reciprocal (0: the border row)
                double A[NTL], B[NTL];
#pragma unroll
                for (int tt = tj0; tt < NTL; ++tt) {
                    const double2* pr = reinterpret_cast<const double2*>(sh.pan[fq][16 * tt + l15]);
                    const double2 u0 = pr[0], u1 = pr[1];
                    double xs = fma(w23.y, u1.y, fma(w23.x, u1.x, fma(w01.y, u0.y, w01.x * u0.x)));
                    if (16 * tt + l15 <= 4 * k + (bord ? 2 : 3)) xs = 0.0;   // pivot rows and everything above: finished
                    A[tt] = xs;
                    B[tt] = -xs * dsel;
                }
#pragma unroll
                for (int ti = tj0; ti < NTL; ++ti)
#pragma unroll
                    for (int tj = tj0; tj <= ti; ++tj)
                        T[fq][ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti], B[tj], T[fq][ti * (ti + 1) / 2 + tj], 0, 0, 0);
            }
            wave_sync();
        }
    }
    if (FIN && npan > 4) {
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
            // the border is only ever filled and updated as a ROW (the panels read the lower triangle): its column comes from the row
            if (sl < 15) row[15] = sh.fin[g][15][sl];
        }
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const double rinv = G5RCP(row_bcast_f64(row[i], i));      // pivot (i, i): lane i's diagonal element
            const double l = row[i] * rinv;                           // rows above the pivot hold zeros here: nothing happens to them
#pragma unroll
            for (int q = i + 1; q < 16; ++q) row[q] = fma(-l, row_bcast_f64(row[q], i), row[q]);      // the pivot row, by symmetry its column
        }
        // the corner (BR, BR) sits in lane 15 of every group: back to where the gate below expects it
#pragma unroll
        for (int fq = 0; fq < NF; ++fq) T[fq][NLT - 1][3] = __shfl(row[15], 16 * fq + 15, WAVE);
    }
    dbg_stamp(9);
    if (lane == WAVE - 1) {                       // lane (kq = 3, l15 = 15) holds element (BR, BR) = -w^T Kr^-1 w of every feature
#pragma unroll
        for (int fq = 0; fq < NF; ++fq) {
            if (j0 + fq < F) {
                const size_t oq = (size_t)b * fv.fmax + j0 + fq;
                if (!fok_g[fq]) { gamma_out[oq] = __builtin_nan(""); accept_out[oq] = 0; }
                else {
                    const double gval = -T[fq][NLT - 1][3] + sh.f[fq].rpsum / op.var;
                    const bool ok = gval < sh.f[fq].thr;                                        // Update.cpp:120 (threshold staged by the front)
                    gamma_out[oq] = gval;
                    accept_out[oq] = ok ? 1 : 0;
                }
            }
        }
    }
    dbg_stamp(10);
}
