// kernels_factored.hip — the structure-exploiting MSCKF path (same posterior as the dense K3..K11 path of
// kernels_msckf.hip / kernels_ekf.hip to FP64 rounding, at about a quarter of the FLOPs).
//
// Per feature the stacked Jacobian factors as  Hx = Gblk * D  (RemoveLostUpdate.cpp:474-501):
//   Gblk = blockdiag(G_o), G_o = Pi~_o R_o^T  (rows-per-obs x 3; stack(G_o) is also Hf)
//   D    = [3*nobs x 6C] sparse: row block o has [p_f]x at theta_o, -I at p_o, -[p_f]x at theta_anchor
// so, with V the left-nullspace basis of Hf (H_j = V^T Hx, r_j = V^T r):
//   K5   k_feat_gate3: S_j = V^T (Gblk Su Gblk^T + s^2 I) V,  Su = D Pcc D^T; projector identity
//        V (V^T S V)^-1 V^T = S^-1 - S^-1 Hf (Hf^T S^-1 Hf)^-1 Hf^T S^-1  and Woodbury down to the SPD matrix
//        K = Su + s^2 N^-1 of dimension 3 nobs (N = blockdiag(G_o^T G_o)); one bordered blocked LDL^T per feature.
//   K7   k_feat_gram2: the stacked-QR factor R only enters the update through A = R^T R = sum_j H_j^T H_j and
//        b = sum_j H_j^T r_j;  H_j^T H_j = sum_o D_o^T N_o D_o - B^T Ns^-1 B  (block-sparse minus rank 3, MFMA GEMM).
//   K8-K11  k_info_update / k_info_apply:  K H = Pc (A Pcc + s^2 I)^-1 A  (push-through identity; A may be singular,
//        rank n-6):  P <- P - (Pc M) Pc^T,  dx = Pc (A Pcc + s^2 I)^-1 b,  Pc = P[:, clone cols].
// gfx950 only.
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include "launch_factored.h"

#include "gate_kernel.h"
#include "gate5_kernel.h"
#include "gate5m_kernel.h"

// K3 + K5 for the window classes 6 / 11 / 16, see gate_kernel.h: workgroups of GATE_FPW waves = GATE_FPW features of one
// filter; with GATE_FPW > 1 wave 0 runs the per-observation front for all of them (64 / GATE_FPW lanes each), then one
// wave per feature.  Measured on MI355X (config 2): 1 -> 0.435 ms, 2 -> 0.435 ms, 4 -> 0.466 ms (the front is bound by
// the latency of its loads, not by issue slots), hence 1.  Tiles in VGPRs (no AGPR copies), 4 waves per SIMD.
#define GATE_FPW 1
template <int CMAX, bool STEREO>
__global__ __launch_bounds__(GATE_FPW * WAVE) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_feat_gate3(
    CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
    int* __restrict__ accept_out, double* __restrict__ rec_out)
{
    gate3_body<CMAX, STEREO, GATE_FPW, false>(cv, fv, op, b0, nb, fmax_used, gamma_out, accept_out, rec_out);
}

// Stereo windows up to 16 clones: the gate in difference coordinates of the observations (gate_kernel.h, gate4_body): a
// 3 (nobs - 1) + 1 bordered system, two 16-row tile rows for an 11-clone window instead of three.  One wave per (feature, filter).
#ifndef GATE4_WPE             // occupancy window of the gate, waves per SIMD.  Measured (512 filters): max 6 (the default: 100 VGPRs ->
#define GATE4_WPE 2           // 4 waves) 0.282 ms; forced to 5 (96 VGPRs, 12 B scratch) 0.291; forced to 6 (80 VGPRs, 36 B) 0.323;
                              // capped at 3 (114 VGPRs) 0.306, at 2: 0.390 - four waves per SIMD is the optimum
#endif
#ifndef GATE4_WPE_MAX
#define GATE4_WPE_MAX 6
#endif
template <int CMAX>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(GATE4_WPE, GATE4_WPE_MAX))) void k_feat_gate4(
    CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    gate4_body<CMAX>(cv, fv, op, b0, nb, fmax_used, gamma_out, accept_out);
}

// Stereo windows up to 11 clones (round 4): the same gate with FOUR features of a filter per wave (gate5_kernel.h) - the front on 16
// lanes per feature, every pair lane's block of P loaded once for the four features, four interleaved eliminations.
#ifndef GATE5_WPE
#define GATE5_WPE 2
#endif
template <int CMAX>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(GATE5_WPE, GATE5_WPE))) void k_feat_gate5(
    CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    gate5_body<CMAX>(cv, fv, op, b0, nb, fmax_used, gamma_out, accept_out);
}

// Mono windows up to 11 clones (round 6): the measurement-space gate as a quasi-definite bordered system on the same machinery
// (gate5m_kernel.h) - four features per wave, two tile rows instead of gate3's three.
template <int CMAX>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(GATE5_WPE, GATE5_WPE))) void k_feat_gate5m(
    CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    gate5m_body<CMAX>(cv, fv, op, b0, nb, fmax_used, gamma_out, accept_out);
}

// ---------------------------------------------------------------------------------------------
// K4 + K6/K7 in Gram form, second generation: block-sparse part + rank-3 MFMA part.
//
// With D_o = [cn X | -pl I | -cn X] on (theta_c, p_c, theta_anchor) (X = [p_f]x, c = slot of obs o),
// N = blockdiag(N_o), Ns = sum_o N_o (= Hf^T Hf) the nullspace-projected information of one feature is
//     H_j^T H_j = D^T (N - N 1 Ns^-1 1^T N) D = sum_o D_o^T N_o D_o  -  B^T Ns^-1 B,     B = sum_o N_o D_o  (3 x 6C)
//     H_j^T r_j = sum_o D_o^T h_o - B^T Ns^-1 hs,                                      hs = sum_o h_o
// i.e. a block-sparse term (6x6 at the observing clone, couplings to the anchor's theta block) minus a
// RANK-3 term.  Summed over a chunk's features the rank-3 terms are one GEMM  Y^T [B | hs]  with
// Y = Ns^-1 B stacked over features (3 rows each): it runs on the matrix cores
// (v_mfma_f64_16x16x4, 4 stacked rows per instruction), no per-pair 3x3 algebra at all.
// The sparse term only needs, per (observing slot c, anchor slot a), the running sums of
// cn X^T N X, cn pl N X, pl N, cn X^T h, pl h: lane (c, a) keeps them in registers and the chunk's
// 6C x (6C+1) partial [A | b] is assembled once at the end.
// grid = (G chunks, nb); a workgroup walks its chunk in batches of GRAM_NB features.
// ---------------------------------------------------------------------------------------------
#define GRAM_NB 8
#define GRAM_NT 256
#ifndef GRAM2_ZFORM
#define GRAM2_ZFORM 1      // the rank-3 term as Z^T Z (one operand panel); 0: Y^T B with Y = Ns^-1 B (rounds 2-5)
#endif
// 1 / sqrt(x) to full precision: v_rsq_f64 + two Newton steps (as kernels_chol.hip's)
__device__ __forceinline__ double gram_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);
    e = fma(-x * y, y, 1.0);
    return fma(y * e, 0.5, y);
}
template <int CMAX>
struct Gram2Cfg {
    static constexpr int NC = 6 * CMAX;
    static constexpr int TI = (NC + 15) / 16;              // tiles over rows of A (columns of Y)
    static constexpr int TJ = (NC + 1 + 15) / 16;          // tiles over columns of [A | b]
    static constexpr int LDW = 16 * TJ;
    static constexpr int NTILE = TI * TJ;
    static constexpr int NUP = TI * TJ - TI * (TI - 1) / 2;   // tiles (ti <= tj): the rank-3 Gram is symmetric
    static constexpr int TPW = (NUP + 3) / 4;              // accumulator tiles per wave
    static constexpr int KR = 3 * GRAM_NB;                 // stacked rows per batch
    static constexpr int SPW = 34;                         // per (feature, slot) sparse scratch: S1 NXs S3 (9 each) s4 s5 (3 each) key
};
template <int CMAX>
struct Gram2Batch {
    using Cfg = Gram2Cfg<CMAX>;
    double Bm[Cfg::KR][Cfg::LDW];
    double Ym[Cfg::KR][Cfg::LDW];
    double sp[CMAX * CMAX <= 128 ? 2 : 1][GRAM_NB][CMAX][Cfg::SPW];      // double-buffered when P3b runs beside the next batch's P2
};
template <int CMAX>
struct Gram2Out {
    using Cfg = Gram2Cfg<CMAX>;
    double A2[Cfg::NC][Cfg::NC + 2];         // rank-3 part (+ hs column), row stride NC+2
    double S[CMAX][CMAX][34];                // per (slot, anchor) sparse sums: S1 NXs S3 s4 s5
};

template <int CMAX, bool STEREO>
__global__ __launch_bounds__(GRAM_NT, 2) void k_feat_gram2(
    FrameView fv, MsckfOpts op, int b0, const int* __restrict__ accept_in, int* __restrict__ used_out,
    double* __restrict__ Apart, int* __restrict__ chunk_used, int G, int rstride)
{
    using Cfg = Gram2Cfg<CMAX>;
    constexpr int NC = Cfg::NC, TJ = Cfg::TJ, LDW = Cfg::LDW, NTILE = Cfg::NTILE, TPW = Cfg::TPW, KR = Cfg::KR;
    constexpr int NUP = Cfg::NUP, TI = Cfg::TI, RPO = STEREO ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Gram2Batch<CMAX>& sb = *reinterpret_cast<Gram2Batch<CMAX>*>(smem_raw);
    Gram2Out<CMAX>& so = *reinterpret_cast<Gram2Out<CMAX>*>(smem_raw);             // epilogue view of the same LDS
    constexpr size_t UNI = sizeof(Gram2Batch<CMAX>) > sizeof(Gram2Out<CMAX>) ? sizeof(Gram2Batch<CMAX>) : sizeof(Gram2Out<CMAX>);
    int* sUse = reinterpret_cast<int*>(smem_raw + ((UNI + 15) / 16) * 16);
    int* sList = sUse + fv.fmax;
    __shared__ int sNu;
    __shared__ double sPose[16][12];                              // the window's clone poses: R (9, row-major), p (3)
    // wave as a scalar: the tile coordinates of a wave's accumulators are wave-uniform (see k_feat_gram_big)
    const int bl = blockIdx.y, b = b0 + bl, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = fv.n_feat[b], C = fv.n_clones[b], ncol = 6 * C;
    for (int e = tid; e < 12 * C; e += GRAM_NT) {
        const int c = e / 12, q = e - 12 * c;
        sPose[c][q] = q < 9 ? fv.clone_R[((size_t)b * fv.cmax + c) * 9 + q] : fv.clone_p[((size_t)b * fv.cmax + c) * 3 + q - 9];
    }

    dbg_stamp(32);
    for (int j = tid; j < F; j += GRAM_NT) {                    // RemoveLostUpdate.cpp:357-359
        int use = accept_in[(size_t)b * fv.fmax + j];
        if (use && op.max_accept > 0) {
            int rank = 0;
            for (int q = 0; q < j; ++q) rank += accept_in[(size_t)b * fv.fmax + q];
            if (rank >= op.max_accept) use = 0;
        }
        sUse[j] = use;
        if (g == 0) used_out[(size_t)b * fv.fmax + j] = use;
    }
    // zero the operand panels once: padding columns (and rows of a short last batch) stay zero
    for (int e = tid; e < KR * LDW; e += GRAM_NT) { (&sb.Bm[0][0])[e] = 0.0; (&sb.Ym[0][0])[e] = 0.0; }
    __syncthreads();
    if (wave == 0) {                                            // ordered list of the used features
        int cnt = 0;
        for (int base = 0; base < F; base += WAVE) {
            const int j = base + lane;
            const bool u = j < F && sUse[j];
            const unsigned long long m = __ballot(u);
            if (u) sList[cnt + __popcll(m & ((1ULL << lane) - 1ULL))] = j;
            cnt += __popcll(m);
        }
        if (lane == 0) sNu = cnt;
    }
    __syncthreads();
    const int nu = sNu, per = (nu + G - 1) / G;
    const int q0 = g * per, q1 = min(nu, q0 + per);

    dbg_stamp(33);
    // MFMA accumulators: wave w owns the upper tiles t = w, w+4, ... (row-major over ti <= tj)
    int tiA[TPW], tjA[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        int t = wave + 4 * u, ti = 0;
        while (ti < TI - 1 && t >= TJ - ti) { t -= TJ - ti; ++ti; }
        tiA[u] = ti; tjA[u] = ti + t;                          // t >= NUP gives tj >= TJ: never launched
    }
    double4_f acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = double4_f{ 0.0, 0.0, 0.0, 0.0 };
    // sparse accumulators of lane (c, a)
    constexpr bool OVL = CMAX * CMAX <= 128;                 // pair lanes fit waves 2-3: P3b overlaps the next batch's P2
    const int ptid = OVL ? tid - 128 : tid;
    const int pc = ptid >= 0 ? ptid / CMAX : 0, pa = ptid >= 0 ? ptid - pc * CMAX : 0;
    const bool pairlane = ptid >= 0 && ptid < CMAX * CMAX;
    double sS1[9], sNX[9], sS3[9], s4[3], s5[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) { sS1[i] = 0.0; sNX[i] = 0.0; sS3[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { s4[i] = 0.0; s5[i] = 0.0; }
    // Round 6 (P3U, window classes whose pair lanes sit on waves 2-3): the sparse sums by lane (slot c, CHUNK j of the 33 values) with
    // one accumulator triple per ANCHOR, instead of lane (slot c, anchor a) with all 33 values.  A feature has ONE anchor - wave-uniform
    // while the wave walks the batch feature by feature - so a lane adds its three values into the triple of that anchor behind a
    // scalar branch: 3 additions per feature.  Lane (c, a) multiplied ten of its eleven lanes' 33 values by zero (the key selected the
    // one anchor): 33 FMAs + 34 LDS values per feature and lane - by the stamps of k_feat_gram3 the sparse sums were the longest
    // issue stream of a batch (3.9 k of ~10 k cycles per SIMD).  Same additions in the same order: bit-identical sums.
#ifndef GRAM2_P3B_UNIFORM
#define GRAM2_P3B_UNIFORM 1
#endif
    constexpr bool P3U = GRAM2_P3B_UNIFORM && CMAX * CMAX <= 128 && 11 * CMAX <= 128;      // 11 chunks of 3 = the 33 values of a slot, on waves 2-3
    double sacc[3 * CMAX];
#pragma unroll
    for (int i = 0; i < 3 * CMAX; ++i) sacc[i] = 0.0;
    const int uc = ptid >= 0 ? ptid / 11 : 0, uj = ptid >= 0 ? ptid - 11 * uc : 0;      // P3U lane: (slot, chunk of three values), 11 chunks per slot
    const bool ulane = P3U && ptid >= 0 && ptid < 11 * CMAX;
    const int kq = lane >> 4, l15 = lane & 15;

    // The per-observation quantities (N_o = G_o^T G_o, h_o = G_o^T r_o) are recomputed here from the frame inputs (one projection
    // per (feature, slot) lane) instead of travelling through a 1.5 KB per-feature record written by the gate kernel: the record
    // cost 113 MiB of HBM writes + 113 MiB of reads per launch of the 512-filter batch, the recomputation ~25 VALU per feature.
    // Lane (f, c) = (tid >> 4, tid & 15) of the first 16 GRAM_NB threads; its raw inputs for the NEXT batch are fetched into
    // registers while the matrix cores run the current one.
    double in_uv[4], in_pf[3];
    unsigned long long in_mask = 0ULL;
    int in_anchor = 0;
    // Round 6, measured and left OFF (-DGRAM_SPLIT=1): waves 2-3 take the SPARSE scratch of the next batch off waves 0-1 - they project
    // the same (feature, slot) lanes again (~120 FP64 operations) and write sp[..] themselves, after their own P3b.  Shader-clock
    // stamps of a batch: P3a 2.5 k cycles (all waves), then P2 9.2 k on waves 0-1 beside P3b 3.5 k on waves 2-3 - half of the workgroup
    // idle for 5.7 k of a batch's 11.7 k cycles.  But the kernel sits AT its register budget (256 VGPRs at two workgroups per CU): with
    // the projection's temporaries live beside the 33 sparse accumulators the allocator spills the LDS offsets of the MFMA loop (88 B
    // of scratch, reloaded in front of every product): 106 -> 158 us per 512 filters.  Handing N_o, h_o over through LDS instead needs
    // a second pair of buffers the 80 KB of a workgroup do not have.
#ifndef GRAM_SPLIT
#define GRAM_SPLIT 0
#endif
    constexpr bool SPLIT = GRAM_SPLIT && CMAX * CMAX <= 128;
    const int ft = SPLIT ? (tid & 127) : tid;                 // (feature, slot) lane of the operand-row phase
    auto fetch = [&](int qb) {
        const int f = ft >> 4, c = ft & 15;
        in_mask = 0ULL; in_anchor = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) in_uv[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) in_pf[i] = 0.0;
        if (ft < GRAM_NB * 16 && qb + f < q1) {
            const size_t oidx = (size_t)b * fv.fmax + sList[qb + f];
            in_mask = fv.obs_mask[oidx]; in_anchor = fv.anchor[oidx];
#pragma unroll
            for (int i = 0; i < 3; ++i) in_pf[i] = fv.pf[oidx * 3 + i];
            if (c < C && ((in_mask >> c) & 1ULL)) {
                const double* z = fv.uv + (oidx * fv.cmax + c) * 4;
                in_uv[0] = z[0]; in_uv[1] = z[1];
                if (STEREO) { in_uv[2] = z[2]; in_uv[3] = z[3]; }
            }
        }
    };
    // sum over the 16 lanes of a feature, left in all of them: rotations within the DPP row (row_ror:8/4/2/1 on the two halves of the
    // double) instead of __shfl_xor, which compiles to two ds_bpermute_b32 per step - 120 LDS-pipe instructions per lane and batch
    // for the 15 sums of P2, the phase the batch waits for
    auto sum16 = [](double v) {
#ifdef GRAM_SHFL_SUM
        v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
#else
        v += row_ror_f64<8>(v); v += row_ror_f64<4>(v); v += row_ror_f64<2>(v); v += row_ror_f64<1>(v);
#endif
        return v;
    };
    fetch(q0);
    // P2 (operand rows + sparse scratch of one batch, lanes (feature, slot) of waves 0-1), P3a (rank-3 part, all waves), P3b (sparse
    // sums, pair lanes).  Window classes with at most 128 (slot, anchor) pairs put the pair lanes on waves 2-3 and double-buffer the
    // sparse scratch, so that P3b of batch i runs BESIDE P2 of batch i+1 instead of after it (P2 6.3 k, P3b 3.5 k of a batch's 14.8 k cycles).
    // rows / sparse: which half of P2 the caller wants (both: the first batch and the window classes without the split)
    auto do_p2 = [&](int qb, int buf, int nthr, bool rows, bool sparse) {      // nthr: threads that make the call (256, or the 128 of waves 0-1)
        const int nbf = min(GRAM_NB, q1 - qb);
        dbg_stamp(34);
        if (rows && nbf < GRAM_NB) {                           // short last batch: clear the unused stacked rows
            for (int e = tid; e < (KR - 3 * nbf) * LDW; e += nthr) { (&sb.Bm[3 * nbf][0])[e] = 0.0; (&sb.Ym[3 * nbf][0])[e] = 0.0; }
        }
        dbg_stamp(35);
        // ---- P2: operand rows B, Y = Ns^-1 B and the sparse scratch, lane = (feature, window slot) -----
        if (ft < nbf * 16) {
            const int f = ft >> 4, c = ft & 15;
            const int a = in_anchor;
            const double px = in_pf[0], py = in_pf[1], pz = in_pf[2];
            bool obs = c < C && ((in_mask >> c) & 1ULL);
            double N[9], h[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) N[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) h[i] = 0.0;
            if (obs) {                                          // RemoveLostUpdate.cpp:435-506 for this (feature, clone)
                double Gm[RPO][3], rs[RPO];
                obs = feat_obs<STEREO>(sPose[c], sPose[c] + 9, in_uv, px, py, pz, op, Gm, rs);      // false: skipped by the NaN guard (:486)
                if (obs) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
#pragma unroll
                        for (int m2 = 0; m2 < 3; ++m2) {
                            double sN = 0.0;
#pragma unroll
                            for (int q = 0; q < RPO; ++q) sN += Gm[q][m] * Gm[q][m2];
                            N[3 * m + m2] = sN;
                        }
                        double hh = 0.0;
#pragma unroll
                        for (int q = 0; q < RPO; ++q) hh += Gm[q][m] * rs[q];
                        h[m] = hh;
                    }
                }
            }
            const double cn = (obs && c != a) ? 1.0 : 0.0, pl = (obs && !(op.selected_variant && c == a)) ? 1.0 : 0.0;
            // Ns = sum_o N_o (= Hf^T Hf), hs = sum_o h_o, Nsa = sum over the observations whose clone is not the anchor
            double Ns[9], hs[3], Nsa[9];
            if (rows) {
                const double n0 = sum16(N[0]), n1 = sum16(N[1]), n2 = sum16(N[2]), n4 = sum16(N[4]), n5 = sum16(N[5]), n8 = sum16(N[8]);
                Ns[0] = n0; Ns[1] = n1; Ns[2] = n2; Ns[3] = n1; Ns[4] = n4; Ns[5] = n5; Ns[6] = n2; Ns[7] = n5; Ns[8] = n8;
#ifdef GRAM_NSA_SUMS
                const double a0 = sum16(cn * N[0]), a1 = sum16(cn * N[1]), a2 = sum16(cn * N[2]), a4 = sum16(cn * N[4]), a5 = sum16(cn * N[5]),
                             a8 = sum16(cn * N[8]);
#else
                // the only observation with cn = 0 is the one AT the anchor slot: Nsa = Ns - N_anchor, the anchor lane's N fetched with
                // six 64-bit shuffles instead of six more 16-lane sums (72 VALU instructions of the phase the batch waits for)
                const int alane = (lane & 48) | (a & 15);
                const double a0 = n0 - __shfl(N[0], alane, WAVE), a1 = n1 - __shfl(N[1], alane, WAVE), a2 = n2 - __shfl(N[2], alane, WAVE),
                             a4 = n4 - __shfl(N[4], alane, WAVE), a5 = n5 - __shfl(N[5], alane, WAVE), a8 = n8 - __shfl(N[8], alane, WAVE);
#endif
                Nsa[0] = a0; Nsa[1] = a1; Nsa[2] = a2; Nsa[3] = a1; Nsa[4] = a4; Nsa[5] = a5; Nsa[6] = a2; Nsa[7] = a5; Nsa[8] = a8;
#pragma unroll
                for (int i = 0; i < 3; ++i) hs[i] = sum16(h[i]);
            }
            if (c < C) {
                double NX[9];
                mulX(N, px, py, pz, NX);                            // N_o X
              if (rows) {
                double Bt[9], Bp[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) { Bt[i] = cn * NX[i]; Bp[i] = -pl * N[i]; }
                if (c == a) {                                       // theta_anchor block: -Nsa X   (the anchor's own cn is 0)
                    double T[9];
                    mulX(Nsa, px, py, pz, T);
#pragma unroll
                    for (int i = 0; i < 9; ++i) Bt[i] = -T[i];
                }
#if GRAM2_ZFORM
                // Round 6: the rank-3 term as a SYMMETRIC product.  With Ns = L D L^T (3 x 3, SPD for a used feature)
                //     B^T Ns^-1 B = Z^T Z,   Z = D^-1/2 L^-1 B      (the hs column rides as D^-1/2 L^-1 hs)
                // - ONE operand panel (Bm holds Z, Ym is not used): the forward substitution replaces inv3sym and two 3 x 3 products and
                // halves the panel stores (P2: 9.2 k -> 6.3 k cycles per batch, shader-clock stamps of k_feat_gram3, gram3_kernel.h)
                const double s0 = Ns[0] > 0.0 ? gram_rsqrt(Ns[0]) : 0.0, r0 = s0 * s0;          // s_k = d_k^-1/2 (0: a pivot that is not positive)
                const double l10 = Ns[1] * r0, l20 = Ns[2] * r0;
                const double d1 = Ns[4] - l10 * Ns[1], s1 = d1 > 0.0 ? gram_rsqrt(d1) : 0.0, r1 = s1 * s1;
                const double t21 = Ns[5] - l20 * Ns[1], l21 = t21 * r1;
                const double d2 = (Ns[8] - l20 * Ns[2]) - l21 * t21, s2 = d2 > 0.0 ? gram_rsqrt(d2) : 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    {
                        const double y0 = Bt[q], y1 = Bt[3 + q] - l10 * y0, y2 = (Bt[6 + q] - l20 * y0) - l21 * y1;
                        sb.Bm[3 * f + 0][6 * c + q] = s0 * y0; sb.Bm[3 * f + 1][6 * c + q] = s1 * y1; sb.Bm[3 * f + 2][6 * c + q] = s2 * y2;
                    }
                    {
                        const double y0 = Bp[q], y1 = Bp[3 + q] - l10 * y0, y2 = (Bp[6 + q] - l20 * y0) - l21 * y1;
                        sb.Bm[3 * f + 0][6 * c + 3 + q] = s0 * y0; sb.Bm[3 * f + 1][6 * c + 3 + q] = s1 * y1; sb.Bm[3 * f + 2][6 * c + 3 + q] = s2 * y2;
                    }
                }
                if (c == 0) {
                    const double y0 = hs[0], y1 = hs[1] - l10 * y0, y2 = (hs[2] - l20 * y0) - l21 * y1;
                    sb.Bm[3 * f + 0][NC] = s0 * y0; sb.Bm[3 * f + 1][NC] = s1 * y1; sb.Bm[3 * f + 2][NC] = s2 * y2;
                }
#else
                double Nsi[9];
                inv3sym(Ns, Nsi);
                double Yt[9], Yp[9];
                mul33(Nsi, Bt, Yt);
                mul33(Nsi, Bp, Yp);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        sb.Bm[3 * f + k][6 * c + q] = Bt[3 * k + q];
                        sb.Bm[3 * f + k][6 * c + 3 + q] = Bp[3 * k + q];
                        sb.Ym[3 * f + k][6 * c + q] = Yt[3 * k + q];
                        sb.Ym[3 * f + k][6 * c + 3 + q] = Yp[3 * k + q];
                    }
                if (c == 0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) sb.Bm[3 * f + k][NC] = hs[k];      // extra column: hs
                }
#endif
              }
              if (sparse) {
                // sparse scratch
                double* sp = sb.sp[buf][f][c < CMAX ? c : 0];
                double S1[9];
                mulXt(NX, px, py, pz, S1);                          // X^T N X
#pragma unroll
                for (int i = 0; i < 9; ++i) { sp[i] = cn * S1[i]; sp[9 + i] = cn * pl * NX[i]; }
                sp[18] = cn * (pz * h[1] - py * h[2]);              // X^T h_o = h_o x p_f
                sp[19] = cn * (px * h[2] - pz * h[0]);
                sp[20] = cn * (py * h[0] - px * h[1]);
#pragma unroll
                for (int i = 0; i < 9; ++i) sp[21 + i] = pl * N[i];                      // pl N_o
#pragma unroll
                for (int i = 0; i < 3; ++i) sp[30 + i] = pl * h[i];                      // pl h_o
                sp[33] = (obs || c == 0) ? (double)a : -1.0;          // key: the anchor slot this contribution belongs to (slot 0 always carries it: P3U reads the feature's anchor there; its values are zero when it does not observe)
              }
            }
        }
    };
    auto do_p3a = [&](int qb) {
        const int nbf = min(GRAM_NB, q1 - qb);
        // ---- P3a: rank-3 part on the matrix cores (next batch's records are fetched meanwhile) -------------
        const int nst = (3 * nbf + 3) >> 2;
#pragma unroll
        for (int st = 0; st < KR / 4; ++st) {                // fully unrolled: the fragment reads of the later steps are
            if (st < nst) {                                  // issued while the earlier MFMAs run
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    if (wave + 4 * u < NUP) {
                        const int ti = tiA[u], tj = tjA[u];
                        const double af = (GRAM2_ZFORM ? sb.Bm : sb.Ym)[4 * st + kq][16 * ti + l15];      // A[i][k] = Y[k][i]  (Z-form: Z[k][i])
                        const double bf = sb.Bm[4 * st + kq][16 * tj + l15];      // B[k][j]
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[u], 0, 0, 0);
                    }
                }
            }
        }
    };
    auto do_p3b = [&](int qb, int buf) {
        const int nbf = min(GRAM_NB, q1 - qb);
        // ---- P3b: sparse part, lane (c, a): branch-free, one level of LDS reads (the key says whose anchor it is) ----
        if (P3U) {
            if (ulane) {                                        // lane (slot uc, chunk uj): values 3 uj .. 3 uj + 2 of the slot's 33
#pragma unroll
                for (int f = 0; f < GRAM_NB; ++f) {
                    if (f < nbf) {
                        const double* sp = sb.sp[buf][f][uc];
                        const int af = __builtin_amdgcn_readfirstlane((int)sb.sp[buf][f][0][33]);      // the feature's anchor slot (slot 0's key always carries it)
                        const double x0 = sp[3 * uj], x1 = sp[3 * uj + 1], x2 = sp[3 * uj + 2];       // zero for a slot that does not observe the feature
#pragma unroll
                        for (int a2 = 0; a2 < CMAX; ++a2) {             // wave-uniform selector per anchor (a scalar), three multiply-adds: no branch
                            const double m = af == a2 ? 1.0 : 0.0;     // (as a switch on the scalar the compiler copied the other thirty
                            sacc[3 * a2] = fma(m, x0, sacc[3 * a2]);    //  accumulators around every case block: 66 v_mov_b64 per feature)
                            sacc[3 * a2 + 1] = fma(m, x1, sacc[3 * a2 + 1]);
                            sacc[3 * a2 + 2] = fma(m, x2, sacc[3 * a2 + 2]);
                        }
                    }
                }
            }
        } else if (pairlane) {
#pragma unroll
            for (int f = 0; f < GRAM_NB; ++f) {
                if (f < nbf) {
                    const double* sp = sb.sp[buf][f][pc];
                    const double m = sp[33] == (double)pa ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < 9; ++i) { sS1[i] = fma(m, sp[i], sS1[i]); sNX[i] = fma(m, sp[9 + i], sNX[i]); sS3[i] = fma(m, sp[21 + i], sS3[i]); }
#pragma unroll
                    for (int i = 0; i < 3; ++i) { s4[i] = fma(m, sp[18 + i], s4[i]); s5[i] = fma(m, sp[30 + i], s5[i]); }
                }
            }
        }
    };
    if (OVL) {
        if (q0 < q1) { if (!SPLIT || wave < 2) do_p2(q0, 0, SPLIT ? 128 : GRAM_NT, true, true); fetch(q0 + GRAM_NB); }
        lds_barrier();
        int it = 0;
        for (int qb = q0; qb < q1; qb += GRAM_NB, ++it) {
            if (it == 3) dbg_stamp(36);
            do_p3a(qb);
            if (it == 3) dbg_stamp(37);
            lds_barrier();                                      // every wave is done with this batch's operand rows
            if (it == 3) dbg_stamp(38);
            if (wave < 2) { if (qb + GRAM_NB < q1) { do_p2(qb + GRAM_NB, (it + 1) & 1, 128, true, !SPLIT); fetch(qb + 2 * GRAM_NB); } }
            else {
                do_p3b(qb, it & 1);
                if (SPLIT && qb + GRAM_NB < q1) { do_p2(qb + GRAM_NB, (it + 1) & 1, 128, false, true); fetch(qb + 2 * GRAM_NB); }
            }
            if (it == 3) dbg_stamp(41);
            lds_barrier();
            if (it == 3) dbg_stamp(42);
        }
    } else {
        for (int qb = q0; qb < q1; qb += GRAM_NB) {
            do_p2(qb, 0, GRAM_NT, true, true);
            lds_barrier();                                      // LDS hand-over only: the next batch's input loads stay in flight
            fetch(qb + GRAM_NB);
            do_p3a(qb);
            do_p3b(qb, 0);
            lds_barrier();
        }
    }

    dbg_stamp(39);
    // ---- epilogue: assemble [A | b] of the chunk ---------------------------------------------------
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        if (wave + 4 * u < NUP) {
            const int ti = tiA[u], tj = tjA[u];
            const int jc = 16 * tj + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r;      // C/D: col = lane&15, row = (lane>>4)+4r
                if (i < NC && jc <= NC) so.A2[i][jc] = acc[u][r];
            }
        }
    }
    if (P3U) {
        if (ulane) {                                            // scratch order (S1, NX, X^T h, pl N, pl h) -> the epilogue's (S1, NX, pl N, X^T h, pl h)
#pragma unroll
            for (int a2 = 0; a2 < CMAX; ++a2)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int v = 3 * uj + i;
                    so.S[uc][a2][v < 18 ? v : (v < 21 ? v + 9 : (v < 30 ? v - 3 : v))] = sacc[3 * a2 + i];
                }
        }
    } else if (pairlane) {
        double* S = so.S[pc][pa];
#pragma unroll
        for (int i = 0; i < 9; ++i) { S[i] = sS1[i]; S[9 + i] = sNX[i]; S[18 + i] = sS3[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { S[27 + i] = s4[i]; S[30 + i] = s5[i]; }
    }
    __syncthreads();
    double* out = Apart + ((size_t)bl * G + g) * rstride;      // [ncol][ncol+1] row-major, b in the last column
    if (pairlane && pc < C && pa < C) {
        const int c = pc, c2 = pa;
        double blk[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) blk[i] = 0.0;
        double bb[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        if (c == c2) {
            for (int a = 0; a < C; ++a) {
                const double* S = so.S[c][a];          // obs at slot c, anchor a
                const double* Sa = so.S[a][c];         // obs at slot a, anchor c  -> (theta_c, theta_c) += S1
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int q2 = 0; q2 < 3; ++q2) {
                        blk[6 * q + q2] += S[3 * q + q2] + Sa[3 * q + q2];
                        blk[6 * q + 3 + q2] -= S[9 + 3 * q2 + q];            // (theta,p) = -NXs^T
                        blk[6 * (3 + q) + q2] -= S[9 + 3 * q + q2];          // (p,theta) = -NXs
                        blk[6 * (3 + q) + 3 + q2] += S[18 + 3 * q + q2];
                    }
#pragma unroll
                for (int q = 0; q < 3; ++q) { bb[q] += S[27 + q] - Sa[27 + q]; bb[3 + q] -= S[30 + q]; }
            }
        } else {
            const double* S = so.S[c][c2];             // obs at slot c, anchor c2
            const double* St = so.S[c2][c];            // obs at slot c2, anchor c
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 3; ++q2) {
                    blk[6 * q + q2] = -S[3 * q + q2] - St[3 * q + q2];       // S1 is symmetric
                    blk[6 * (3 + q) + q2] = S[9 + 3 * q + q2];               // (p_c, theta_a) = +NXs
                    blk[6 * q + 3 + q2] = St[9 + 3 * q2 + q];                // (theta_a, p_c') = +NXs^T
                }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int q2 = 0; q2 < 6; ++q2)
                {
                const int ri = 6 * c + q, rj = 6 * c2 + q2;
                // only tiles ti <= tj were accumulated: element (ri, rj) with ri/16 > rj/16 is read from its mirror
                const double a2 = (ri >> 4) <= (rj >> 4) ? so.A2[ri][rj] : so.A2[rj][ri];
                out[(size_t)ri * (ncol + 1) + rj] = blk[6 * q + q2] - a2;
            }
            if (c == c2) out[(size_t)(6 * c + q) * (ncol + 1) + ncol] = bb[q] - so.A2[6 * c + q][NC];
        }
    }
    if (tid == 0) chunk_used[bl * G + g] = max(0, q1 - q0);
    dbg_stamp(40);
}

#ifdef INGVIO_ALT_KERNELS      // k_feat_gram3 (round 6, measured and rejected: 115 against 107 us per 512 filters): one operand panel (Z^T Z),
#include "gram3_kernel.h"      // double-buffered, one barrier per batch - variant build only, INGVIO_GRAM=3
#endif

// ---------------------------------------------------------------------------------------------
// K8/K9/K11 in information form, one workgroup per filter:
//   [M | t] = (A Pcc + s^2 I)^-1 [A | b]  by Gauss-Jordan with partial pivoting in LDS,
//   T = Pc M (-> Tout), Pc copy (-> Pcout), dx = Pc t.   P <- P - T Pc^T is done by k_downdate.
// ---------------------------------------------------------------------------------------------
#define INFO_NT 512

// The solve always runs at the compile-time size NC = 6 * (c_max class): clones missing from the
// window are zero rows/columns of A, for which K1 = s^2 I and M = 0 (harmless), so no loop needs a
// runtime guard and every register array is fully used.
template <int NC>
__global__ __launch_bounds__(INFO_NT, 2) void k_info_update(
    CovView cv, FrameView fv, int b0, const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
    const double* __restrict__ noise_all, double* __restrict__ Mall, int mstride, double* __restrict__ Pcall, int ystride,
    double* __restrict__ dx_all, int* __restrict__ m_out, int* __restrict__ nc_out, int* __restrict__ status,
    const int* __restrict__ marg_idx, int* __restrict__ pc_base_out)
{
    constexpr int LA = 2 * NC + 1, MP = (NC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* aug = reinterpret_cast<double*>(smem_raw);
    int* sCol = reinterpret_cast<int*>(aug + (size_t)NC * LA);
    __shared__ unsigned long long sBest[2];
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int C = fv.n_clones[b], ncol = 6 * C, n = cv.n[b], ld = cv.ldp;
    double* dx = dx_all + (size_t)b * ld;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        for (int r = tid; r < n; r += INFO_NT) dx[r] = 0.0;
        if (tid == 0) { m_out[bl] = 0; nc_out[bl] = ncol; pc_base_out[bl] = -1; }
        return;
    }
    const double* P = cov_ptr(cv, b);
    const double var = noise_all[bl];
    dbg_stamp(0);
    for (int c = tid; c < NC; c += INFO_NT) { const int cc = c < ncol ? c : 0; sCol[c] = fv.clone_idx[(size_t)b * fv.cmax + cc / 6] + cc % 6; }
    if (tid < 2) sBest[tid] = 0ULL;
    const int tx = tid & 63, ty = tid >> 6;               // 64 x 8 thread grid: no runtime div/mod in the loops
    for (int i = ty; i < NC; i += INFO_NT / 64) {
#pragma unroll
        for (int jq = 0; jq < (NC + 1 + 63) / 64; ++jq) {
            const int j = tx + 64 * jq;
            if (j > NC) continue;
            const int jj = j == NC ? ncol : j;                  // b lives in column ncol of the partials
            double s = 0.0;
            if (i < ncol && (j < ncol || j == NC)) {
                const size_t e = (size_t)i * (ncol + 1) + jj;
                for (int g0 = 0; g0 < G; g0 += 4) {             // four partial loads in flight
                    double t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int g = g0 + u;
                        t[u] = (g < G && chunk_used[bl * G + g]) ? Apart[((size_t)bl * G + g) * rstride + e] : 0.0;
                    }
                    s += (t[0] + t[1]) + (t[2] + t[3]);
                }
            }
            aug[i * LA + NC + j] = s;
        }
    }
    __syncthreads();
    dbg_stamp(1);
    // zero-copy operand for k_info_apply: when the update is written out of place (fused marginalisation) and the
    // window's clones are one contiguous column block, Pc is P itself and the copy below is skipped
    const bool fused = marg_idx && marg_idx[bl] >= 0;
    const int contig = __syncthreads_and(tid >= ncol || sCol[tid < NC ? tid : 0] == sCol[0] + tid);
    const bool zero_copy = fused && contig && sCol[0] + ((NC + 3) & ~3) <= ld;
    // K1 = A Pcc + s^2 I.  Lane j keeps column j of Pcc in registers (NC independent, coalesced loads: one
    // memory latency instead of a dependent chain); KG row groups share the rows.
    {
        constexpr int KG = INFO_NT / NC;
        const int j = tid % NC, g = tid / NC;
        if (g < KG) {
            constexpr int NH = (NC + 1) / 2;                   // two halves of the k range: half the registers
            const int gj = sCol[j];
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int k0 = h * NH;
                double pc[NH];
#pragma unroll
                for (int k = 0; k < NH; ++k) pc[k] = (k0 + k < NC) ? P[gj + (size_t)sCol[k0 + k < NC ? k0 + k : 0] * ld] : 0.0;
                for (int i = g; i < NC; i += KG) {
                    double s = h == 0 ? ((i == j) ? var : 0.0) : aug[i * LA + j];
                    const double* arow = aug + i * LA + NC + k0;
#pragma unroll
                    for (int k = 0; k < NH; ++k) s += (k0 + k < NC ? arow[k] : 0.0) * pc[k];
                    aug[i * LA + j] = s;
                }
            }
        }
    }
    __syncthreads();
    dbg_stamp(2);
    // Gauss-Jordan with implicit partial pivoting, the augmented matrix [K1 | A | b] in REGISTERS:
    // lane (tx, ty) owns columns {tx, tx + TXN} of rows {ty, ty + TYN, ...}.  Per step only the pivot
    // row and the pivot column pass through LDS; the next pivot is found by the owners of column
    // k+1 with an LDS atomicMax on (|value| bits, row).  No row is ever moved: perm[k] = pivot row.
    constexpr int TXN = (LA + 1) / 2, TYN = INFO_NT / TXN, RPT = (NC + TYN - 1) / TYN;
    double* rowbuf = aug;                      // LA      (LDS is free again once the registers are loaded)
    double* colbuf = aug + LA + 1;             // NC
    double* sPivVal = colbuf + NC;             // NC
    int* sInv = reinterpret_cast<int*>(sPivVal + NC);      // NC : row -> solution index
    const int gx = tid % TXN, gy = tid / TXN;
    const bool active = gy < TYN;
    const int j0 = gx, j1 = gx + TXN;
    double v[RPT][2];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int i = gy + TYN * q;
        v[q][0] = (active && i < NC) ? aug[i * LA + j0] : 0.0;
        v[q][1] = (active && i < NC && j1 < LA) ? aug[i * LA + j1] : 0.0;
        if (active && j0 == 0 && i < NC) {
            const unsigned long long key = ((unsigned long long)__double_as_longlong(fabs(v[q][0])) & ~0xFFULL) | (unsigned long long)(255 - i);
            atomicMax(&sBest[0], key);
        }
    }
    unsigned usedmask = 0u;
    __syncthreads();
    for (int k = 0; k < NC; ++k) {
        const unsigned long long best = sBest[k & 1];
        const int p = 255 - (int)(best & 0xFFULL);
        if (active) {
            if (gy == p % TYN) {                           // owner of the pivot row publishes it
                const int qp = p / TYN;
                double r0 = 0.0, r1 = 0.0;
#pragma unroll
                for (int q = 0; q < RPT; ++q) if (q == qp) { r0 = v[q][0]; r1 = v[q][1]; }
                rowbuf[j0] = r0;
                if (j1 < LA) rowbuf[j1] = r1;
                usedmask |= 1u << qp;
            }
            const int kc = k >= TXN ? 1 : 0;
            if (gx == k - kc * TXN) {                      // owner of column k publishes it
#pragma unroll
                for (int q = 0; q < RPT; ++q) { const int i = gy + TYN * q; if (i < NC) colbuf[i] = kc ? v[q][1] : v[q][0]; }
            }
        }
        if (tid == 0) { sBest[(k + 1) & 1] = 0ULL; sInv[p] = k; if ((best >> 8) == 0ULL) atomicOr(&status[b], 4); }
        __syncthreads();
        if (active) {
            const double piv = rowbuf[k];
            if (tid == 0) sPivVal[k] = piv;
            const double inv = fast_rcp(piv);
            const double r0 = rowbuf[j0], r1 = j1 < LA ? rowbuf[j1] : 0.0;
            const int kn = k + 1, knc = kn >= TXN ? 1 : 0;
            const bool own_next = kn < NC && gx == kn - knc * TXN;
            unsigned long long mykey = 0ULL;                   // this thread's best candidate: one LDS atomic per owner
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int i = gy + TYN * q;
                if (i < NC && i != p) {
                    const double f = colbuf[i] * inv;
                    v[q][0] -= f * r0;
                    v[q][1] -= f * r1;
                    if (own_next && !((usedmask >> q) & 1u)) {
                        const double nv = knc ? v[q][1] : v[q][0];
                        const unsigned long long key = ((unsigned long long)__double_as_longlong(fabs(nv)) & ~0xFFULL) | (unsigned long long)(255 - i);
                        mykey = key > mykey ? key : mykey;
                    }
                }
            }
            if (mykey) atomicMax(&sBest[kn & 1], mykey);
        }
        __syncthreads();
    }
    // solution: row i holds component k = sInv[i], scaled by its pivot; stage [M | t] back into LDS rows
    double* sol = aug + LA + 1 + 3 * NC + 8;           // NC x (NC + 1), past the small buffers
    if (active) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int i = gy + TYN * q;
            if (i < NC) {
                const int ks = sInv[i];
                const double d = fast_rcp(sPivVal[ks]);
                if (j0 >= NC) sol[ks * (NC + 1) + (j0 - NC)] = v[q][0] * d;
                if (j1 >= NC && j1 < LA) sol[ks * (NC + 1) + (j1 - NC)] = v[q][1] * d;
            }
        }
    }
    __syncthreads();
    dbg_stamp(3);
    // publish M (row-major MP x MP, zero padded) and t for the apply kernel; save Pc = P[:, clone cols]
    // (the apply kernel updates P in place and must read the PRE-update columns)
    double* Mg = Mall + (size_t)bl * mstride;
    for (int i = ty; i < MP; i += INFO_NT / 64)
        for (int j = tx; j < MP; j += 64) Mg[(size_t)i * MP + j] = (i < NC && j < NC) ? sol[i * (NC + 1) + j] : 0.0;
    for (int i = tid; i < MP; i += INFO_NT) Mg[(size_t)MP * MP + i] = i < NC ? sol[i * (NC + 1) + NC] : 0.0;
    double* Pc = Pcall + (size_t)bl * ystride;
    if (!zero_copy)
    for (int k = ty; k < MP; k += INFO_NT / 64) {
        const int gk = k < NC ? sCol[k] : 0;
        const bool real = k < ncol;
        for (int r = tx; r < n; r += 64) Pc[r + (size_t)k * ld] = real ? P[r + (size_t)gk * ld] : 0.0;
    }
    dbg_stamp(4);
    if (tid == 0) { m_out[bl] = ncol; nc_out[bl] = ncol; pc_base_out[bl] = zero_copy ? sCol[0] : -1; }
}

// ---------------------------------------------------------------------------------------------
// K8 + K10 + K11, MFMA FP64: one wave per 16-row tile ri of the state.
//   T_ri = Pc[ri,:] M                      (16 x n, v_mfma_f64_16x16x4_f64, M streamed from L2)
//   dx[ri] = Pc[ri,:] t
//   P[ri, rj] -= T_ri Pc[rj,:]^T  for every 16-column tile rj <= ri, mirrored into the upper
//   triangle (the reference's 0.5 (P + P^T), StateManager.cpp:411, without a second pass).
// T_ri goes through LDS once to turn the MFMA C/D layout into the A-operand layout.
// K4 = MP / 4 is a compile-time constant: all operand loads of a tile are issued before its MFMAs.
// grid = (ceil(nt / 4), nb), 4 waves per workgroup.
// ---------------------------------------------------------------------------------------------

// The posterior is written with STREAMING stores: nothing in this kernel re-reads it, and 240 MB of write-allocated lines evict what
// the next kernels want from L2 / MALL (the snapshot strips, the next frame's inputs).  Measured: apply 0.153 -> 0.144 ms and the
// rest of the step faster too (propagate 0.051 -> 0.046, gate 0.286 -> 0.276): 0.730 -> 0.698 ms per step.
#define APPLY_STORE(p, v) NT_STORE(p, v)      // dev_common.h; -DINGVIO_NO_NT builds the ordinary-store variant for A/B runs
#ifndef APPLY_PF
#define APPLY_PF 2      // steps the prior's tiles run ahead where the registers allow it (see PF in k_info_apply)
#endif
#define APPLY_LOADP(p) NT_LOAD(p)             // the prior's tiles (each read once) as streaming loads as well: 0.706 -> 0.693 ms per step
// YW > 0 (round 4): a second, rank-YW downdate rides on the same sweep - P - T Pc^T - Yg Yg^T with Yg [n][YW] (ld = ldp) the
// Cholesky-form gain of an in-frame GNSS update (GnssUpdate.cpp:290 right after the MSCKF update of the same frame): its columns
// join T's as A fragments and Pc's in the staged B tile.  gm_all[bl] == 0: no GNSS update for that filter.
template <int NC, int TW, int YW = 0>
__global__ __launch_bounds__(256, 2) void k_info_apply(CovView cv, int b0, int nb, int wgpf, const double* __restrict__ Mall, int mstride,
                                                       const double* __restrict__ Pcall, int ystride, const int* __restrict__ m_all,
                                                       double* __restrict__ dx_all, int* __restrict__ status,
                                                       const int* __restrict__ marg_idx, int msize, const int* __restrict__ pc_base,
                                                       const double* __restrict__ Ygall = nullptr, size_t ygstride = 0, const int* __restrict__ gm_all = nullptr,
                                                       int* __restrict__ flip_cnt = nullptr)
{
    constexpr int MP = (NC + 3) & ~3, K4 = MP / 4, JT = (MP + 15) / 16, KY = YW / 4, MPY = MP + YW;
    // sT (layout change of T, first phase) and sB (B-operand tile Pc[16 tj .. +16][0..MP), staged once per workgroup
    // per step, second phase) share their LDS
    // TW tile columns per step (round 3: 2 - half as many steps, each a dependent chain global load -> LDS -> barrier -> MFMA -> store)
    constexpr int BW = 16 * TW;
    // First phase: M itself in LDS (windows up to 11 clones: 37 KB, two workgroups per CU still fit) - the T phase reads it 8 times per
    // workgroup, and from L2 every one of its 2 x JT products was a dependent round trip of 1-3 us under load.
    constexpr bool MLDS = MP <= 68;
    constexpr int ST_DOUBLES = 4 * 16 * (MP + 2), SB_DOUBLES = 2 * MPY * BW, SV_DOUBLES = 4 * 16 * 17, SM_DOUBLES = MLDS ? MP * MP : 0;
    // TLDS: the T rows of a wave's SHORT tile row (tiR[0], a few steps of the sweep) stay in LDS and are read as A fragments where they
    // are used; only the long row's live in registers (34 fewer: room for a second set of the prior's tiles in flight).  The sweep's
    // buffers then take M's place instead of T's.
    constexpr bool TLDS = MLDS;
    constexpr int SWEEP_DOUBLES = SB_DOUBLES + SV_DOUBLES;
    constexpr int REG_A = TLDS ? ST_DOUBLES : (ST_DOUBLES > SWEEP_DOUBLES ? ST_DOUBLES : SWEEP_DOUBLES);
    constexpr int REG_B = TLDS ? (SM_DOUBLES > SWEEP_DOUBLES ? SM_DOUBLES : SWEEP_DOUBLES) : SM_DOUBLES;
    __shared__ __attribute__((aligned(16))) double sTB[REG_A + REG_B];
    double (*sT)[16][MP + 2] = reinterpret_cast<double (*)[16][MP + 2]>(sTB);
    double* sweep_base = TLDS ? sTB + REG_A : sTB;
    double (*sB)[MPY][BW] = reinterpret_cast<double (*)[MPY][BW]>(sweep_base);
    double (*sV)[16][17] = reinterpret_cast<double (*)[16][17]>(sweep_base + SB_DOUBLES);      // the stores' transposition buffers: second phase only
    double* sM = sTB + REG_A;
    // XCD-aware order: the workgroups of one filter share an L2 (they all stream the same Pc and M)
    const int wg = blockIdx.x, xcd = wg & 7, tq = wg >> 3;
    const int bl = xcd + 8 * (tq / wgpf), part = tq % wgpf;
    if (bl >= nb) return;
    const int b = b0 + bl;
    const bool upd = m_all[bl] != 0;
    const bool updY = YW > 0 && gm_all[bl] != 0;
    const int midx = marg_idx ? marg_idx[bl] : -1;            // fused StateManager::marginalize of [midx, midx+msize)
    const bool fused = midx >= 0;
    if (!upd && !updY && !fused) return;
    // The sweep is tiled in the index space of the OUTPUT (after the fused marginalisation): the rows and columns [midx, midx + msize)
    // are never formed, and - what matters - every 128-byte run of the posterior is written whole.  Tiled on the prior's indices
    // (rounds 2-3) all columns past the marginalised clone were written at a 48-byte offset, i.e. as partial lines: streaming stores
    // of rows at that offset run at 2.6 TB/s against 4.9 TB/s aligned (tools/micro/hbm_mix.hip).  The READS are the misaligned side now.
    const int n = cv.n[b], ld = cv.ldp;
    const int no = fused ? n - msize : n, nt = (no + 15) >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;      // (wave as a scalar, readfirstlane: 130 -> 141 us per launch - left a vector value)
    if (2 * part * 4 >= nt) return;                           // whole workgroup idle (uniform)
    // Fused marginalisation (round 6): the halves flip and n shrinks inside this kernel - what k_post_marg did in a dependent launch of
    // its own behind it (5 us per step whatever the batch).  Every workgroup that works on the filter (8 part < nt) has read cur / n
    // above and arrives here; the last ARRIVAL flips (the others hold their pointers in registers; an idle workgroup that starts later
    // reads the smaller n and stays idle).  The arrival is issued first thing and looked at after the first memory round trip of the
    // set-up (flip_check below): nothing of it lives into the sweep - as an epilogue it kept n, nt and b alive through the kernel,
    // which runs at the SGPR limit (106, 18 spilled): 28 spilled, +13 us; the flip moved into the solve cost the write-back 11 us as well.
    int flip_old = -2;
    if (flip_cnt && fused && tid == 0) flip_old = atomicAdd(&flip_cnt[bl], 1);
    auto flip_check = [&]() __attribute__((always_inline)) {
        if (flip_old == ((nt + 7) >> 3) - 1) { flip_cnt[bl] = 0; cv.cur[b] ^= 1; cv.n[b] = n - msize; }
    };
    const int pw = part * 4 + wave;                           // this wave owns tile rows pw and nt-1-pw: nt+1 tiles, balanced
    const bool wave_on = 2 * pw < nt;
    const int tiR[2] = { pw, nt - 1 - pw };
    const int nrows = !wave_on ? 0 : (tiR[1] > tiR[0] ? 2 : 1);
    const int tjmax = nt - 1 - part * 4;                      // the workgroup's longest row
    const double* P = cov_ptr(cv, b);
    double* dst = fused ? cov_alt_ptr(cv, b) : cov_ptr(cv, b);
    // pc_base >= 0: the clone columns are a contiguous block of the (untouched, out-of-place) prior itself; columns
    // past the window only meet zero rows of M
    const int pcb = upd ? pc_base[bl] : -1;
    const double* Pc = pcb >= 0 ? P + (size_t)pcb * ld : Pcall + (size_t)bl * ystride;
    const double* M = Mall + (size_t)bl * mstride;
    const double* tvec = M + (size_t)MP * MP;
    const double* Yg = YW > 0 ? Ygall + (size_t)bl * ygstride : nullptr;
    const int l15 = lane & 15, kq = lane >> 4;
    auto src_of = [&](int o) __attribute__((always_inline)) { return (fused && o >= midx) ? o + msize : o; };      // output index -> index in the prior

    dbg_stamp(12);
    // ---- tiles (ti, tj): all waves walk tj together; the B tile is staged in LDS once for the four waves ----------
    constexpr int STG = (MPY * BW + 255) / 256;
    double stgA[STG], stgB[STG];                               // two B tiles in flight: one needed at the end of this step, one at the end of the next
    auto stage_load = [&](int tjj, double (&stg)[STG]) {       // element e = k * BW + r  ->  Pc[src(16 tjj + r)][k]  (k >= MP: Yg[..][k - MP])
#pragma unroll
        for (int u = 0; u < STG; ++u) {
            const int e = tid + 256 * u, k = e / BW, r = e - k * BW;
            const int sr = src_of(min(16 * tjj + r, no - 1));
            double v = 0.0;
            if (e < MP * BW) { if (upd) v = Pc[sr + (size_t)k * ld]; }
            else if (YW > 0 && e < MPY * BW) { if (updY) v = Yg[sr + (size_t)(k - MP) * ld]; }
            stg[u] = v;
        }
    };
    auto stage_store = [&](int buf, const double (&stg)[STG]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < STG; ++u) { const int e = tid + 256 * u; if (e < MPY * BW) (&sB[buf][0][0])[e] = stg[u]; }
    };
    // A tile strictly below the diagonal whose rows all exist needs no per-element predicate (120 of the 136 tiles at N = 249): its
    // loads and stores are issued back to back.  (As one predicated region per element - what the general form below compiles to -
    // a tile's eight stores and four LDS reads were eight dependent exec-mask branches, each waiting for its own LDS read: 1200-1600
    // cycles per tile on a loaded CU, measured with shader-clock stamps, next to 1100 for the tile's 17 MFMAs.)
    bool neg_diag = false;                                     // StateManager.cpp:413-421, reported once at the end
    auto store_tile = [&](int ti, int tj, const double4_f& acc, const double (&pv)[4]) __attribute__((always_inline)) {
        // element (row, col), row >= col: stored through the mirrored address (col fastest, coalesced); its transpose
        // goes through LDS so that the second store runs along rows, coalesced as well
        const int col = tj * 16 + l15, row0 = ti * 16 + kq, row2 = ti * 16 + l15, col20 = tj * 16 + kq;
        double v[4], t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = pv[r] - acc[r]; sV[wave][kq + 4 * r][l15] = v[r]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = sV[wave][l15][kq + 4 * r];
        __builtin_amdgcn_wave_barrier();
        double* d1 = dst + col + (size_t)row0 * ld;
        double* d2 = dst + row2 + (size_t)col20 * ld;
#if defined(APPLY_ABL) && (APPLY_ABL & 1)      // ablation probe: no stores (one that never happens keeps the values alive)
        if (v[0] + v[1] + v[2] + v[3] + t[0] + t[1] + t[2] + t[3] == 1.2345e-300) dst[0] = v[0];
#else
        if (ti > tj && ti * 16 + 15 < no) {
#pragma unroll
            for (int r = 0; r < 4; ++r) APPLY_STORE(d1 + (size_t)(4 * r) * ld, v[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) APPLY_STORE(d2 + (size_t)(4 * r) * ld, t[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * r;
                if (row < no && col < no && row >= col) APPLY_STORE(d1 + (size_t)(4 * r) * ld, v[r]);
                neg_diag |= row == col && row < no && v[r] < 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col2 = col20 + 4 * r;
                if (row2 < no && col2 < no && row2 > col2) APPLY_STORE(d2 + (size_t)(4 * r) * ld, t[r]);
            }
        }
#endif
    };
    auto load_p = [&](int ti, int tj, double (&pv)[4]) __attribute__((always_inline)) {
#if defined(APPLY_ABL) && (APPLY_ABL & 2)      // ablation probe: the prior is not read
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[r] = 1.0 + r;
#else
        const int col = tj * 16 + l15, row0 = ti * 16 + kq;
        if (ti > tj && ti * 16 + 15 < no) {                   // mirrored (coalesced) address; rows and columns in the prior's index space
            const double* p0 = P + src_of(col);
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[r] = APPLY_LOADP(p0 + (size_t)src_of(row0 + 4 * r) * ld);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * r;
                pv[r] = (row < no && col < no && row >= col) ? APPLY_LOADP(&P[src_of(col) + (size_t)src_of(row) * ld]) : 0.0;
            }
        }
#endif
    };
    // The first B tile and the first prior tiles are requested BEFORE the T phase (round 4; -DAPPLY_LATE: after it, as before): its
    // ~170 MFMAs per wave (4.5 us) otherwise run with nothing of this workgroup in flight on the memory side (0.147 -> 0.143 ms).
    // The prior's tiles run APPLY_PF steps ahead of the MFMAs that consume them (a step of one wave is 2 x K4 MFMAs, ~1 us: one step
    // of lookahead is less than the latency of HBM under load).
    constexpr int PF = (TLDS && YW == 0 && TW == 1) ? APPLY_PF : 1;
    double pq[PF + 1][2][TW][4];
    double (&pv)[2][TW][4] = pq[0];
    auto load_step = [&](int tj0, double (&dstp)[2][TW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < TW; ++q) if (h < nrows && tj0 + q <= tiR[h]) load_p(tiR[h], tj0 + q, dstp[h][q]);
    };
    constexpr int MST = (SM_DOUBLES + 255) / 256;
    double mreg[MST > 0 ? MST : 1];
    if (MLDS && upd) {                                         // requested first: the first thing the T phase needs
#pragma unroll
        for (int u = 0; u < MST; ++u) { const int e = tid + 256 * u; mreg[u] = e < MP * MP ? M[e] : 0.0; }
    }
#ifndef APPLY_LATE
    stage_load(0, stgA);
    if (TW <= tjmax) stage_load(TW, stgB);
#pragma unroll
    for (int d = 0; d < PF; ++d) load_step(d * TW, pq[d]);
#endif
    if (MLDS && upd) {
#pragma unroll
        for (int u = 0; u < MST; ++u) { const int e = tid + 256 * u; if (e < MP * MP) sM[e] = mreg[u]; }
        lds_barrier();
    }
    flip_check();
    dbg_stamp(11);
    // ---- T rows of this wave's (up to) two tile rows, kept as A-operand fragments ------------------------------
    double tfrag[TLDS ? 1 : 2][K4];                            // TLDS: row tiR[1] only
    double yfrag[2][KY > 0 ? KY : 1];
    if (YW > 0 && updY) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h < nrows) {
                const int ra = src_of(min(tiR[h] * 16 + l15, no - 1));
#pragma unroll
                for (int k4 = 0; k4 < KY; ++k4) yfrag[h][k4] = Yg[ra + (size_t)(4 * k4 + kq) * ld];      // A[i][k] = Yg[i][k]
            }
    }
    if (upd && fused && part == 0 && wave == 0) {      // the correction of the states about to be marginalised: no tile covers them
        for (int q = lane; q < msize; q += WAVE) {
            double d = 0.0;
            for (int k = 0; k < MP; ++k) d += Pc[midx + q + (size_t)k * ld] * tvec[k];
            dx_all[(size_t)b * ld + midx + q] = d;
        }
    }
    if (upd) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = TLDS ? 1 - hh : hh;                  // TLDS: the long row first, the short row's T is what remains in sT
            if (h < nrows) {
                const int ti = tiR[h];
                const int ra = src_of(min(ti * 16 + l15, no - 1));           // clamped: rows past the end are computed but never stored
                double afrag[K4];
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) afrag[k4] = (Pc + (size_t)(4 * k4) * ld)[ra + kq * ld];      // uniform base + one lane offset
                {
                    double d = 0.0;
#pragma unroll
                    for (int k4 = 0; k4 < K4; ++k4) d += afrag[k4] * tvec[4 * k4 + kq];
                    d += __shfl_xor(d, 16, WAVE);
                    d += __shfl_xor(d, 32, WAVE);
                    if (kq == 0 && ti * 16 + l15 < no) dx_all[(size_t)b * ld + ra] = d;      // dx stays in the prior's index space
                }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
                    const int jc = min(jt * 16 + l15, MP - 1);
                    double bfrag[K4];
#pragma unroll
                    for (int k4 = 0; k4 < K4; ++k4)      // B[k][j] = M[k][j]
                        bfrag[k4] = MLDS ? sM[(4 * k4 + kq) * MP + jc] : (M + (size_t)(4 * k4) * MP)[kq * MP + jc];
#pragma unroll
                    for (int k4 = 0; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[k4], bfrag[k4], acc, 0, 0, 0);
                    if (jt * 16 + l15 < MP) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sT[wave][kq + 4 * r][jc] = acc[r];      // C/D: col = lane&15, row = (lane>>4)+4r
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (!TLDS || h == 1) {
#pragma unroll
                    for (int k4 = 0; k4 < K4; ++k4) tfrag[TLDS ? 0 : h][k4] = sT[wave][l15][4 * k4 + kq];      // A[i][k] = T[i][k]
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

#ifdef APPLY_LATE
    stage_load(0, stgA);
    if (TW <= tjmax) stage_load(TW, stgB);
#endif
    lds_barrier();                                           // every wave is done with sT
    stage_store(0, stgA);
    lds_barrier();
#ifdef APPLY_LATE
#pragma unroll
    for (int d = 0; d < PF; ++d) load_step(d * TW, pq[d]);
#endif
    // step s reads sB[s & 1]; the registers that leave for sB[(s + 1) & 1] at its end were requested a step earlier (stgB in even
    // steps, stgA in odd ones), and the other set is requested now for the step after
    auto sweep_step = [&](int tjj, int buf, double (&stg_req)[STG], const double (&stg_out)[STG]) __attribute__((always_inline)) {
        const bool more = tjj + TW <= tjmax;
        if (tjj == 4 * TW) dbg_stamp(44);
        if (tjj + 2 * TW <= tjmax) stage_load(tjj + 2 * TW, stg_req);
        load_step(tjj + PF * TW, pq[PF]);
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            const int tj = tjj + q;
            const bool do0 = nrows > 0 && tj <= tiR[0], do1 = nrows > 1 && tj <= tiR[1];
            double bfrag[K4];
            double bfy[KY > 0 ? KY : 1];
            if (upd && (do0 || do1)) {
#pragma unroll
                for (int k4 = 0; k4 < K4; ++k4) bfrag[k4] = sB[buf][4 * k4 + kq][16 * q + l15];      // B[k][j] = Pc[16 tj + j][k]
            }
            if (YW > 0 && updY && (do0 || do1)) {
#pragma unroll
                for (int k4 = 0; k4 < KY; ++k4) bfy[k4] = sB[buf][MP + 4 * k4 + kq][16 * q + l15];  // B[k][j] = Yg[16 tj + j][k]
            }
            if (tjj == 4 * TW) dbg_stamp(48);
            if (do0) {
                double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
                if (upd) {
                    if (TLDS) {
                        double a0[K4];
#pragma unroll
                        for (int k4 = 0; k4 < K4; ++k4) a0[k4] = sT[wave][l15][4 * k4 + kq];
#pragma unroll
                        for (int k4 = 0; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k4], bfrag[k4], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int k4 = 0; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tfrag[0][k4], bfrag[k4], acc, 0, 0, 0);
                    }
                }
#ifdef INGVIO_DBG_STAMPS
                if (tjj == 4 * TW) { if (acc[0] == 1.234e-300) dst[0] = 0.0; dbg_stamp(49); }
#endif
                if (YW > 0 && updY) {
#pragma unroll
                    for (int k4 = 0; k4 < KY; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yfrag[0][k4], bfy[k4], acc, 0, 0, 0);
                }
                store_tile(tiR[0], tj, acc, pv[0][q]);
                if (tjj == 4 * TW) dbg_stamp(50);
            }
            if (do1) {
                double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
                if (upd) {
#pragma unroll
                    for (int k4 = 0; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tfrag[TLDS ? 0 : 1][k4], bfrag[k4], acc, 0, 0, 0);
                }
                if (YW > 0 && updY) {
#pragma unroll
                    for (int k4 = 0; k4 < KY; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(yfrag[1][k4], bfy[k4], acc, 0, 0, 0);
                }
                store_tile(tiR[1], tj, acc, pv[1][q]);
            }
        }
        if (tjj == 4 * TW) dbg_stamp(45);
        if (more) stage_store(buf ^ 1, stg_out);
        if (tjj == 4 * TW) dbg_stamp(46);
#pragma unroll
        for (int d = 0; d < PF; ++d)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < TW; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pq[d][h][q][r] = pq[d + 1][h][q][r];
#if defined(APPLY_ABL) && (APPLY_ABL & 8)      // ablation probe (results invalid): the waves of a workgroup do not wait for each other
        __builtin_amdgcn_wave_barrier();
#else
        lds_barrier();
#endif
        if (tjj == 4 * TW) dbg_stamp(47);
    };
    for (int tjj = 0; tjj <= tjmax; tjj += 2 * TW) {
        sweep_step(tjj, 0, stgA, stgB);
        if (tjj + TW <= tjmax) sweep_step(tjj + TW, 1, stgB, stgA);
    }
    if ((upd || updY) && __any(neg_diag) && lane == 0) atomicOr(&status[b], 2);
    dbg_stamp(13);
}

// ---------------------------------------------------------------------------------------------
// K10 (+ fused K12) for FEW filters (round 5, VERDICT r04 #5: the reference is ONE real-time filter, IngvioNode.cpp:36).  k_info_apply
// gives a filter two workgroups whose waves walk 17 tiles each, one dependent step after the other: 42 us for one filter at N = 249 on
// an otherwise idle chip.  Here every 16 x 16 tile is one wave's whole job - T = Pc M in one launch (80 waves per filter), P - T Pc^T
// over the lower tiles in a second (136 waves) - with all of a wave's operands requested at once.  Same products in the same order as
// k_info_apply (the T tile is formed transposed so that its store runs along rows): bit-identical posterior and dx.  Costs L2 traffic
// (every tile re-reads its 2 x 16 x MP operands) - for batches that fill the chip k_info_apply stays.
template <int NC>
__global__ __launch_bounds__(256) void k_apply_T_flat(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                      int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx, int msize,
                                                      const int* __restrict__ pc_base, double* __restrict__ Tall, size_t tstride,
                                                      double* __restrict__ dx_all)
{
    constexpr int MP = (NC + 3) & ~3, K4 = MP / 4, JT = (MP + 15) / 16;
    const int bl = blockIdx.y, b = b0 + bl;
    if (m_all[bl] == 0) return;
    const int n = cv.n[b], ld = cv.ldp, nt = (n + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int t = blockIdx.x * 4 + wave;
    if (t >= nt * JT) return;
    const int ti = t / JT, tj = t - ti * JT;
    const double* P = cov_ptr(cv, b);
    const int pcb = pc_base[bl];
    const double* Pc = pcb >= 0 ? P + (size_t)pcb * ld : Pcall + (size_t)bl * ystride;
    const double* M = Mall + (size_t)bl * mstride;
    const double* tvec = M + (size_t)MP * MP;
    const int ra = min(16 * ti + l15, n - 1), jc = min(16 * tj + l15, MP - 1);
    double pf[K4], mf[K4];
#pragma unroll
    for (int k4 = 0; k4 < K4; ++k4) pf[k4] = (Pc + (size_t)(4 * k4) * ld)[ra + kq * ld];             // Pc[row][k]
#pragma unroll
    for (int k4 = 0; k4 < K4; ++k4) mf[k4] = (M + (size_t)(4 * k4) * MP)[kq * MP + jc];              // M[k][col]
    const int midx = marg_idx ? marg_idx[bl] : -1;
    if (tj == 0) {                                             // dx = Pc t, in the prior's index space
        double d = 0.0;
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) d += pf[k4] * tvec[4 * k4 + kq];
        d += __shfl_xor(d, 16, WAVE);
        d += __shfl_xor(d, 32, WAVE);
        const bool marg_row = midx >= 0 && ra >= midx && ra < midx + msize;
        if (kq == 0 && 16 * ti + l15 < n && !marg_row) dx_all[(size_t)b * ld + ra] = d;
    }
    if (midx >= 0 && t == 0) {      // the states about to be marginalised: summed the way k_info_apply sums them (one lane per row, k ascending)
        for (int q = lane; q < msize; q += WAVE) {
            double d = 0.0;
            for (int k = 0; k < MP; ++k) d += Pc[midx + q + (size_t)k * ld] * tvec[k];
            dx_all[(size_t)b * ld + midx + q] = d;
        }
    }
    double4_f acc = { 0.0, 0.0, 0.0, 0.0 };                    // acc[r] = T[row 16 ti + l15][col 16 tj + kq + 4 r]
#pragma unroll
    for (int k4 = 0; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(mf[k4], pf[k4], acc, 0, 0, 0);
    double* T = Tall + (size_t)bl * tstride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 16 * tj + kq + 4 * r;
        if (16 * ti + l15 < n && col < MP) T[ra + (size_t)col * ld] = acc[r];
    }
}

template <int NC>
__global__ __launch_bounds__(256) void k_apply_sym_flat(CovView cv, int b0, const double* __restrict__ Pcall, int ystride, const int* __restrict__ m_all,
                                                        const int* __restrict__ marg_idx, int msize, const int* __restrict__ pc_base,
                                                        const double* __restrict__ Tall, size_t tstride, int* __restrict__ status)
{
    constexpr int MP = (NC + 3) & ~3, K4 = MP / 4;
    __shared__ double sV[4][16][17];
    const int bl = blockIdx.y, b = b0 + bl;
    const bool upd = m_all[bl] != 0;
    const int midx = marg_idx ? marg_idx[bl] : -1;
    const bool fused = midx >= 0;
    if (!upd && !fused) return;
    const int n = cv.n[b], ld = cv.ldp, no = fused ? n - msize : n, nt = (no + 15) >> 4;      // tiles in the index space of the OUTPUT
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    int t = blockIdx.x * 4 + wave;
    if (t >= nt * (nt + 1) / 2) return;
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const double* P = cov_ptr(cv, b);
    double* dst = fused ? cov_alt_ptr(cv, b) : cov_ptr(cv, b);
    const int pcb = upd ? pc_base[bl] : -1;
    const double* Pc = pcb >= 0 ? P + (size_t)pcb * ld : Pcall + (size_t)bl * ystride;
    const double* T = Tall + (size_t)bl * tstride;
    auto src_of = [&](int o) __attribute__((always_inline)) { return (fused && o >= midx) ? o + msize : o; };
    const int col = tj * 16 + l15, row0 = ti * 16 + kq, row2 = ti * 16 + l15, col20 = tj * 16 + kq;
    const bool inner = ti > tj && ti * 16 + 15 < no;
    double pv[4];
    {
        const int sc = src_of(min(col, no - 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * r;
            // mirrored (coalesced) address, clamped: the load itself is unconditional and its value masked by a factor (a load under
            // a condition becomes an exec-masked block with its own wait - four dependent round trips)
            const double v = APPLY_LOADP(&P[sc + (size_t)src_of(min(row, no - 1)) * ld]);
            pv[r] = v * ((inner || (row < no && col < no && row >= col)) ? 1.0 : 0.0);
        }
    }
    double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
    if (upd) {
        const int ra = src_of(min(ti * 16 + l15, no - 1)), rb = src_of(min(tj * 16 + l15, no - 1));
        double tf[K4], bf[K4];
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) tf[k4] = (T + (size_t)(4 * k4) * ld)[ra + kq * ld];          // A[i][k] = T[16 ti + i][k]
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) bf[k4] = (Pc + (size_t)(4 * k4) * ld)[rb + kq * ld];         // B[k][j] = Pc[16 tj + j][k]
        __builtin_amdgcn_sched_barrier(0);                     // every operand requested before the first product (a latency path: registers are free)
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[k4], bf[k4], acc, 0, 0, 0);
    }
    double v[4], tr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = pv[r] - acc[r]; sV[wave][kq + 4 * r][l15] = v[r]; }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[r] = sV[wave][l15][kq + 4 * r];
    __builtin_amdgcn_wave_barrier();
    double* d1 = dst + col + (size_t)row0 * ld;
    double* d2 = dst + row2 + (size_t)col20 * ld;
    bool neg_diag = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * r;
        if (inner || (row < no && col < no && row >= col)) APPLY_STORE(d1 + (size_t)(4 * r) * ld, v[r]);
        neg_diag |= row == col && row < no && v[r] < 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col2 = col20 + 4 * r;
        if (inner || (row2 < no && col2 < no && row2 > col2)) APPLY_STORE(d2 + (size_t)(4 * r) * ld, tr[r]);
    }
    if (upd && __any(neg_diag) && lane == 0) atomicOr(&status[b], 2);
}

// ---------------------------------------------------------------------------------------------
// Columns of the MSCKF posterior BEFORE it is written (in-frame GNSS update): W[:, c] = (P - (Pc M) Pc^T)[:, colmap[c]],
// c < nc <= 16 - all the GNSS update reads of the covariance (GnssUpdate.cpp:148-290 runs ekfUpdate on var_order = [SE23, YOF,
// clock states, FS]: 15 columns).  One wave per 16-row tile: T_ri = Pc[ri, :] M as in k_info_apply, then one more 16-wide product
// against the gathered rows Pc[colmap[c], :].  Rows of a clone that the frame will marginalise are computed too (nobody reads them).
// ---------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void k_post_cols(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                   int ystride, const int* __restrict__ m_all, const int* __restrict__ pc_base,
                                                   const int* __restrict__ colmap_all, const int* __restrict__ nc_all, int cstride,
                                                   double* __restrict__ Wall, size_t wstride)
{
    // (Pc M) Pc[v, :]^T = Pc Z with Z = M Pc[v, :]^T (MP x 16): the workgroup forms Z once (JT tile rows of 16, K4 MFMAs each, dealt
    // to the four waves) and every wave then needs K4 MFMAs for its 16 rows instead of (JT + 1) K4 through T = Pc M
    constexpr int MP = (NC + 3) & ~3, K4 = MP / 4, JT = (MP + 15) / 16;
    __shared__ __attribute__((aligned(16))) double sZ[16 * JT][17];
    const int bl = blockIdx.y, b = b0 + bl;
    const int nc = nc_all[bl];
    if (nc == 0) return;                                         // no GNSS rows staged for this filter (uniform for the workgroup)
    const int n = cv.n[b], ld = cv.ldp, nt = (n + 15) >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int ti = blockIdx.x * 4 + wave;
    const bool upd = m_all[bl] != 0;
    const double* P = cov_ptr(cv, b);
    const int pcb = upd ? pc_base[bl] : -1;
    const double* Pc = pcb >= 0 ? P + (size_t)pcb * ld : Pcall + (size_t)bl * ystride;
    const double* M = Mall + (size_t)bl * mstride;
    const int* cm = colmap_all + (size_t)bl * cstride;
    double* W = Wall + (size_t)bl * wstride;
    const int gcol = cm[min(l15, nc - 1)];                       // state column behind W's column l15
    if (upd) {
        double bfrag[K4];                                        // B[k][j] = Pc[colmap[j]][k]
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) bfrag[k4] = Pc[gcol + (size_t)(4 * k4 + kq) * ld];
        for (int t = wave; t < JT; t += 4) {
            double4_f z = { 0.0, 0.0, 0.0, 0.0 };
            const int mr = min(16 * t + l15, MP - 1);
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) z = __builtin_amdgcn_mfma_f64_16x16x4f64(M[(size_t)mr * MP + 4 * k4 + kq], bfrag[k4], z, 0, 0, 0);      // A[i][k] = M[16 t + i][k]
#pragma unroll
            for (int r = 0; r < 4; ++r) sZ[16 * t + kq + 4 * r][l15] = z[r];
        }
    }
    lds_barrier();
    if (ti >= nt) return;
    double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
    if (upd) {
        const int ra = min(ti * 16 + l15, n - 1);
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64((Pc + (size_t)(4 * k4) * ld)[ra + kq * ld], sZ[4 * k4 + kq][l15], acc, 0, 0, 0);
    }
    if (l15 < nc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = ti * 16 + kq + 4 * r;
            if (row < n) W[row + (size_t)l15 * ld] = P[row + (size_t)gcol * ld] - acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int CMAX, bool STEREO>
static void launch_ft(const FactoredLaunch& L, hipStream_t st)
{
    if (L.stage == 0) {
        const int nb8 = (L.nb + 7) / 8 * 8;
        // INGVIO_GATE=3 selects the first-generation gate (K + 4 border rows) for comparison; mono always takes it (its K lives in
        // the 2-rows-per-observation measurement space, where Hf is not a stack of identities)
        // The product library carries ONE stereo fallback, INGVIO_GATE=4; the first-generation stereo gate and the other measured-and-
        // rejected variants are compiled only with -DINGVIO_ALT_KERNELS (build_var/alt, tests/test_gpu_alternatives.py)
#ifdef INGVIO_ALT_KERNELS
        static const bool gate3 = [] { const char* e = getenv("INGVIO_GATE"); return e && e[0] == '3'; }();
#else
        constexpr bool gate3 = false;
#endif
        static const bool gate4 = [] { const char* e = getenv("INGVIO_GATE"); return e && e[0] == '4'; }();      // one feature per wave (round 3)
        if constexpr (STEREO && CMAX <= 11) {
            if (!gate3 && !gate4) {
                LAUNCH_GATE(L, (k_feat_gate5<CMAX>), dim3(nb8 * ((L.fmax_used + 3) / 4)), dim3(WAVE), 0, st, L.cv, L.fv, L.op, L.b0, L.nb, L.fmax_used,
                                   L.gamma, L.accept);
                return;
            }
        }
        if constexpr (!STEREO && CMAX == 11) {                        // mono, 7..11 clones: four features per wave (INGVIO_GATE=3: the first-generation gate)
            static const bool gate3m = [] { const char* e = getenv("INGVIO_GATE"); return e && e[0] == '3'; }();
            if (!gate3m) {
                LAUNCH_GATE(L, (k_feat_gate5m<CMAX>), dim3(nb8 * ((L.fmax_used + 3) / 4)), dim3(WAVE), 0, st, L.cv, L.fv, L.op, L.b0, L.nb, L.fmax_used,
                                   L.gamma, L.accept);
                return;
            }
        }
        if constexpr (STEREO) {
            if (!gate3) {
                LAUNCH_GATE(L, (k_feat_gate4<CMAX>), dim3(nb8 * L.fmax_used), dim3(WAVE), 0, st, L.cv, L.fv, L.op, L.b0, L.nb, L.fmax_used,
                                   L.gamma, L.accept);
                return;
            }
        }
#ifndef INGVIO_ALT_KERNELS
        if constexpr (!STEREO)                                         // k_feat_gate3<CMAX, true> is not instantiated in the product library
#endif
        LAUNCH_GATE(L, (k_feat_gate3<CMAX, STEREO>), dim3(nb8 * ((L.fmax_used + GATE_FPW - 1) / GATE_FPW)), dim3(GATE_FPW * WAVE), 0, st,
                           L.cv, L.fv, L.op, L.b0, L.nb, L.fmax_used, L.gamma, L.accept, L.rec);
    } else {
#ifdef INGVIO_ALT_KERNELS
        if constexpr (CMAX * CMAX <= 128) {                          // window classes 6 and 11, INGVIO_GRAM=3: the third generation (gram3_kernel.h)
            static const bool gram3 = [] { const char* e = getenv("INGVIO_GRAM"); return e && e[0] == '3'; }();
            if (gram3) {
                constexpr size_t uni3 = sizeof(Gram3Batch<CMAX>) > sizeof(Gram2Out<CMAX>) ? sizeof(Gram3Batch<CMAX>) : sizeof(Gram2Out<CMAX>);
                const size_t sm3 = ((uni3 + 15) / 16) * 16 + 2 * sizeof(int) * (size_t)L.fv.fmax;
                static size_t attr_sm3 = 0;
                if (sm3 > attr_sm3) {
                    hipFuncSetAttribute((const void*)k_feat_gram3<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm3);
                    attr_sm3 = sm3;
                }
                hipLaunchKernelGGL((k_feat_gram3<CMAX, STEREO>), dim3(L.G, L.nb), dim3(GRAM_NT), sm3, st,
                                   L.fv, L.op, L.b0, L.accept, L.used, L.Apart, L.chunk_used, L.G, L.rstride);
                return;
            }
        }
#endif
        constexpr size_t uni = sizeof(Gram2Batch<CMAX>) > sizeof(Gram2Out<CMAX>) ? sizeof(Gram2Batch<CMAX>) : sizeof(Gram2Out<CMAX>);
        const size_t sm = ((uni + 15) / 16) * 16 + 2 * sizeof(int) * (size_t)L.fv.fmax;
        static size_t attr_sm = 0;
        if (sm > attr_sm) {
            hipFuncSetAttribute((const void*)k_feat_gram2<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            attr_sm = sm;
        }
        hipLaunchKernelGGL((k_feat_gram2<CMAX, STEREO>), dim3(L.G, L.nb), dim3(GRAM_NT), sm, st,
                           L.fv, L.op, L.b0, L.accept, L.used, L.Apart, L.chunk_used, L.G, L.rstride);
    }
}

// out[bl][e] = sum over the used chunks g of Apart[bl][g][e], in chunk order (the order the solve itself used to add them in);
// used_out[bl] = the chunks' counts summed.  Every load is issued whether its chunk is used or not (its value is selected away): a load under
// the condition is a dependent round trip per chunk.
__global__ __launch_bounds__(256) void k_chunk_sum(const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
                                                   double* __restrict__ out, int* __restrict__ used_out)
{
    const int bl = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    const int* cu = chunk_used + (size_t)bl * G;
    const double* A = Apart + (size_t)bl * G * rstride;
    unsigned long long mask = 0;                              // G <= 64 (a 32-bit mask aliased chunks 32.. of the variant build)
    int total = 0;
    for (int g = 0; g < G; ++g) { mask |= (cu[g] != 0 ? 1ull : 0ull) << g; total += cu[g]; }
    if (e < rstride) {
        double s = 0.0;
#pragma unroll 8
        for (int g = 0; g < G; ++g) {
            const double x = A[(size_t)g * rstride + e];
            s += ((mask >> g) & 1ull) ? x : 0.0;
        }
        out[(size_t)bl * rstride + e] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) used_out[bl] = total;
}

int factored_rec_size(int cmax)
{
    if (cmax > 16) return cmax <= bigwin_cmax() ? bigwin_rec_size() : 0;
    const int cls = cmax <= 6 ? 6 : (cmax <= 11 ? 11 : (cmax <= 12 ? 12 : 16));
    return rec_size(cls);
}

int launch_factored(const FactoredLaunch& L, hipStream_t st)
{
    if (L.fv.cmax > 16) return launch_bigwin(L, st);          // large windows: kernels_bigwin.hip
    const int cm_sel = (L.c_used > 0 && L.c_used <= L.fv.cmax) ? L.c_used : L.fv.cmax;      // window class by the frames (launch_factored.h)
    const int ncm = 6 * cm_sel;
    if (L.stage == 4) {                                       // columns of the posterior for an in-frame GNSS update (between stages 2 and 3)
        const int nt = (L.n_cap + 15) / 16;
#define POSTCOLS_DISPATCH(NC)                                                                                           \
        hipLaunchKernelGGL((k_post_cols<NC>), dim3((nt + 3) / 4, L.nb), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride, L.m_out, \
                           L.pc_base, L.gcolmap, L.gnc, L.gcstride, L.gW, L.gWstride);
        if (ncm <= 36) { POSTCOLS_DISPATCH(36) } else if (ncm <= 66) { POSTCOLS_DISPATCH(66) } else if (ncm <= 72) { POSTCOLS_DISPATCH(72) } else { POSTCOLS_DISPATCH(96) }
#undef POSTCOLS_DISPATCH
        return 0;
    }
    if (L.stage == 3) {
        const int nt = (L.n_cap + 15) / 16, wgpf = ((nt + 1) / 2 + 3) / 4, nb8 = (L.nb + 7) / 8 * 8;
        // two tile columns per step (half the steps): measured 0.194 against 0.151 ms (256 VGPRs + 76 B scratch) - selectable only
#ifdef INGVIO_ALT_KERNELS
        static const bool tw1 = [] { const char* e = getenv("INGVIO_APPLY_TW"); return !(e && e[0] == '2'); }();
#define APPLY_TW2(NC) else if (!tw1) hipLaunchKernelGGL((k_info_apply<NC, 2>), dim3(nb8 * wgpf), dim3(256), 0, st, L.cv, L.b0, L.nb, wgpf, L.T, L.mstride, L.Pc, \
                           L.ystride, L.m_out, L.dx, L.status, L.marg_idx, L.marg_size, L.pc_base);
#else
#define APPLY_TW2(NC)
#endif
        // few filters: one wave per tile, two flat launches (k_apply_T_flat / k_apply_sym_flat)
        if (!L.gY && L.Tflat && L.nb <= L.flat_nb) {
            const int JTx = ncm <= 36 ? 3 : (ncm <= 72 ? 5 : 6);
#define FLAT_DISPATCH(NC)                                                                                             \
            hipLaunchKernelGGL((k_apply_T_flat<NC>), dim3((nt * JTx + 3) / 4, L.nb), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride, \
                               L.m_out, L.marg_idx, L.marg_size, L.pc_base, L.Tflat, L.tfstride, L.dx);                 \
            hipLaunchKernelGGL((k_apply_sym_flat<NC>), dim3((nt * (nt + 1) / 2 + 3) / 4, L.nb), dim3(256), 0, st, L.cv, L.b0, L.Pc, L.ystride, \
                               L.m_out, L.marg_idx, L.marg_size, L.pc_base, L.Tflat, L.tfstride, L.status);
            if (ncm <= 36) { FLAT_DISPATCH(36) } else if (ncm <= 66) { FLAT_DISPATCH(66) } else if (ncm <= 72) { FLAT_DISPATCH(72) } else { FLAT_DISPATCH(96) }
#undef FLAT_DISPATCH
            return 0;
        }
#define APPLY_DISPATCH(NC)                                                                                            \
        if (L.gY) { hipLaunchKernelGGL((k_info_apply<NC, 1, 16>), dim3(nb8 * wgpf), dim3(256), 0, st, L.cv, L.b0, L.nb, wgpf, L.T, L.mstride, L.Pc, \
                           L.ystride, L.m_out, L.dx, L.status, L.marg_idx, L.marg_size, L.pc_base, L.gY, L.gYstride, L.gm, flip); if (flip) *L.did_flip = 1; } \
        APPLY_TW2(NC)                                                                                                 \
        else { hipLaunchKernelGGL((k_info_apply<NC, 1>), dim3(nb8 * wgpf), dim3(256), 0, st, L.cv, L.b0, L.nb, wgpf, L.T, L.mstride, L.Pc, \
                           L.ystride, L.m_out, L.dx, L.status, L.marg_idx, L.marg_size, L.pc_base, (const double*)nullptr, (size_t)0, (const int*)nullptr, flip); if (flip) *L.did_flip = 1; }
        int* const flip = (L.marg_idx && L.did_flip) ? L.flip_cnt : nullptr;
        // class 72 = a 12-clone window: an 11-pose window in sliding-window mode holds 12 clones at update time (SwMargUpdate.cpp:412-419)
        if (ncm <= 36) { APPLY_DISPATCH(36) } else if (ncm <= 66) { APPLY_DISPATCH(66) } else if (ncm <= 72) { APPLY_DISPATCH(72) } else { APPLY_DISPATCH(96) }
#undef APPLY_DISPATCH
#undef APPLY_TW2
        return 0;
    }
    if (L.stage == 2 && L.G > 1 && L.Asum) {
        // Few filters (G = 512 / B chunks per filter, up to 16): the solve used to add the G partials element by element while it built
        // its A fragments - 4 us per chunk on the critical path of ONE workgroup (B = 1: 104 us with 16 chunks against 42 with one).
        // A flat launch sums them first (same order of additions: bit-identical A), the solve reads one partial.
        hipLaunchKernelGGL(k_chunk_sum, dim3((L.rstride + 255) / 256, L.nb), dim3(256), 0, st, L.Apart, L.chunk_used, L.G, L.rstride, L.Asum, L.used_sum);
        FactoredLaunch L1 = L;
        L1.Apart = L.Asum; L1.chunk_used = L.used_sum; L1.G = 1; L1.Asum = nullptr;
        return launch_factored(L1, st);
    }
    if (L.stage == 2) {
        // default: the symmetric LDL^T solve on the matrix cores (kernels_solve.hip); INGVIO_INFO_SOLVE=gj selects the older
        // Gauss-Jordan on A Pcc + s^2 I below (kept for comparison and for the 12..16-clone class)
#ifdef INGVIO_ALT_KERNELS
        static const bool use_gj = [] { const char* e = getenv("INGVIO_INFO_SOLVE"); return e && !strcmp(e, "gj"); }();
#else
        constexpr bool use_gj = false;
#endif
        if (!use_gj && launch_info_solve(L, st) == 0) return 0;
#define INFO_DISPATCH(NC)                                                                                                   \
        {                                                                                                                   \
            const size_t sm = sizeof(double) * (size_t)NC * (2 * NC + 1) + sizeof(int) * (size_t)NC + 16;                     \
            hipFuncSetAttribute((const void*)k_info_update<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);         \
            hipLaunchKernelGGL(k_info_update<NC>, dim3(L.nb), dim3(INFO_NT), sm, st, L.cv, L.fv, L.b0, L.Apart, L.chunk_used, \
                               L.G, L.rstride, L.noise, L.T, L.mstride, L.Pc, L.ystride, L.dx, L.m_out, L.nc_out, L.status,  \
                               L.marg_idx, L.pc_base);                                                                      \
        }
        if (ncm <= 36) { INFO_DISPATCH(36) } else if (ncm <= 66) { INFO_DISPATCH(66) } else if (ncm <= 72) { INFO_DISPATCH(72) } else { INFO_DISPATCH(96) }
#undef INFO_DISPATCH
        return 0;
    }
    const int cm = cm_sel;
    // class 12 (round 6): an 11-pose window in sliding-window mode holds 12 clones at update time; on the 16-clone instantiations the
    // Gram kernel spills (132 B of scratch per lane) and both kernels loop over four slots that do not exist
    const int cls = cm <= 6 ? 6 : (cm <= 11 ? 11 : (cm <= 12 ? 12 : (cm <= 16 ? 16 : -1)));
    if (cls < 0) return -1;
#define DISPATCH(CM)                                                         \
    if (cls == CM) { if (L.stereo) launch_ft<CM, true>(L, st); else launch_ft<CM, false>(L, st); return 0; }
    DISPATCH(6)
    DISPATCH(11)
    DISPATCH(12)
    DISPATCH(16)
#undef DISPATCH
    return -1;
}

int dbg_read_factored(long long* out, int n) { return dbg_read_local(out, n); }
