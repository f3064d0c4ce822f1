// gram3_kernel.h — K4 + K6 + K7 in Gram form, a third generation for the window classes of up to 11 clones (round 6).  MEASURED AND
// REJECTED: 114-117 us per 512 filters against k_feat_gram2's 106-108; compiled only with -DINGVIO_ALT_KERNELS (build_var/alt), selected
// with INGVIO_GRAM=3, kept correct by tests/test_gpu_alternatives.py.
//
// k_feat_gram2 walks a chunk in batches of 8 features: all four waves run the rank-3 GEMM Y^T [B | hs] of batch i (P3a, 2.5 k cycles),
// then waves 0-1 build the operand rows of batch i + 1 (P2, 9.2 k) beside the sparse sums of batch i on waves 2-3 (P3b, 3.5 k) - half of
// the workgroup idle for 5.7 k of a batch's 11.7 k cycles (shader-clock stamps, 512 filters x 150 features x 11 clones).  Decoupling the
// wave pairs needs the operand panels of batch i + 1 written while those of batch i are read: a second pair of panels, which the
// 80 KB of a workgroup (two per CU) do not have.  With Ns = L D L^T the rank-3 term is
//     B^T Ns^-1 B = Z^T Z,     Z = D^-1/2 L^-1 B       (3 x 6C per feature; the hs column rides as D^-1/2 L^-1 hs)
// i.e. ONE panel instead of the pair (B, Y = Ns^-1 B): double-buffered it takes the LDS the pair took, and P2 gets cheaper (the forward
// substitution replaces two 3 x 3 products and halves the panel stores: 9.2 k -> 6.3 k cycles, 3.1 k of issue).  Per batch: waves 0-1
// P2 of batch i + 1, waves 2-3 P3b of batch i, the tiles of the rank-3 product dealt to all four waves, ONE workgroup barrier; the two
// roles are separate loops so that neither carries the other's registers (181 VGPRs, no scratch).
// What the stamps said: with ALL tiles on waves 2-3 their two SIMDs carry 14 k cycles per batch (the matrix pipes are per SIMD, and
// the two workgroups of a CU put their waves 2-3 on the same two SIMDs) - 144 us; with the tiles dealt 5 / 5 / 3 / 2 every SIMD issues for
// ~10 k of a batch's 11.7 k cycles - the kernel is issue-bound (per batch and workgroup: P2 2 x 3.1 k, the 96 MFMAs 6.1 k, P3b
// 2 x 3.9 k) and the idle time of k_feat_gram2's schedule is not what limits it.  What would: P3b multiplies ten of its eleven
// (slot, anchor) lanes by zero (the key selects the one anchor a feature has); a form that adds a feature's slot terms into the one
// (slot, anchor) accumulator needs the accumulators in LDS (33 KB) or the features ordered by anchor.
#pragma once

template <int CMAX>
struct Gram3Batch {
    using Cfg = Gram2Cfg<CMAX>;
    double Zm[2][Cfg::KR][Cfg::LDW];                              // the operand panel, double-buffered
    double sp[2][GRAM_NB][CMAX][Cfg::SPW];                        // per (feature, slot) sparse scratch, double-buffered
};

template <int CMAX, bool STEREO>
__global__ __launch_bounds__(GRAM_NT, 2) void k_feat_gram3(
    FrameView fv, MsckfOpts op, int b0, const int* __restrict__ accept_in, int* __restrict__ used_out,
    double* __restrict__ Apart, int* __restrict__ chunk_used, int G, int rstride)
{
    using Cfg = Gram2Cfg<CMAX>;
    constexpr int NC = Cfg::NC, TJ = Cfg::TJ, LDW = Cfg::LDW, NTILE = Cfg::NTILE, TPW = Cfg::TPW, KR = Cfg::KR;
    constexpr int NUP = Cfg::NUP, TI = Cfg::TI, RPO = STEREO ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    static_assert(CMAX * CMAX <= 128, "the (slot, anchor) pair lanes sit on waves 2-3");
    Gram3Batch<CMAX>& sb = *reinterpret_cast<Gram3Batch<CMAX>*>(smem_raw);
    Gram2Out<CMAX>& so = *reinterpret_cast<Gram2Out<CMAX>*>(smem_raw);             // epilogue view of the same LDS
    constexpr size_t UNI = sizeof(Gram3Batch<CMAX>) > sizeof(Gram2Out<CMAX>) ? sizeof(Gram3Batch<CMAX>) : sizeof(Gram2Out<CMAX>);
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
    for (int e = tid; e < 2 * KR * LDW; e += GRAM_NT) (&sb.Zm[0][0][0])[e] = 0.0;
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
    // sparse accumulators of lane (c, a)
    const int ptid = tid - 128;
    const int pc = ptid >= 0 ? ptid / CMAX : 0, pa = ptid >= 0 ? ptid - pc * CMAX : 0;
    const bool pairlane = ptid >= 0 && ptid < CMAX * CMAX;
    const int kq = lane >> 4, l15 = lane & 15;

    // The per-observation quantities (N_o = G_o^T G_o, h_o = G_o^T r_o) are recomputed here from the frame inputs (one projection
    // per (feature, slot) lane) instead of travelling through a 1.5 KB per-feature record written by the gate kernel: the record
    // cost 113 MiB of HBM writes + 113 MiB of reads per launch of the 512-filter batch, the recomputation ~25 VALU per feature.
    // Lane (f, c) = (tid >> 4, tid & 15) of the first 16 GRAM_NB threads; its raw inputs for the NEXT batch are fetched into
    // registers while the matrix cores run the current one.
    double in_uv[4], in_pf[3];
    unsigned long long in_mask = 0ULL;
    int in_anchor = 0;
    const int ft = tid;                                        // (feature, slot) lane of the operand-row phase (waves 0-1)
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
    // called by waves 0-1 (128 threads): operand rows into Zm[buf], sparse scratch into sp[buf]
    auto do_p2 = [&](int qb, int buf) {
        constexpr bool rows = true, sparse = true;
        const int nbf = min(GRAM_NB, q1 - qb);
        dbg_stamp(34);
        if (nbf < GRAM_NB) {                                   // short last batch: clear the unused stacked rows of this buffer
            for (int e = tid; e < (KR - 3 * nbf) * LDW; e += 128) (&sb.Zm[buf][3 * nbf][0])[e] = 0.0;
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
                // Z = D^-1/2 L^-1 B with Ns = L D L^T (3 x 3, SPD for a used feature): B^T Ns^-1 B = Z^T Z - ONE operand panel
                // (k_feat_gram2 stages B and Y = Ns^-1 B), which is what lets the panel be double-buffered in the same LDS
                const double s0 = Ns[0] > 0.0 ? gram_rsqrt(Ns[0]) : 0.0, r0 = s0 * s0;          // s_k = d_k^-1/2 (0: a pivot that is not positive -
                const double l10 = Ns[1] * r0, l20 = Ns[2] * r0;                              // the feature then contributes nothing through that row)
                const double d1 = Ns[4] - l10 * Ns[1], s1 = d1 > 0.0 ? gram_rsqrt(d1) : 0.0, r1 = s1 * s1;
                const double t21 = Ns[5] - l20 * Ns[1], l21 = t21 * r1;
                const double d2 = (Ns[8] - l20 * Ns[2]) - l21 * t21, s2 = d2 > 0.0 ? gram_rsqrt(d2) : 0.0;
                double (*Z)[LDW] = sb.Zm[buf];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    {
                        const double y0 = Bt[q], y1 = Bt[3 + q] - l10 * y0, y2 = (Bt[6 + q] - l20 * y0) - l21 * y1;
                        Z[3 * f + 0][6 * c + q] = s0 * y0; Z[3 * f + 1][6 * c + q] = s1 * y1; Z[3 * f + 2][6 * c + q] = s2 * y2;
                    }
                    {
                        const double y0 = Bp[q], y1 = Bp[3 + q] - l10 * y0, y2 = (Bp[6 + q] - l20 * y0) - l21 * y1;
                        Z[3 * f + 0][6 * c + 3 + q] = s0 * y0; Z[3 * f + 1][6 * c + 3 + q] = s1 * y1; Z[3 * f + 2][6 * c + 3 + q] = s2 * y2;
                    }
                }
                if (c == 0) {                                       // extra column: D^-1/2 L^-1 hs
                    const double y0 = hs[0], y1 = hs[1] - l10 * y0, y2 = (hs[2] - l20 * y0) - l21 * y1;
                    Z[3 * f + 0][NC] = s0 * y0; Z[3 * f + 1][NC] = s1 * y1; Z[3 * f + 2][NC] = s2 * y2;
                }
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
                sp[33] = obs ? (double)a : -1.0;                       // key: the anchor slot this contribution belongs to
              }
            }
        }
    };
    // MFMA accumulators.  The matrix pipes are per SIMD: ALL four waves take tiles of the rank-3 product (upper tiles t, row-major over
    // ti <= tj) - waves 0-1 a third of them each beside their P2, waves 2-3 share the last third beside their P3b (per batch and wave,
    // issue cycles: P2 3.1 k, P3b 3.9 k, a tile 0.38 k; with all tiles on waves 2-3 their two SIMDs carried 14 k per batch, the
    // other two 6 k)
    constexpr int N01 = (NUP + 2) / 3, N2 = (NUP - 2 * N01 + 1) / 2, N3 = NUP - 2 * N01 - N2;
    const int tcnt = wave < 2 ? N01 : (wave == 2 ? N2 : N3), tbeg = wave < 2 ? wave * N01 : (wave == 2 ? 2 * N01 : 2 * N01 + N2);
    int tiA[N01], tjA[N01];
#pragma unroll
    for (int u = 0; u < N01; ++u) {
        int t = u < tcnt ? tbeg + u : NUP, ti = 0;
        while (ti < TI - 1 && t >= TJ - ti) { t -= TJ - ti; ++ti; }
        tiA[u] = ti; tjA[u] = ti + t;                          // t >= NUP gives tj >= TJ: never launched
    }
    double4_f acc[N01];
#pragma unroll
    for (int u = 0; u < N01; ++u) acc[u] = double4_f{ 0.0, 0.0, 0.0, 0.0 };
    auto do_p3a = [&](int qb, int buf) {                      // this wave's tiles of the rank-3 part of batch qb, Z^T Z over its stacked rows
        const int nbf = min(GRAM_NB, q1 - qb);
        const int nst = (3 * nbf + 3) >> 2;
#pragma unroll
        for (int st = 0; st < KR / 4; ++st) {                // fully unrolled: the fragment reads of the later steps are
            if (st < nst) {                                  // issued while the earlier MFMAs run
#pragma unroll
                for (int u = 0; u < N01; ++u) {
                    if (u < tcnt) {
                        const int ti = tiA[u], tj = tjA[u];
                        const double af = sb.Zm[buf][4 * st + kq][16 * ti + l15];      // A[i][k] = Z[k][i]
                        const double bf = sb.Zm[buf][4 * st + kq][16 * tj + l15];      // B[k][j] = Z[k][j]
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[u], 0, 0, 0);
                    }
                }
            }
        }
    };
    auto store_acc = [&]() {                                  // epilogue, first half: the accumulator tiles into the epilogue view of the LDS
#pragma unroll
        for (int u = 0; u < N01; ++u) {
            if (u < tcnt) {
                const int ti = tiA[u], tj = tjA[u];
                const int jc = 16 * tj + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ti + kq + 4 * r;      // C/D: col = lane&15, row = (lane>>4)+4r
                    if (i < NC && jc <= NC) so.A2[i][jc] = acc[u][r];
                }
            }
        }
    };
    // Pipeline: ONE barrier per batch.  Waves 0-1 build batch i + 1 (operand rows into the other panel, sparse scratch into the other
    // scratch buffer) while waves 2-3 run the matrix cores and the sparse sums on batch i.  The two roles are separate LOOPS (the same
    // number of barriers each), so that neither carries the other's registers: as one loop with a branch inside, the 16 accumulator
    // tiles and the 33 sparse sums of waves 2-3 were live through P2 of waves 0-1 - 308 B of scratch per lane.
    if (wave < 2) {
        if (q0 < q1) { do_p2(q0, 0); fetch(q0 + GRAM_NB); }
        lds_barrier();
        int it = 0;
        for (int qb = q0; qb < q1; qb += GRAM_NB, ++it) {
            if (it == 3) dbg_stamp(36);
            if (qb + GRAM_NB < q1) { do_p2(qb + GRAM_NB, (it + 1) & 1); fetch(qb + 2 * GRAM_NB); }
            if (it == 3) dbg_stamp(37);
            do_p3a(qb, it & 1);
            if (it == 3) dbg_stamp(41);
            lds_barrier();
            if (it == 3) dbg_stamp(42);
        }
        store_acc();
    } else {
        double sS1[9], sNX[9], sS3[9], s4[3], s5[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) { sS1[i] = 0.0; sNX[i] = 0.0; sS3[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { s4[i] = 0.0; s5[i] = 0.0; }
        auto do_p3b = [&](int qb, int buf) {
            const int nbf = min(GRAM_NB, q1 - qb);
            // ---- P3b: sparse part, lane (c, a): branch-free, one level of LDS reads (the key says whose anchor it is) ----
            if (pairlane) {
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
        lds_barrier();
        int it = 0;
        for (int qb = q0; qb < q1; qb += GRAM_NB, ++it) {
            do_p3a(qb, it & 1);
            do_p3b(qb, it & 1);
            lds_barrier();
        }
        store_acc();
        if (pairlane) {
            double* S = so.S[pc][pa];
#pragma unroll
            for (int i = 0; i < 9; ++i) { S[i] = sS1[i]; S[9 + i] = sNX[i]; S[18 + i] = sS3[i]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { S[27 + i] = s4[i]; S[30 + i] = s5[i]; }
        }
    }
    dbg_stamp(39);
    // ---- epilogue: assemble [A | b] of the chunk ---------------------------------------------------
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
