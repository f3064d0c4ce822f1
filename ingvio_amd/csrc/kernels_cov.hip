// kernels_cov.hip — covariance strip kernels: K1 (fused IMU propagation), K2 (clone augmentation),
// K12 (marginalisation), addVariableIndependent.  All are HBM/latency bound: each touches the
// 15(+gnss)-wide strip or the N^2 matrix exactly once, coalesced along the column-major leading
// dimension.  gfx950 only.
#include "dev_common.h"

#define PROP_THREADS 256
#define NA_MAX 20      // 15 IMU + 4 clock biases + clock drift
#define PROP_KCH 10    // IMU steps staged in LDS per fetch

// ---------------------------------------------------------------------------------------------
// K1: StateManager::propagateStateCov (StateManager.cpp:42-119), k IMU steps fused.
// Every workgroup composes Phi_tot = Phi_k..Phi_1 and Q_tot = sum Phi_{k..s+1} Q_s Phi_{k..s+1}^T
// in LDS (15x15 work, redundantly per row tile), then applies
//     P[r, A] <- P[r, A] Phi_A^T          for rows r outside the active set A
//     P[A, A] <- Phi_A P[A, A] Phi_A^T + Q_A, symmetrised
// where A = {0..14} + the GNSS clock states (clock-bias <- clock-drift coupling, :56-86, and the
// clock process noise, :99-116).  grid = (row tiles, nb).
// ---------------------------------------------------------------------------------------------
// body of K2 for one filter, executed by a whole workgroup of NT threads (ends with the size bump by thread 0)
template <int NT>
__device__ __forceinline__ void augment_filter(CovView cv, int b, const double* __restrict__ Rp, double (*sJP)[21])
{
    const int tid = threadIdx.x;
    const int n = cv.n[b], ld = cv.ldp;
    double* P = cov_ptr(cv, b);
    double R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = Rp[i];
    for (int c = tid; c < n; c += NT) {
        double e[6], q[6], jp[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { e[j] = P[c + (size_t)j * ld]; q[j] = P[c + (size_t)(15 + j) * ld]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            jp[i] = e[i] + R[3 * i] * q[0] + R[3 * i + 1] * q[1] + R[3 * i + 2] * q[2];
            jp[3 + i] = e[3 + i] + R[3 * i] * q[3] + R[3 * i + 1] * q[4] + R[3 * i + 2] * q[5];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            P[c + (size_t)(n + i) * ld] = jp[i];
            P[(n + i) + (size_t)c * ld] = jp[i];
            if (c < 21) sJP[i][c] = jp[i];
        }
    }
    __syncthreads();
    if (tid < 36) {
        const int i = tid / 6, i2 = tid % 6;
        auto X = [&](int a, int c) {            // (J P J^T)[a][c] = JP[a][:21] . J[c][:]
            const int off = c < 3 ? 15 : 18, rr = c % 3;
            return sJP[a][c] + R[3 * rr] * sJP[a][off] + R[3 * rr + 1] * sJP[a][off + 1] + R[3 * rr + 2] * sJP[a][off + 2];
        };
        P[(n + i) + (size_t)(n + i2) * ld] = 0.5 * (X(i, i2) + X(i2, i));      // :293
    }
    __syncthreads();
    if (tid == 0) cv.n[b] = n + 6;
}

template <bool SPLIT>      // SPLIT: the composition's steps on two waves (few filters: the launch site picks it for up to 64)
__global__ __launch_bounds__(PROP_THREADS) void k_propagate(
    CovView cv, int b0, const double* __restrict__ Phi, const double* __restrict__ G,
    const double* __restrict__ dts, int k, const int* __restrict__ gnss_idx,
    double sg0, double sg1, double sg2, double sg3, int enable_gnss, double scb, double srw,
    const double* __restrict__ augR, int* __restrict__ status_clear, const double* __restrict__ snap, const int* __restrict__ n_snap)
{
    const int bl = blockIdx.y, b = b0 + bl, tid = threadIdx.x;
    if (status_clear && blockIdx.x == 0 && tid == 0) status_clear[b] = 0;
    // snap != nullptr (single-tile launches with the fused clone only): the step starts from the SNAPSHOT of the prior
    // (ingvio_frame_run(restore_prior) right after a fused frame step: half 0 still equals the snapshot outside the propagation's rows
    // and columns A, which this kernel rewrites anyway) - everything the propagation reads comes from the snapshot, everything it
    // writes goes to half 0: the restore pass (k_restore_strips, 23 us per 512 filters) and its traffic are gone (round 6)
    const int ld = cv.ldp;
    const int n = snap ? n_snap[b] : cv.n[b];
    double* P = snap ? cv.Pbase + (size_t)b * cv.pstride : cov_ptr(cv, b);
    const double* Ps = snap ? snap + (size_t)b * cv.pstride : P;
    if (snap && tid == 0) cv.cur[b] = 0;

    __shared__ double sT1[225];                                  // the fused clone's 6 x 21 scratch
    // a chunk of steps' (Phi, G~) fetched at once (one memory latency per chunk); after the composition the
    // same LDS holds the new strip, transposed for the row-wise store
    __shared__ __attribute__((aligned(16))) double sAll[NA_MAX * (PROP_THREADS + 1) > PROP_KCH * 405 ? NA_MAX * (PROP_THREADS + 1) : PROP_KCH * 405];
    double* const sStrip = sAll;
    __shared__ __attribute__((aligned(16))) double sPhiA[NA_MAX * NA_MAX], sQA[NA_MAX * NA_MAX], sX[NA_MAX * NA_MAX], sY[NA_MAX * NA_MAX];
    __shared__ int sA[NA_MAX];
    __shared__ int sNA;
    __shared__ double sDt[64];
    __shared__ double sQg[26];                                  // GNSS clock block of the composed step (+ total time), from the compose loop's idle thread

    // fused clone: P[c, 15..20] (extrinsics columns: never in the active set) of this thread's row, untouched by the propagation
    // when the row is outside the active set - requested now, used at the very end
    double qpre[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    if (augR && tid < n) {
#pragma unroll
        for (int j = 0; j < 6; ++j) qpre[j] = Ps[tid + (size_t)(15 + j) * ld];
    }
    dbg_stamp(16);
    const double* PhiB = Phi + (size_t)bl * k * 225;
    const double* GB = G + (size_t)bl * k * 180;
    const double* dtB = dts + (size_t)bl * k;
    for (int s = tid; s < k; s += PROP_THREADS) sDt[s] = dtB[s];
    // Loads that depend on nothing go out first and in this order: the clock-state indices (tiny, the strip's columns depend on
    // them), the LAST chunk of (Phi, G) (the composition starts with it), then the strip rows P[r, A] of this wave as MFMA
    // B-operand fragments: lane (kq, l15) holds, for each of its four 16-row tiles rt, P[row(rt, l15), A[4 t + kq]], t = 0..4.
    int giq[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) giq[g] = (enable_gnss && gnss_idx) ? gnss_idx[bl * 5 + g] : -1;
    constexpr int PER = (PROP_KCH * 405 + PROP_THREADS - 1) / PROP_THREADS;
    const int nchunk = (k + PROP_KCH - 1) / PROP_KCH;
    double vch[PER];
    auto chunk_load = [&](int ci) {                              // all of a chunk's loads are issued before the first LDS store
        const int s0 = ci * PROP_KCH, cnt = min(PROP_KCH, k - s0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + u * PROP_THREADS, q = e / 405, w = e - q * 405;
            vch[u] = e < cnt * 405 ? (w < 225 ? PhiB[(s0 + q) * 225 + w] : GB[(s0 + q) * 180 + (w - 225)]) : 0.0;
        }
    };
    auto chunk_store = [&](int ci) {
        const int s0 = ci * PROP_KCH, cnt = min(PROP_KCH, k - s0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + u * PROP_THREADS, w = e % 405;
            if (e < cnt * 405) sAll[e] = w < 225 ? vch[u] : vch[u] * (w < 225 + 45 ? sg0 : w < 225 + 90 ? sg1 : w < 225 + 135 ? sg2 : sg3);   // G_tmp, :92-96
        }
    };
    chunk_load(nchunk - 1);
    const int lane = tid & 63, kq = lane >> 4, l15 = lane & 15, wv = tid >> 6;
    double pvf[4][5];                                            // the strip fragments
    int naq = 15;
    double aap[2] = { 0.0, 0.0 };                                // this thread's (up to) two elements of the A x A block, tile 0 only
    int cg[5] = { 0, 0, 0, 0, 0 };                               // active-set columns 15..19: the clock states that are present, in order
    auto in_active = [&](int row) { return row < 15 || row == giq[0] || row == giq[1] || row == giq[2] || row == giq[3] || row == giq[4]; };
    {
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            if (giq[g] >= 0) {
#pragma unroll
                for (int q = 0; q < 5; ++q) cg[q] = (naq - 15 == q) ? giq[g] : cg[q];
                ++naq;
            }
        }
#pragma unroll
        for (int t4 = 0; t4 < 5; ++t4) {
            // column 4 t + kq of the active set (padding: column 0, its Phi_A entries are zero)
            int cc = 4 * t4 + kq;
            if (t4 == 3) cc = kq == 3 ? cg[0] : cc;
            if (t4 == 4) cc = kq == 0 ? cg[1] : (kq == 1 ? cg[2] : (kq == 2 ? cg[3] : cg[4]));
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const int row = blockIdx.x * PROP_THREADS + 64 * wv + 16 * rt + l15;
                pvf[rt][t4] = row < n ? Ps[row + (size_t)cc * ld] : 0.0;
            }
        }
        if (blockIdx.x == 0 && gridDim.x > 1) {                 // (a single tile takes Phi_A P_AA from its strip product, see the A x A block below)
            auto acol = [&](int i) { return i < 15 ? i : (i == 15 ? cg[0] : i == 16 ? cg[1] : i == 17 ? cg[2] : i == 18 ? cg[3] : cg[4]); };
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + u * PROP_THREADS;
                if (e < naq * naq) { const int a = e / naq, c = e - a * naq; aap[u] = Ps[acol(a) + (size_t)acol(c) * ld]; }
            }
        }
    }
    // the 5x5 clock block's recursion (sequential in the steps, a few FLOP each) rides on thread 255, beside the matrix-core
    // composition of wave 0
    // ---- composition of the k IMU steps on the matrix cores, ONE wave, no barrier inside a chunk (round 3) -----------------------
    // With Psi_s = Phi_k ... Phi_s:   Phi_tot = Psi_1,   Q_tot = sum_s dt_s (Psi_s G~_s)(Psi_s G~_s)^T   (:51 + :97 composed).
    // Going backwards in s the transposes chain through the MFMA layouts without any data movement: a 16 x 16 result in the C/D
    // layout (lane (kq, l15), register r = element (kq + 4 r, l15)) IS the B operand of the next product (k-step r), and for
    // X^T X it is the A operand as well:
    //     Psi_s^T = Phi_s^T Psi_{s+1}^T          A = Phi_s^T fragments from LDS, B = the registers of Psi_{s+1}^T        (4 MFMA)
    //     X_s     = G~_s^T Psi_s^T   (12 x 15)   A = G~_s^T fragments from LDS, B = the registers of Psi_s^T             (4 MFMA)
    //     Q      += dt_s X_s^T X_s               A = B = the registers of X_s                                            (3 MFMA)
    // (the first version ran two barrier-separated phases of 15-term LDS dot products per step: 3.7 k cycles per step, 37 k of the
    // kernel's 145 k; the steps of a chunk now take 11 MFMAs each).
    // Phi_A / Q_A zeroed and the active set written NOW, by everybody, under the loads above (round 6: this used to follow the
    // composition as zero fill - barrier - thread 0's serial bookkeeping - barrier: 5 k of the kernel's 75 k cycles); every thread has
    // computed naq and the clock columns cg[] from the indices itself
    auto acol_of = [&](int i) { return i < 15 ? i : (i == 15 ? cg[0] : i == 16 ? cg[1] : i == 17 ? cg[2] : i == 18 ? cg[3] : cg[4]); };
    for (int a = tid; a < NA_MAX * NA_MAX; a += PROP_THREADS) { sPhiA[a] = 0.0; sQA[a] = 0.0; }
    if (tid < NA_MAX) sA[tid] = tid < naq ? acol_of(tid) : 0;      // padding: loads stay unconditional, Phi_A rows/cols there are zero
    if (tid == 0) sNA = naq;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 PsiT = { 0.0, 0.0, 0.0, 0.0 }, Qacc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int r = 0; r < 4; ++r) PsiT[r] = (kq + 4 * r == l15) ? 1.0 : 0.0;
    for (int ci = nchunk - 1; ci >= 0; --ci) {
        const int s0 = ci * PROP_KCH, cnt = min(PROP_KCH, k - s0);
        if (ci != nchunk - 1) chunk_load(ci);
        chunk_store(ci);
        lds_barrier();                                          // LDS only: the strip fragments requested above stay in flight under the composition (__syncthreads would drain them: 22 k cycles of "fetch" at 512 filters)
        if (ci == nchunk - 1) dbg_stamp(21);
        // Round 6 (last day): the chunk's steps in two halves on two waves - wave 0 carries the chain through the upper half while wave 1
        // composes the lower half from the identity, Psi_L^T and Q_L = sum dt (Psi_L,s G~)(..)^T; then, on wave 0,
        //     Psi^T <- Psi_L^T Psi_U^T   and   Q <- Q_U + Psi_U Q_L Psi_U^T = Q_U + (Psi_U^T)^T (Q_L Psi_U^T)
        // (12 MFMAs; Psi_L^T and Q_L reach wave 0 as A fragments through LDS, everything else is the C/D = B = transposed-A identity of
        // the layouts above).  A chunk of ten steps was 110 dependent MFMAs on one wave - 12.4 k of the kernel's 75 k cycles, and 5 of
        // the 18 us a single filter's propagation takes; now 55 + 12.
        auto run_steps = [&](int s_hi, int s_lo, d4& Pt, d4& Qa) __attribute__((always_inline)) {
            for (int s = s_hi; s >= s_lo; --s) {
                const double* sStep = sAll + (s - s0) * 405;
                const double dt = sDt[s];
                const bool rowok = l15 < 15;
                double af[4], gf[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int kk = 4 * t4 + kq;
                    const bool ok = rowok && kk < 15;
                    af[t4] = ok ? sStep[kk + 15 * l15] : 0.0;                               // Phi_s^T [i = l15][k = kk]
                    gf[t4] = (ok && l15 < 12) ? sStep[225 + kk + 15 * l15] : 0.0;           // G~_s^T  [i = l15][k = kk]
                }
                d4 nw = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) nw = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t4], Pt[t4], nw, 0, 0, 0);
                Pt = nw;
                d4 X = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) X = __builtin_amdgcn_mfma_f64_16x16x4f64(gf[t4], Pt[t4], X, 0, 0, 0);
#pragma unroll
                for (int t4 = 0; t4 < 3; ++t4) Qa = __builtin_amdgcn_mfma_f64_16x16x4f64(X[t4], dt * X[t4], Qa, 0, 0, 0);
            }
        };
        // (a template flag, not a run-time one: with every CU busy the second wave's products are not free - 512 filters: 40 -> 55 us -
        // and the two-wave form behind a run-time condition still cost the full batch 19 us through its code generation alone)
        const bool split = SPLIT && cnt >= 4;
        const int sm = split ? s0 + cnt / 2 : s0;                // lower half [s0, sm), upper half [sm, s0 + cnt)
        double* const sPL = sX;                                   // Psi_L^T and Q_L, 16 x 16 row-major (sX / sY are free until the A x A block)
        double* const sQL = sY;
        if (!SPLIT) {
            if (tid < 64) {
                for (int s = s0 + cnt - 1; s >= s0; --s) {
                    const double* sStep = sAll + (s - s0) * 405;
                    const double dt = sDt[s];
                    const bool rowok = l15 < 15;
                    double af[4], gf[4];
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int kk = 4 * t4 + kq;
                        const bool ok = rowok && kk < 15;
                        af[t4] = ok ? sStep[kk + 15 * l15] : 0.0;                               // Phi_s^T [i = l15][k = kk]
                        gf[t4] = (ok && l15 < 12) ? sStep[225 + kk + 15 * l15] : 0.0;           // G~_s^T  [i = l15][k = kk]
                    }
                    d4 nw = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) nw = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t4], PsiT[t4], nw, 0, 0, 0);
                    PsiT = nw;
                    d4 X = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) X = __builtin_amdgcn_mfma_f64_16x16x4f64(gf[t4], PsiT[t4], X, 0, 0, 0);
#pragma unroll
                    for (int t4 = 0; t4 < 3; ++t4) Qacc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[t4], dt * X[t4], Qacc, 0, 0, 0);
                }
            }
        }
        else if (tid < 64) run_steps(s0 + cnt - 1, sm, PsiT, Qacc);
        else if (wv == 1 && split) {
            d4 Pl = { 0.0, 0.0, 0.0, 0.0 }, Ql = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int r = 0; r < 4; ++r) Pl[r] = (kq + 4 * r == l15) ? 1.0 : 0.0;
            run_steps(sm - 1, s0, Pl, Ql);
#pragma unroll
            for (int r = 0; r < 4; ++r) { sPL[(kq + 4 * r) * 16 + l15] = Pl[r]; sQL[(kq + 4 * r) * 16 + l15] = Ql[r]; }      // C/D: element (kq + 4 r, l15)
        }
        if (wv == 3 && ci == nchunk - 1) {
            // beside wave 0's first chunk: the 5 x 5 clock block's recursion over ALL steps (sequential in the steps, :56-86 and
            // :99-116 composed), one element per lane of wave 3: lane e = 5 a + c holds qg[a][c]; a step is two lane exchanges
            // (row 4 into the present rows, then the updated column 4 into the present columns) and the noise terms.
            const int ea = lane / 5, ec = lane - 5 * ea;
            auto present = [&](int q) { return (q == 0 ? giq[0] : q == 1 ? giq[1] : q == 2 ? giq[2] : q == 3 ? giq[3] : giq[4]) >= 0; };
            const bool el = lane < 25, pa = el && present(ea), pc = el && present(ec), has_fs = giq[4] >= 0;
            double v = 0.0, Tg = 0.0;
#pragma unroll 1
            for (int s = 0; s < k; ++s) {
                const double dt = sDt[s];
                if (has_fs) {
                    Tg += dt;
                    const double x = __shfl(v, 20 + (el ? ec : 0), WAVE);          // qg[4][c]
                    if (pa && ea < 4) v += dt * x;
                    const double y = __shfl(v, el ? 5 * ea + 4 : 0, WAVE);          // qg[r][4], after the row pass
                    if (pc && ec < 4) v += dt * y;
                }
                if (pa && pc) {
                    if (ea != 4 && ec != 4) v += dt * scb * scb + dt * dt * dt * srw * srw;      // :110
                    else if (ea == 4 && ec == 4) v += dt * srw * srw;                            // :112
                    else v += dt * dt * srw * srw;                                               // :114
                }
            }
            if (el) sQg[lane] = v;
            if (lane == 0) sQg[25] = Tg;
        }
        lds_barrier();                                        // the chunk's LDS may be overwritten; wave 1's half is in LDS
        if (split) {
            if (tid < 64) {
                double al[4], aq[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) { al[t4] = sPL[l15 * 16 + 4 * t4 + kq]; aq[t4] = sQL[l15 * 16 + 4 * t4 + kq]; }      // A[i = l15][k = 4 t4 + kq]
                d4 Z = { 0.0, 0.0, 0.0, 0.0 }, Pn = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[t4], PsiT[t4], Z, 0, 0, 0);            // Q_L Psi_U^T
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) Qacc = __builtin_amdgcn_mfma_f64_16x16x4f64(PsiT[t4], Z[t4], Qacc, 0, 0, 0);      // + (Psi_U^T)^T (Q_L Psi_U^T)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(al[t4], PsiT[t4], Pn, 0, 0, 0);          // Psi_L^T Psi_U^T
                PsiT = Pn;
            }
            lds_barrier();                                    // sPL / sQL may be rewritten by the next chunk
        }
    }
    dbg_stamp(17);
    // the composed step into Phi_A / Q_A: wave 0 its 15 x 15 blocks, 25 lanes of wave 1 the GNSS clock block (position of clock state g
    // in the active set: 15 + the number of present clock states before it) - disjoint entries, one barrier
    if (tid < 64) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = kq + 4 * r;
            if (l15 < 15 && c < 15) {
                sPhiA[l15 * NA_MAX + c] = PsiT[r];              // Psi^T[c][l15] = Phi_tot(l15, c)
                sQA[c * NA_MAX + l15] = Qacc[r];                // Q(c, l15)
            }
        }
    } else if (wv == 1 && lane < 25) {
        const int ga = lane / 5, gc = lane - 5 * ga;
        auto gidx = [&](int q) { return q == 0 ? giq[0] : q == 1 ? giq[1] : q == 2 ? giq[2] : q == 3 ? giq[3] : giq[4]; };
        auto locof = [&](int q) { int l = 15; for (int u = 0; u < 5; ++u) l += (u < q && gidx(u) >= 0) ? 1 : 0; return l; };
        if (gidx(ga) >= 0 && gidx(gc) >= 0) {
            const int la = locof(ga), lc = locof(gc);
            sQA[la * NA_MAX + lc] = sQg[lane];
            if (ga == gc) sPhiA[la * NA_MAX + la] = 1.0;
            if (gc == 4 && ga < 4) sPhiA[la * NA_MAX + lc] = sQg[25];      // clock bias <- clock drift over the total time (has_fs: gidx(4) >= 0 here)
        }
    }
    lds_barrier();
    dbg_stamp(18);
    const int na = sNA;
    // ---- strip rows outside A on the matrix cores:  (P[r, A] Phi_A^T)^T = Phi_A P[r, A]^T, 16 rows r per tile -------------------
    // A operand = Phi_A fragments (ten per lane, read from LDS once), B operand = the prefetched strip fragments; the result tile
    // (i = active index, j = row) lands with the rows along the 16 lanes: the lower-strip store P[r, A[i]] is coalesced, the upper
    // strip goes through LDS as before.  (First version: every thread one row, 20 outputs x 20 FMAs with Phi_A broadcast from LDS -
    // 8 waves x 10 ds_read_b128 per output column made the phase LDS-issue bound: 23 k of the kernel's 91 k cycles.)
    {
        double phf[2][5];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int t4 = 0; t4 < 5; ++t4) {
                const int i = 16 * it + l15;
                phf[it][t4] = i < NA_MAX ? sPhiA[i * NA_MAX + 4 * t4 + kq] : 0.0;          // Phi_A[i][k = 4 t + kq]
            }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int rloc = 64 * wv + 16 * rt + l15, row = blockIdx.x * PROP_THREADS + rloc;
            const bool ina = in_active(row);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                d4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int t4 = 0; t4 < 5; ++t4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(phf[it][t4], pvf[rt][t4], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * it + kq + 4 * r;
                    if (i < na) {
                        sStrip[i * (PROP_THREADS + 1) + rloc] = acc[r];
                        if (row < n && !ina) P[row + (size_t)sA[i] * ld] = acc[r];
                    }
                }
            }
        }
    }
    lds_barrier();
    // upper strip = transpose (:89): lanes run along the active index a so that each store instruction
    // covers 16-element row segments instead of 64 different columns
    {
        const int a = tid & 15, rr0 = tid >> 4;
        for (int rr = rr0; rr < PROP_THREADS; rr += PROP_THREADS / 16) {
            const int r2 = blockIdx.x * PROP_THREADS + rr;
            if (r2 < n && !in_active(r2)) {
                if (a < na) P[sA[a] + (size_t)r2 * ld] = sStrip[a * (PROP_THREADS + 1) + rr];
                if (a + 16 < na) P[sA[a + 16] + (size_t)r2 * ld] = sStrip[(a + 16) * (PROP_THREADS + 1) + rr];
            }
        }
    }
    dbg_stamp(19);
    // A x A block, tile 0 only
    if (blockIdx.x == 0) {
        if (gridDim.x > 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                      // requested at the top of the kernel (behind the strip's stores a load
                const int e = tid + u * PROP_THREADS;          // issued here would wait for their acknowledgements: one vmcnt queue)
                if (e < na * na) { const int a = e / na, c = e - a * na; sX[a * NA_MAX + c] = aap[u]; }
            }
            lds_barrier();
            for (int e = tid; e < na * na; e += PROP_THREADS) {
                const int a = e / na, c = e % na;
                double acc = 0.0;
                for (int l = 0; l < na; ++l) acc += sPhiA[a * NA_MAX + l] * sX[l * NA_MAX + c];
                sY[a * NA_MAX + c] = acc;
            }
            lds_barrier();
        }
        // Single tile (round 6): Y = Phi_A P[A, A] is already there - the strip product above ran for EVERY row of the tile, the rows of
        // the active set included (computed, not stored): sStrip[i][A_c] = sum_k Phi_A[i][k] P[A_c][A_k] = (Phi_A P_AA)[i][c], P symmetric
        // bit for bit.  The block's loads, its staging and the first of the two dot-product passes (two barriers) are gone.
        const bool ytile = gridDim.x == 1;
        for (int e = tid; e < na * na; e += PROP_THREADS) {
            const int a = e / na, c = e % na;
            double acc = 0.0;
            for (int l = 0; l < na; ++l) acc += (ytile ? sStrip[a * (PROP_THREADS + 1) + sA[l]] : sY[a * NA_MAX + l]) * sPhiA[c * NA_MAX + l];
            sX[a * NA_MAX + c] = acc + sQA[a * NA_MAX + c];
        }
        lds_barrier();
        for (int e = tid; e < na * na; e += PROP_THREADS) {
            const int a = e / na, c = e % na;
            P[sA[a] + (size_t)sA[c] * ld] = 0.5 * (sX[a * NA_MAX + c] + sX[c * NA_MAX + a]);   // :118
        }
    }
    dbg_stamp(20);
    // fused K2 (single-tile launches only: this workgroup owns the whole filter): the clone's rows/columns are built
    // from the just-propagated P[0:21, :]
    if (augR) {
        // StateManager::augmentSlidingWindowPose (StateManager.cpp:279-293) from what this workgroup still holds: the new strip
        // (sStrip: P[r, A] of the rows outside A), the new A x A block (sX, before symmetrisation) and the prefetched extrinsics
        // columns - re-reading the rows it has just stored cost a store -> load round trip (43 k of the kernel's 145 k cycles)
        lds_barrier();
        double (*sJP)[21] = reinterpret_cast<double (*)[21]>(sT1);          // 6 x 21 <= 225
        double R[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = augR[bl * 9 + i];
        const int c = tid;
        if (c < n) {
            int ac = c < 15 ? c : -1;
            for (int q = 15; q < na; ++q) if (sA[q] == c) ac = q;
            double e[6], qv[6], jp[6];
            if (ac < 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { e[j] = sStrip[j * (PROP_THREADS + 1) + tid]; qv[j] = qpre[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    e[j] = 0.5 * (sX[ac * NA_MAX + j] + sX[j * NA_MAX + ac]);              // the symmetrised A x A block (:118)
                    qv[j] = sStrip[ac * (PROP_THREADS + 1) + 15 + j];                      // P[15 + j, A] of row 15 + j (outside A)
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                jp[i] = e[i] + R[3 * i] * qv[0] + R[3 * i + 1] * qv[1] + R[3 * i + 2] * qv[2];
                jp[3 + i] = e[3 + i] + R[3 * i] * qv[3] + R[3 * i + 1] * qv[4] + R[3 * i + 2] * qv[5];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                P[c + (size_t)(n + i) * ld] = jp[i];
                P[(n + i) + (size_t)c * ld] = jp[i];
                if (c < 21) sJP[i][c] = jp[i];
            }
        }
        lds_barrier();
        if (tid < 36) {
            const int i = tid / 6, i2 = tid % 6;
            auto X = [&](int a, int cc) {            // (J P J^T)[a][cc] = JP[a][:21] . J[cc][:]
                const int off = cc < 3 ? 15 : 18, rr = cc % 3;
                return sJP[a][cc] + R[3 * rr] * sJP[a][off] + R[3 * rr + 1] * sJP[a][off + 1] + R[3 * rr + 2] * sJP[a][off + 2];
            };
            P[(n + i) + (size_t)(n + i2) * ld] = 0.5 * (X(i, i2) + X(i2, i));      // :293
        }
        lds_barrier();
        if (tid == 0) cv.n[b] = n + 6;
    }
    dbg_stamp(22);
}

// ---------------------------------------------------------------------------------------------
// K2: StateManager::augmentSlidingWindowPose covariance part (StateManager.cpp:279-293).
// One workgroup per filter; the 6 new rows/cols are J*P[0:21,:], read through the symmetric
// counterpart P[c, 0:21] so consecutive lanes read consecutive addresses.
// ---------------------------------------------------------------------------------------------
struct Mat9Arg { double m[9]; };
__global__ __launch_bounds__(256) void k_augment(CovView cv, int b0, const double* __restrict__ Rs)
{
    __shared__ double sJP[6][21];
    augment_filter<256>(cv, b0 + blockIdx.x, Rs + blockIdx.x * 9, sJP);
}
// ONE filter: the rotation rides as a kernel argument (no 72-byte host-to-device copy in front of the kernel)
__global__ __launch_bounds__(256) void k_augment_imm(CovView cv, int b, Mat9Arg R)
{
    __shared__ double sJP[6][21];
    __shared__ double sR[9];
    if (threadIdx.x < 9) sR[threadIdx.x] = R.m[threadIdx.x];
    __syncthreads();
    augment_filter<256>(cv, b, sR, sJP);
}

// ---------------------------------------------------------------------------------------------
// K12: StateManager::marginalize covariance part (StateManager.cpp:163-177), out of place into
// the filter's other ping-pong buffer; k_post then flips `cur` and shrinks n.  grid = (col tiles, nb)
// ---------------------------------------------------------------------------------------------
#define MARG_COLS 8
// idxs == nullptr: ONE filter, its index rides as a kernel argument (idx_imm) - a single real-time filter then marginalises without
// the 7 us host-to-device copy of four bytes in front of the kernel
__global__ __launch_bounds__(256) void k_marginalize(CovView cv, int b0, const int* __restrict__ idxs, int size, int idx_imm)
{
    const int bl = blockIdx.y, b = b0 + bl, tid = threadIdx.x;
    const int idx = idxs ? idxs[bl] : idx_imm;
    if (idx < 0) return;
    const int n = cv.n[b], ld = cv.ldp, nn = n - size;
    const double* src = cov_ptr(cv, b);
    double* dst = cov_alt_ptr(cv, b);
    for (int jj = 0; jj < MARG_COLS; ++jj) {
        const int j = blockIdx.x * MARG_COLS + jj;
        if (j >= nn) break;
        const int sj = j < idx ? j : j + size;
        for (int i = tid; i < nn; i += 256) {
            const int si = i < idx ? i : i + size;
            NT_STORE(&dst[i + (size_t)j * ld], NT_LOAD(&src[si + (size_t)sj * ld]));
        }
    }
}

__global__ void k_post_marg(CovView cv, int b0, int nb, const int* __restrict__ idxs, int size, int idx_imm)
{
    const int bl = blockIdx.x * blockDim.x + threadIdx.x;
    if (bl >= nb) return;
    if ((idxs ? idxs[bl] : idx_imm) < 0) return;
    const int b = b0 + bl;
    cv.cur[b] ^= 1;
    cv.n[b] -= size;
}

// StateManager::addVariableIndependent (StateManager.cpp:194-214). One workgroup per filter.
__global__ __launch_bounds__(256) void k_append(CovView cv, int b0, int size, const double* __restrict__ blk)
{
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int n = cv.n[b], ld = cv.ldp;
    double* P = cov_ptr(cv, b);
    for (int e = tid; e < (n + size) * size; e += 256) {
        const int c = e / size, j = e % size;
        P[c + (size_t)(n + j) * ld] = 0.0;
        P[(n + j) + (size_t)c * ld] = 0.0;
    }
    __syncthreads();
    for (int e = tid; e < size * size; e += 256) {
        const int i = e % size, j = e / size;
        P[(n + i) + (size_t)(n + j) * ld] = blk[(size_t)bl * size * size + e];
    }
    __syncthreads();
    if (tid == 0) cv.n[b] = n + size;
}

// device-to-device snapshot of n / cur (the covariances are copied with hipMemcpyAsync)
__global__ void k_copy_ints(int* dst, const int* src, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i];
}

// Small host-to-device uploads as a KERNEL that reads the pinned staging slab over the bus (hipHostMalloc memory is mapped into the
// device's address space) instead of hipMemcpyAsync: the copy engine's completion reaches the compute queue 10-17 us after the copy
// itself ended (rocprofv3: 11 us copy, the first kernel behind it 17 us later) - for a single real-time filter a third of what the
// kernels in front of the Kalman solve take.  Words of 4 bytes (every upload of this library is a multiple), 16 where both sides allow.
__global__ __launch_bounds__(256) void k_upload_words(unsigned* __restrict__ dst, const unsigned* __restrict__ src, size_t nwords, int vec4)
{
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if (vec4) {
        const size_t n4 = nwords >> 2;
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (size_t i = i0; i < n4; i += stride) d4[i] = s4[i];
        for (size_t i = (n4 << 2) + i0; i < nwords; i += stride) dst[i] = src[i];
    } else {
        for (size_t i = i0; i < nwords; i += stride) dst[i] = src[i];
    }
}

// snapshot / restore of every filter's live covariance (benchmark hygiene: each timed step starts
// from the same prior).  grid = (col tiles, B).
__global__ __launch_bounds__(256) void k_snapshot(CovView cv, double* __restrict__ snap, int* __restrict__ n_snap)
{
    const int b = blockIdx.y, tid = threadIdx.x, n = cv.n[b], ld = cv.ldp;
    const double* src = cov_ptr(cv, b);
    double* dst = snap + (size_t)b * cv.pstride;
    for (int jj = 0; jj < MARG_COLS; ++jj) {
        const int j = blockIdx.x * MARG_COLS + jj;
        if (j >= n) break;
        for (int i = tid; i < n; i += 256) dst[i + (size_t)j * ld] = src[i + (size_t)j * ld];
    }
    if (blockIdx.x == 0 && tid == 0) n_snap[b] = n;
}
__global__ __launch_bounds__(256) void k_restore(CovView cv, int b0, const double* __restrict__ snap, const int* __restrict__ n_snap)
{
    const int b = b0 + blockIdx.y, tid = threadIdx.x, n = n_snap[b], ld = cv.ldp;
    const double* src = snap + (size_t)b * cv.pstride;
    double* dst = cv.Pbase + (size_t)b * cv.pstride;          // half 0
    for (int jj = 0; jj < MARG_COLS; ++jj) {
        const int j = blockIdx.x * MARG_COLS + jj;
        if (j >= n) break;
        for (int i = tid; i < n; i += 256) NT_STORE(&dst[i + (size_t)j * ld], NT_LOAD(&src[i + (size_t)j * ld]));      // a plain copy: streaming both ways
    }
}
// Partial restore after a fused frame step (propagate + clone + out-of-place update/marginalise): half 0 still holds
// the snapshot except for the rows/columns of the propagation's active set A = {0..14} + clock states (the six clone
// rows/cols lie beyond n_snap).  grid = (ceil(n_cap/256), B), thread = row/column index.
__global__ __launch_bounds__(256) void k_restore_strips(CovView cv, int b0, const double* __restrict__ snap, const int* __restrict__ n_snap,
                                                        const int* __restrict__ gnss_idx)
{
    const int b = b0 + blockIdx.y, n = n_snap[b], ld = cv.ldp;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) { cv.cur[b] = 0; cv.n[b] = n; }           // nothing in this kernel reads cur / n
    if (r >= n) return;
    const double* src = snap + (size_t)b * cv.pstride;
    double* dst = cv.Pbase + (size_t)b * cv.pstride;          // half 0
    int A[NA_MAX], na = 15;
#pragma unroll
    for (int a = 0; a < 15; ++a) A[a] = a;
#pragma unroll
    for (int g = 0; g < 5; ++g) { const int gi = gnss_idx ? gnss_idx[b * 5 + g] : -1; A[15 + g] = gi >= 0 ? gi : 0; if (gi >= 0) na = 16 + g; }
    double v[NA_MAX], w[NA_MAX];                          // both strips in flight before the first store (the kernel is pure latency)
#pragma unroll
    for (int a = 0; a < NA_MAX; ++a) v[a] = src[r + (size_t)A[a] * ld];
#pragma unroll
    for (int a = 0; a < NA_MAX; ++a) w[a] = src[A[a] + (size_t)r * ld];
#pragma unroll
    for (int a = 0; a < NA_MAX; ++a) if (a < na) dst[r + (size_t)A[a] * ld] = v[a];
#pragma unroll
    for (int a = 0; a < NA_MAX; ++a) if (a < na) dst[A[a] + (size_t)r * ld] = w[a];
}
__global__ void k_post_restore(CovView cv, int b0, int nb, const int* __restrict__ n_snap)
{
    const int bl = blockIdx.x * blockDim.x + threadIdx.x, b = b0 + bl;
    if (bl < nb) { cv.cur[b] = 0; cv.n[b] = n_snap[b]; }
}

// ---------------------------------------------------------------------------------------------
#include "launch_ekf.h"
void launch_propagate(CovView cv, int b0, int nb, int n_cap, const double* Phi, const double* G, const double* dt, int k,
                      const int* gnss_idx, const double sigma[4], int enable_gnss, double scb, double srw, hipStream_t st,
                      const double* augR, int* status_clear, const double* snap, const int* n_snap)
{
    const int tiles = (n_cap + PROP_THREADS - 1) / PROP_THREADS;
    const bool fuse = augR && tiles == 1;
if (nb <= 64) {     hipLaunchKernelGGL(k_propagate<true>, dim3(tiles, nb), dim3(PROP_THREADS), 0, st, cv, b0, Phi, G, dt, k, gnss_idx,
                       sigma[0], sigma[1], sigma[2], sigma[3], enable_gnss, scb, srw, fuse ? augR : nullptr, status_clear,
                       fuse ? snap : nullptr, n_snap); }
    else {     hipLaunchKernelGGL(k_propagate<false>, dim3(tiles, nb), dim3(PROP_THREADS), 0, st, cv, b0, Phi, G, dt, k, gnss_idx,
                       sigma[0], sigma[1], sigma[2], sigma[3], enable_gnss, scb, srw, fuse ? augR : nullptr, status_clear,
                       fuse ? snap : nullptr, n_snap); }
    if (augR && !fuse) hipLaunchKernelGGL(k_augment, dim3(nb), dim3(256), 0, st, cv, b0, augR);
}
bool propagate_can_restore(int n_cap) { return (n_cap + PROP_THREADS - 1) / PROP_THREADS == 1; }
void launch_augment(CovView cv, int b0, int nb, const double* R, hipStream_t st)
{
    hipLaunchKernelGGL(k_augment, dim3(nb), dim3(256), 0, st, cv, b0, R);
}
void launch_augment_one(CovView cv, int b, const double* R_host, hipStream_t st)
{
    Mat9Arg R;
    for (int i = 0; i < 9; ++i) R.m[i] = R_host[i];
    hipLaunchKernelGGL(k_augment_imm, dim3(1), dim3(256), 0, st, cv, b, R);
}
void launch_marginalize(CovView cv, int b0, int nb, int n_cap, const int* idx, int size, hipStream_t st, int idx_imm)
{
    hipLaunchKernelGGL(k_marginalize, dim3((n_cap + MARG_COLS - 1) / MARG_COLS, nb), dim3(256), 0, st, cv, b0, idx, size, idx_imm);
    hipLaunchKernelGGL(k_post_marg, dim3((nb + 255) / 256), dim3(256), 0, st, cv, b0, nb, idx, size, idx_imm);
}
void launch_post_marg(CovView cv, int b0, int nb, const int* idx, int size, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_marg, dim3((nb + 255) / 256), dim3(256), 0, st, cv, b0, nb, idx, size, -1);
}
void launch_append(CovView cv, int b0, int nb, int size, const double* blk, hipStream_t st)
{
    hipLaunchKernelGGL(k_append, dim3(nb), dim3(256), 0, st, cv, b0, size, blk);
}
void launch_upload_words(void* dst, const void* src_pinned, size_t bytes, hipStream_t st)
{
    const size_t nwords = bytes >> 2;
    const int vec4 = (((uintptr_t)dst | (uintptr_t)src_pinned) & 15) == 0 ? 1 : 0;
    const size_t items = vec4 ? (nwords >> 2) + 3 : nwords;
    int blocks = (int)((items + 255) / 256);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_upload_words, dim3(blocks), dim3(256), 0, st, (unsigned*)dst, (const unsigned*)src_pinned, nwords, vec4);
}
void launch_copy_ints(int* dst, const int* src, int count, hipStream_t st)
{
    hipLaunchKernelGGL(k_copy_ints, dim3((count + 255) / 256), dim3(256), 0, st, dst, src, count);
}
void launch_snapshot(CovView cv, int n_cap, double* snap, int* n_snap, hipStream_t st)
{
    hipLaunchKernelGGL(k_snapshot, dim3((n_cap + MARG_COLS - 1) / MARG_COLS, cv.B), dim3(256), 0, st, cv, snap, n_snap);
}
void launch_restore_strips(CovView cv, int b0, int nb, int n_cap, const double* snap, const int* n_snap, const int* gnss_idx, hipStream_t st)
{
    hipLaunchKernelGGL(k_restore_strips, dim3((n_cap + 255) / 256, nb), dim3(256), 0, st, cv, b0, snap, n_snap, gnss_idx);
}
void launch_restore(CovView cv, int b0, int nb, int n_cap, const double* snap, const int* n_snap, hipStream_t st)
{
    hipLaunchKernelGGL(k_restore, dim3((n_cap + MARG_COLS - 1) / MARG_COLS, nb), dim3(256), 0, st, cv, b0, snap, n_snap);
    hipLaunchKernelGGL(k_post_restore, dim3((nb + 255) / 256), dim3(256), 0, st, cv, b0, nb, n_snap);
}

int dbg_read_cov(long long* out, int n) { return dbg_read_local(out, n); }
