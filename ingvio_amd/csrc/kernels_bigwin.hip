// kernels_bigwin.hip — the factored MSCKF path (K3..K11) for LARGE sliding windows, 17..36 clones (the reference's
// shipped configs use 21..35, config/*/ingvio_{stereo,mono}.yaml: max_sliding_window_poses).  Same mathematics as
// kernels_factored.hip (see its header and gate_kernel.h); what no longer fits LDS / registers at 6C = 216 columns
// lives in L2-resident global workspaces:
//   k_feat_gate3_big     the shared gate body, one wave per SIMD (16x16 tiles in VGPRs + AGPRs)
//   k_feat_gram_big      rank-3 part on the matrix cores (105 upper tiles over 8 waves); the block-sparse sums
//                        per (observing slot, anchor slot) accumulate in a global array, slot c always by the same
//                        threads (no atomics, deterministic)
//   k_info_update_big    [A Pcc + s^2 I | A | b] in global memory: K1 by MFMA, Gauss-Jordan with implicit partial
//                        pivoting, one workgroup of 1024 threads per filter
//   k_info_apply_big     T = Pc M kept in LDS, K = 216 streamed in MFMA steps of 4
// Correct and parallel, not tuned: the batched config-2 benchmark never takes this path.  gfx950 only.
#include <algorithm>
#include "launch_factored.h"
#include "launch_chol.h"
#include "block64.h"
#include <stdlib.h>
#include <string.h>
#include "gate_kernel.h"

#define BIG_CMAX 36
#define BIG_NC (6 * BIG_CMAX)

// Window classes of the gate: its LDS (the packed triangle of K: 3C (3C + 1) / 2 doubles) and its tile count are set by the class,
// so a 30-clone window runs the 32 class (48 KB per wave, 3 waves per CU, 28 lower tiles) instead of the 36 class (60 KB, 2
// waves, 36 tiles).  The record keeps the 36-clone stride that k_feat_gram_big reads.
template <bool STEREO, int CM>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_feat_gate3_big(
    CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used, double* __restrict__ gamma_out,
    int* __restrict__ accept_out, double* __restrict__ rec_out)
{
    gate3_body<CM, STEREO, 1, true, REC_HDR + REC_OBS * BIG_CMAX>(cv, fv, op, b0, nb, fmax_used, gamma_out, accept_out, rec_out);
}

// Stereo, round 3: the gate in difference coordinates of the observations with TWO waves per feature (gate_kernel.h, gate4_big_body).
template <int CM>
__global__ __launch_bounds__(2 * WAVE) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_feat_gate4_big(CovView cv, FrameView fv, MsckfOpts op, int b0, int nb, int fmax_used,
                                                             double* __restrict__ gamma_out, int* __restrict__ accept_out, double* __restrict__ rec_out)
{
    gate4_big_body<CM, REC_HDR + REC_OBS * BIG_CMAX>(cv, fv, op, b0, nb, fmax_used, gamma_out, accept_out, rec_out);
}

template <bool STEREO, int CM>
static void launch_gate_big(const FactoredLaunch& L, hipStream_t st)
{
    const int nb8 = (L.nb + 7) / 8 * 8;
#ifdef INGVIO_ALT_KERNELS
    static const bool gate3 = [] { const char* e = getenv("INGVIO_GATE"); return e && e[0] == '3'; }();      // first generation, for comparison
#else
    constexpr bool gate3 = false;                                     // product library: k_feat_gate3_big serves mono only
#endif
    if constexpr (STEREO) {
        if (!gate3) {
            const size_t sm = sizeof(Gate4BigShared<CM>);
            static bool attr_set = false;
            if (!attr_set) { hipFuncSetAttribute((const void*)k_feat_gate4_big<CM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_set = true; }
            LAUNCH_GATE(L, (k_feat_gate4_big<CM>), dim3(nb8 * L.fmax_used), dim3(2 * WAVE), sm, st, L.cv, L.fv, L.op, L.b0, L.nb, L.fmax_used,
                               L.gamma, L.accept, L.rec);
            return;
        }
    }
#ifndef INGVIO_ALT_KERNELS
    if constexpr (!STEREO)
#endif
    LAUNCH_GATE(L, (k_feat_gate3_big<STEREO, CM>), dim3(nb8 * L.fmax_used), dim3(WAVE), 0, st,
                       L.cv, L.fv, L.op, L.b0, L.nb, L.fmax_used, L.gamma, L.accept, L.rec);
}

// ---------------------------------------------------------------------------------------------
// K4 + K6/K7 (see k_feat_gram2): grid = (G chunks, nb), 8 waves, batches of 4 features.
// ---------------------------------------------------------------------------------------------
#define GB_NB 4
#define GB_NT 512
#define GB_SW 34                                           // per (slot, anchor): S1(9) NXs(9) s4(3) S3(9) s5(3) + pad
struct GBCfg {
    static constexpr int CMAX = BIG_CMAX, NC = BIG_NC;
    static constexpr int TI = (NC + 15) / 16, TJ = (NC + 1 + 15) / 16, LDW = 16 * TJ;
    static constexpr int NUP = TI * TJ - TI * (TI - 1) / 2;
    static constexpr int NW = GB_NT / WAVE, TPW = (NUP + NW - 1) / NW;
    static constexpr int REC = REC_HDR + REC_OBS * CMAX;
    static constexpr int KR = 3 * GB_NB;
    static constexpr int PRE = (GB_NB * REC + GB_NT - 1) / GB_NT;
};
struct GBBatch {
    double rec[GB_NB][GBCfg::REC];
    double Bm[GBCfg::KR][GBCfg::LDW];
    double Ym[GBCfg::KR][GBCfg::LDW];
    double sp[GB_NB][GBCfg::CMAX][GB_SW];
};

__global__ __launch_bounds__(GB_NT, 1) void k_feat_gram_big(
    FrameView fv, MsckfOpts op, int b0, const int* __restrict__ accept_in, int* __restrict__ used_out,
    const double* __restrict__ rec_in, double* __restrict__ Apart, int* __restrict__ chunk_used, int G, int rstride,
    double* __restrict__ Sg_all)
{
    using Cfg = GBCfg;
    constexpr int CMAX = Cfg::CMAX, NC = Cfg::NC, TI = Cfg::TI, TJ = Cfg::TJ, LDW = Cfg::LDW, NUP = Cfg::NUP, NW = Cfg::NW;
    constexpr int TPW = Cfg::TPW, REC = Cfg::REC, KR = Cfg::KR, PRE = Cfg::PRE;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GBBatch& sb = *reinterpret_cast<GBBatch*>(smem_raw);
    int* sUse = reinterpret_cast<int*>(smem_raw + ((sizeof(GBBatch) + 15) / 16) * 16);
    int* sList = sUse + fv.fmax;
    __shared__ int sNu;
    __shared__ unsigned long long sAmask;
    // wave as a SCALAR (round 5): the tile coordinates (tiA, tjA) of a wave's 14 accumulators are wave-uniform; as vector values their
    // 28 LDS offsets were spilled and every MFMA of the batch loop waited for a scratch reload (s_waitcnt vmcnt(0), i.e. also for the
    // record prefetch issued just before): 18.5 k of a batch's 31 k cycles for 2.7 k cycles of matrix work
    const int bl = blockIdx.y, b = b0 + bl, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = fv.n_feat[b], C = fv.n_clones[b], ncol = 6 * C;
    double* Sg = Sg_all + ((size_t)bl * G + g) * (size_t)CMAX * CMAX * GB_SW;

    dbg_stamp(48);
    for (int j = tid; j < F; j += GB_NT) {                      // RemoveLostUpdate.cpp:357-359
        int use = accept_in[(size_t)b * fv.fmax + j];
        if (use && op.max_accept > 0) {
            int rank = 0;
            for (int q = 0; q < j; ++q) rank += accept_in[(size_t)b * fv.fmax + q];
            if (rank >= op.max_accept) use = 0;
        }
        sUse[j] = use;
        if (g == 0) used_out[(size_t)b * fv.fmax + j] = use;
    }
    for (int e = tid; e < KR * LDW; e += GB_NT) { (&sb.Bm[0][0])[e] = 0.0; (&sb.Ym[0][0])[e] = 0.0; }
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
        // The anchors of THIS chunk's features (round 5): the sparse sums Sg[slot][anchor] are touched for those anchor columns only.
        // A single real-time filter loses a handful of tracks per frame - one feature per chunk - and used to zero all C x C pairs here
        // (198 KB of stores at 27 clones) and to walk all of them again in the epilogue: 25 k + 53 k of the kernel's 120 k cycles.
        __builtin_amdgcn_wave_barrier();                        // this wave's own sList entries
        const int per0 = (cnt + G - 1) / G, qa = g * per0, qb = min(cnt, qa + per0);
        unsigned long long m = 0ULL;
        for (int q = qa + lane; q < qb; q += WAVE) m |= 1ULL << (int)rec_in[((size_t)b * fv.fmax + sList[q]) * REC + 1];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off, WAVE);
        if (lane == 0) sAmask = m;
    }
    __syncthreads();
    const int nu = sNu, per = (nu + G - 1) / G;
    const int q0 = g * per, q1 = min(nu, q0 + per);
    const unsigned long long amask = sAmask;
    // zeroed by the thread that later adds into the same addresses (slot c, values part + 8 j: the read-modify-write of the batch loop):
    // a thread sees its own stores in order, so no workgroup barrier has to wait for them here; every other reader of Sg sits behind
    // the __syncthreads at the end of the batch loop
    if (tid < CMAX * 8) {
        const int c = tid >> 3, part = tid & 7;
        if (c < C) {
            for (unsigned long long rest = amask; rest; rest &= rest - 1ULL) {
                const int a2 = __ffsll((long long)rest) - 1;
                if (a2 >= C) break;
                double* S = Sg + ((size_t)c * CMAX + a2) * GB_SW;
#pragma unroll
                for (int j = 0; j < 5; ++j) { const int v = part + 8 * j; if (v < GB_SW) S[v] = 0.0; }
            }
        }
    }

    int tiA[TPW], tjA[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        int t = wave + NW * u, ti = 0;
        while (ti < TI - 1 && t >= TJ - ti) { t -= TJ - ti; ++ti; }
        tiA[u] = ti; tjA[u] = ti + t;
    }
    double4_f acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) acc[u] = double4_f{ 0.0, 0.0, 0.0, 0.0 };
    const int kq = lane >> 4, l15 = lane & 15;

    double pre[PRE];
    // The next batch's records.  UNCONDITIONAL loads from clamped indices (round 5): as `cond ? rec_in[..] : 0.0` every one of the five
    // loads sat in its own exec-masked block behind an s_waitcnt vmcnt(0) - five dependent memory trips per batch, and the MFMA loop
    // behind them (its accumulators' scratch reloads wait on the same counter): 18.5 k of a batch's 31 k cycles.  Entries past the
    // batch's features are never read (the staging store and the operand phase are bounded by nbf).
    auto fetch = [&](int qb) {
        if (nu == 0) return;                                    // uniform: nothing listed, nothing to address
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = tid + u * GB_NT, f = min(e / REC, GB_NB - 1), w = e - (e / REC) * REC;
            const int qi = min(qb + f, nu - 1);
            pre[u] = rec_in[((size_t)b * fv.fmax + sList[qi]) * REC + w];
        }
    };
    fetch(q0);
    dbg_stamp(49);
    for (int qb = q0; qb < q1; qb += GB_NB) {
        const int nbf = min(GB_NB, q1 - qb);
#pragma unroll
        for (int u = 0; u < PRE; ++u) { const int e = tid + u * GB_NT; if (e < nbf * REC) (&sb.rec[0][0])[e] = pre[u]; }
        if (nbf < GB_NB) {
            for (int e = tid; e < (KR - 3 * nbf) * LDW; e += GB_NT) { (&sb.Bm[3 * nbf][0])[e] = 0.0; (&sb.Ym[3 * nbf][0])[e] = 0.0; }
        }
        lds_barrier();
        if (qb == q0 + 2 * GB_NB) dbg_stamp(53);
        // operand rows B, Y = Ns^-1 B and the sparse scratch, lane = (feature, window slot)
        if (tid < nbf * WAVE) {
            const int f = tid >> 6, c = tid & 63;
            if (c < C) {
                const double* rc = sb.rec[f];
                const int a = (int)rc[1];
                const double px = rc[2], py = rc[3], pz = rc[4];
                const unsigned long long mask = (unsigned long long)rc[5];
                const bool obs = (mask >> c) & 1ULL;
                const int o = __popcll(mask & ((1ULL << c) - 1ULL));
                const double* ro = rc + REC_HDR + REC_OBS * (obs ? o : 0);
                const double* Nsi = rc + 6;
                const double* hs = rc + 15;
                const double* Nsa = rc + 18;
                double Bt[9], Bp[9], NX[9];
                const double cn = (obs && ro[1] != 0.0) ? 1.0 : 0.0, pl = (obs && ro[2] != 0.0) ? 1.0 : 0.0;
                mulX(ro + 3, px, py, pz, NX);
#pragma unroll
                for (int i = 0; i < 9; ++i) { Bt[i] = cn * NX[i]; Bp[i] = -pl * ro[3 + i]; }
                if (c == a) {
                    double T[9];
                    mulX(Nsa, px, py, pz, T);
#pragma unroll
                    for (int i = 0; i < 9; ++i) Bt[i] = -T[i];
                }
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
                    for (int k = 0; k < 3; ++k) sb.Bm[3 * f + k][NC] = hs[k];
                }
                double* sp = sb.sp[f][c];
                double S1[9];
                mulXt(NX, px, py, pz, S1);
#pragma unroll
                for (int i = 0; i < 9; ++i) { sp[i] = cn * S1[i]; sp[9 + i] = cn * pl * NX[i]; sp[21 + i] = pl * ro[3 + i]; }
                sp[18] = cn * (pz * ro[13] - py * ro[14]);
                sp[19] = cn * (px * ro[14] - pz * ro[12]);
                sp[20] = cn * (py * ro[12] - px * ro[13]);
#pragma unroll
                for (int i = 0; i < 3; ++i) sp[30 + i] = pl * ro[12 + i];
                sp[33] = obs ? (double)a : -1.0;
            }
        }
        lds_barrier();
        if (qb == q0 + 2 * GB_NB) dbg_stamp(54);
        fetch(qb + GB_NB);
        const int nst = (3 * nbf + 3) >> 2;
#pragma unroll
        for (int st = 0; st < KR / 4; ++st) {
            if (st < nst) {
                // (all 28 operand fragments of a step read first and the products after them: 144 against 131 us per launch - the
                // fragments' 56 registers push the accumulators into scratch; in groups of 2 or 4 tiles: 146; padding the panels' row
                // stride against bank conflicts: no change)
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    if (wave + NW * u < NUP) {
                        const double af = sb.Ym[4 * st + kq][16 * tiA[u] + l15];
                        const double bf = sb.Bm[4 * st + kq][16 * tjA[u] + l15];
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[u], 0, 0, 0);
                    }
                }
            }
        }
        if (qb == q0 + 2 * GB_NB) dbg_stamp(55);
        // sparse sums: slot c is always handled by the same 8 threads (part = which of the 33 values), features in order
        // The batch's contributions are first merged by anchor (later features into the earlier one with the same key, in feature
        // order), then ALL read-modify-writes of the batch are issued together: one memory trip per batch instead of a dependent
        // load -> add -> store chain per feature (which was 25 k of the 28 k cycles of a batch).
        if (tid < CMAX * 8) {
            const int c = tid >> 3, part = tid & 7;
            if (c < C) {
                int key[GB_NB];
                double val[GB_NB][5];
#pragma unroll
                for (int f = 0; f < GB_NB; ++f) {
                    const double* sp = sb.sp[f][c];
                    key[f] = f < nbf ? (int)sp[33] : -1;
#pragma unroll
                    for (int j = 0; j < 5; ++j) { const int v = part + 8 * j; val[f][j] = (key[f] >= 0 && v < 33) ? sp[v] : 0.0; }
                }
#pragma unroll
                for (int f = 1; f < GB_NB; ++f) {
                    bool merged = false;
#pragma unroll
                    for (int f2 = 0; f2 < f; ++f2) {
                        if (!merged && key[f] >= 0 && key[f2] == key[f]) {
#pragma unroll
                            for (int j = 0; j < 5; ++j) val[f2][j] += val[f][j];
                            merged = true;
                        }
                    }
                    if (merged) key[f] = -1;
                }
                double old[GB_NB][5];
#pragma unroll
                for (int f = 0; f < GB_NB; ++f) {
                    const double* S = Sg + ((size_t)c * CMAX + (key[f] >= 0 ? key[f] : 0)) * GB_SW;
#pragma unroll
                    for (int j = 0; j < 5; ++j) { const int v = part + 8 * j; old[f][j] = (key[f] >= 0 && v < 33) ? S[v] : 0.0; }
                }
#pragma unroll
                for (int f = 0; f < GB_NB; ++f) {
                    if (key[f] >= 0) {
                        double* S = Sg + ((size_t)c * CMAX + key[f]) * GB_SW;
#pragma unroll
                        for (int j = 0; j < 5; ++j) { const int v = part + 8 * j; if (v < 33) S[v] = old[f][j] + val[f][j]; }
                    }
                }
            }
        }
        lds_barrier();
    }

    __syncthreads();                                           // the other threads' global sums (Sg) are read below
    dbg_stamp(50);
    // ---- epilogue: [A | b] of the chunk = sparse part - rank-3 part, assembled in global memory ---------------------
    double* out = Apart + ((size_t)bl * G + g) * rstride;      // [ncol][ncol+1] row-major, b in the last column
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        if (wave + NW * u < NUP) {
            const int ti = tiA[u], tj = tjA[u];
            const int jc = 16 * tj + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r;
                if (i < ncol) {
                    if (jc < ncol) {
                        out[(size_t)i * (ncol + 1) + jc] = -acc[u][r];
                        if (ti != tj) out[(size_t)jc * (ncol + 1) + i] = -acc[u][r];
                    } else if (jc == NC) {
                        out[(size_t)i * (ncol + 1) + ncol] = -acc[u][r];
                    }
                }
            }
        }
    }
    __syncthreads();
    dbg_stamp(51);
    // diagonal blocks (c, c): the sum over ALL anchors a.  One thread per (slot, output element): its 2 C loads are independent
    // (one or two memory trips), consecutive threads read consecutive elements of a pair's sums.  One thread per slot walking the
    // anchors was a 60-load dependent chain and 60 % of this kernel's time for a single filter.
    for (int q = tid; q < C * 42; q += GB_NT) {
        const int c = q / 42, v = q - 42 * c;
        int is, ia = -1, row, col;
        double ss, sa = 0.0;
        if (v < 36) {
            const int m6 = v / 6, k6 = v - 6 * m6;
            row = m6; col = 6 * c + k6;
            if (m6 < 3 && k6 < 3) { is = 3 * m6 + k6; ss = 1.0; ia = is; sa = 1.0; }
            else if (m6 < 3) { is = 9 + 3 * (k6 - 3) + m6; ss = -1.0; }                    // (theta, p) = -NXs^T
            else if (k6 < 3) { is = 9 + 3 * (m6 - 3) + k6; ss = -1.0; }                    // (p, theta) = -NXs
            else { is = 21 + 3 * (m6 - 3) + (k6 - 3); ss = 1.0; }
        } else {
            const int i = v - 36;
            row = i; col = ncol;
            if (i < 3) { is = 18 + i; ss = 1.0; ia = 18 + i; sa = -1.0; }
            else { is = 30 + (i - 3); ss = -1.0; }
        }
        double sum = 0.0;
        const int iaq = ia >= 0 ? ia : 0;
        const bool c_anchor = (amask >> c) & 1ULL;                                          // Sg[.][c] exists only then
        for (int a0 = 0; a0 < C; a0 += 12) {                                               // up to 24 independent loads per pass, added in order
            double x[12], y[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int a = min(a0 + u, C - 1);
                x[u] = ((amask >> a) & 1ULL) ? Sg[((size_t)c * CMAX + a) * GB_SW + is] : 0.0;      // obs at slot c, anchor a
                y[u] = c_anchor ? Sg[((size_t)a * CMAX + c) * GB_SW + iaq] : 0.0;                   // obs at slot a, anchor c
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) if (a0 + u < C) sum += ss * x[u] + sa * y[u];
        }
        out[(size_t)(6 * c + row) * (ncol + 1) + col] += sum;
    }
    // off-diagonal blocks (c, c2): one (slot, anchor) pair each way
    for (int q = tid; q < C * C; q += GB_NT) {
        const int c = q / C, c2 = q - c * C;
        if (c == c2) continue;
        const bool hasS = (amask >> c2) & 1ULL, hasSt = (amask >> c) & 1ULL;      // obs at slot c with anchor c2 / at slot c2 with anchor c
        if (!hasS && !hasSt) continue;                                      // neither exists: the block keeps its rank-3 part
        double blk[36];                                                     // the rank-3 part already in `out`: all loads in flight together
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int k = 0; k < 6; ++k) blk[6 * m + k] = (m >= 3 && k >= 3) ? 0.0 : out[(size_t)(6 * c + m) * (ncol + 1) + 6 * c2 + k];
        const double* S = Sg + ((size_t)c * CMAX + c2) * GB_SW;             // obs at slot c, anchor c2
        const double* St = Sg + ((size_t)c2 * CMAX + c) * GB_SW;            // obs at slot c2, anchor c
        double s1[9], st1[9], s2[9], st2[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            s1[i] = hasS ? S[i] : 0.0; s2[i] = hasS ? S[9 + i] : 0.0;
            st1[i] = hasSt ? St[i] : 0.0; st2[i] = hasSt ? St[9 + i] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                blk[6 * m + k] += -s1[3 * m + k] - st1[3 * m + k];
                blk[6 * (3 + m) + k] += s2[3 * m + k];
                blk[6 * m + 3 + k] += st2[3 * k + m];
            }
#pragma unroll
        for (int m = 0; m < 6; ++m) {
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (m < 3 || k < 3) out[(size_t)(6 * c + m) * (ncol + 1) + 6 * c2 + k] = blk[6 * m + k];      // the (p, p) quarter gets nothing here
        }
    }
    if (tid == 0) chunk_used[bl * G + g] = max(0, q1 - q0);
    dbg_stamp(52);
}

#ifdef INGVIO_ALT_KERNELS      // the Gauss-Jordan solve of round 1 (INGVIO_BIG_SOLVE=gj), variant builds only
// ---------------------------------------------------------------------------------------------
// K8/K9/K11 (see k_info_update): [K1 | A | b], K1 = A Pcc + s^2 I, in the global workspace Wk (NC x LA row-major).
// One workgroup of 1024 threads per filter.
// ---------------------------------------------------------------------------------------------
#define IB_NT 1024
__global__ __launch_bounds__(IB_NT, 1) void k_info_update_big(
    CovView cv, FrameView fv, int b0, const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
    const double* __restrict__ noise_all, double* __restrict__ Mall, int mstride, double* __restrict__ Pcall, int ystride,
    double* __restrict__ dx_all, int* __restrict__ m_out, int* __restrict__ nc_out, int* __restrict__ status,
    const int* __restrict__ marg_idx, int* __restrict__ pc_base_out, double* __restrict__ Wk_all)
{
    constexpr int NC = BIG_NC, LA = 2 * NC + 1, MP = NC;
    __shared__ int sCol[NC];
    __shared__ int sInv[NC];
    __shared__ int sUsed[NC];
    __shared__ double sPivVal[NC];
    __shared__ double Pan[NC][9];                      // the current panel's 8 columns (+1 pad)
    __shared__ double Fm[8][NC];                       // multipliers of the panel's pivots
    __shared__ double Rw[8][LA];                       // the panel's pivot rows, right of the panel
    __shared__ unsigned long long sBest[2];
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = fv.n_clones[b], ncol = 6 * C, n = cv.n[b], ld = cv.ldp;
    double* dx = dx_all + (size_t)b * ld;
    double* Wk = Wk_all + (size_t)bl * NC * LA;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        for (int r = tid; r < n; r += IB_NT) dx[r] = 0.0;
        if (tid == 0) { m_out[bl] = 0; nc_out[bl] = ncol; pc_base_out[bl] = -1; }
        return;
    }
    const double* P = cov_ptr(cv, b);
    const double var = noise_all[bl];
    dbg_stamp(48);
    for (int c = tid; c < NC; c += IB_NT) {
        const int cc = c < ncol ? c : 0;
        sCol[c] = fv.clone_idx[(size_t)b * fv.cmax + cc / 6] + cc % 6;
        sUsed[c] = 0;
    }
    if (tid < 2) sBest[tid] = 0ULL;
    // [A | b] from the chunk partials
    for (int e = tid; e < NC * (NC + 1); e += IB_NT) {
        const int i = e / (NC + 1), j = e - i * (NC + 1);
        double s = 0.0;
        if (i < ncol && (j < ncol || j == NC)) {
            const size_t src = (size_t)i * (ncol + 1) + (j == NC ? ncol : j);
            for (int g0 = 0; g0 < G; g0 += 8) {                               // eight partial loads in flight
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int g = g0 + u;
                    t[u] = (g < G && chunk_used[bl * G + g]) ? Apart[((size_t)bl * G + g) * rstride + src] : 0.0;
                }
                s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
        }
        Wk[(size_t)i * LA + NC + j] = s;
    }
    __syncthreads();
    dbg_stamp(49);
    const bool fused = marg_idx && marg_idx[bl] >= 0;
    bool contig_t = true;
    for (int c = tid; c < ncol; c += IB_NT) contig_t = contig_t && (sCol[c] == sCol[0] + c);
    const int contig = __syncthreads_and(contig_t);
    const bool zero_copy = fused && contig && sCol[0] + MP <= ld;
    // K1 = A Pcc + s^2 I on the matrix cores: wave w owns tiles t = w, w + 16, ...  Computed as K1^T = Pcc A (both
    // symmetric) so that both operand loads run along the fast index (A-operand: 16 consecutive clone columns of P,
    // B-operand: 16 consecutive entries of a row of A); only the tile store is strided.
    {
        constexpr int TT = (NC + 15) / 16, NWV = IB_NT / WAVE;
        const int kq = lane >> 4, l15 = lane & 15;
        for (int t = wave; t < TT * TT; t += NWV) {
            const int ti = t / TT, tj = t - ti * TT;
            const int ia = min(16 * ti + l15, NC - 1), jb = min(16 * tj + l15, NC - 1);
            const int gia = sCol[ia];
            double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
            constexpr int UF = 9;                                             // operand loads of 9 steps in flight
            static_assert((NC / 4) % UF == 0, "unroll factor");
            for (int s0 = 0; s0 < NC / 4; s0 += UF) {
                double af[UF], bf[UF];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const int k = 4 * (s0 + u) + kq;
                    af[u] = P[gia + (size_t)sCol[k] * ld];                    // Pcc[i][k]
                    bf[u] = Wk[(size_t)k * LA + NC + jb];                     // A[k][j]
                }
#pragma unroll
                for (int u = 0; u < UF; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
            }
            const int jc = 16 * tj + l15;                                     // acc[r] = K1^T[16 ti + kq + 4r][jc] = K1[jc][..]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r;
                if (i < NC && jc < NC) Wk[(size_t)jc * LA + i] = acc[r] + (i == jc ? var : 0.0);
            }
        }
    }
    __syncthreads();
    dbg_stamp(50);
    // Blocked Gauss-Jordan with implicit partial pivoting, panels of PB pivots.  The panel's columns live in LDS only
    // (nothing left of the current pivot is ever read again); the columns right of the panel are read and written ONCE
    // per panel:  W[i][j] -= sum_k F[k][i] R_k[j], with F[k][i] the multipliers of pivot k (0 at its own pivot row) and
    // R_k the pivot row as it stands when it is used (corrected for the earlier pivots of the panel).
    constexpr int PB = 8;
    static_assert(NC % PB == 0, "panel width");
    for (int k0 = 0; k0 < NC; k0 += PB) {
        for (int e = tid; e < NC * PB; e += IB_NT) { const int i = e / PB, jj = e - i * PB; Pan[i][jj] = Wk[(size_t)i * LA + k0 + jj]; }
        __syncthreads();
        for (int kk = 0; kk < PB; ++kk) {
            const int k = k0 + kk;
            if (tid < 256) {                                                  // NC <= 256 candidates, 4 waves
                unsigned long long key = 0ULL;
                if (tid < NC && !sUsed[tid])
                    key = ((unsigned long long)__double_as_longlong(fabs(Pan[tid][kk])) & ~0xFFULL) | (unsigned long long)(255 - tid);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(key, off, WAVE); key = o > key ? o : key; }
                if (lane == 0 && key) atomicMax(&sBest[k & 1], key);
            }
            __syncthreads();
            const unsigned long long best = sBest[k & 1];
            const int p = 255 - (int)(best & 0xFFULL);
            const double piv = Pan[p][kk];
            if (tid < NC) Fm[kk][tid] = tid == p ? 0.0 : Pan[tid][kk] * fast_rcp(piv);
            for (int j = k0 + PB + tid; j < LA; j += IB_NT) {                  // pivot row, right of the panel
                double r = Wk[(size_t)p * LA + j];
                for (int kp = 0; kp < kk; ++kp) r -= Fm[kp][p] * Rw[kp][j];
                Rw[kk][j] = r;
            }
            __syncthreads();
            if (tid == 0) {
                sBest[(k + 1) & 1] = 0ULL; sInv[p] = k; sUsed[p] = 1; sPivVal[k] = piv;
                if ((best >> 8) == 0ULL) atomicOr(&status[b], 4);
            }
            for (int e = tid; e < NC * PB; e += IB_NT) {                        // the panel's own later columns, in LDS
                const int i = e / PB, jj = e - i * PB;
                if (jj > kk && i != p) Pan[i][jj] -= Fm[kk][i] * Pan[p][jj];
            }
            __syncthreads();
        }
        {
            const int tx = tid & 255, ty = tid >> 8;
            const int j0 = k0 + PB + tx, j1 = j0 + 256;
            double r0[PB], r1[PB];
#pragma unroll
            for (int kk = 0; kk < PB; ++kk) { r0[kk] = j0 < LA ? Rw[kk][j0] : 0.0; r1[kk] = j1 < LA ? Rw[kk][j1] : 0.0; }
            constexpr int RG = IB_NT / 256, RF = 4;                           // rows in flight per thread (measured: 4 beats 8 and 14)
            for (int ib = ty; ib < NC; ib += RF * RG) {
                double w0[RF], w1[RF];
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    const int i = ib + RG * u;
                    const double* wr = Wk + (size_t)(i < NC ? i : 0) * LA;
                    w0[u] = (i < NC && j0 < LA) ? wr[j0] : 0.0;
                    w1[u] = (i < NC && j1 < LA) ? wr[j1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < RF; ++u) {
                    const int i = ib + RG * u;
                    if (i < NC) {
                        double a0 = w0[u], a1 = w1[u];
#pragma unroll
                        for (int kk = 0; kk < PB; ++kk) { const double f = Fm[kk][i]; a0 -= f * r0[kk]; a1 -= f * r1[kk]; }
                        double* wr = Wk + (size_t)i * LA;
                        if (j0 < LA) wr[j0] = a0;
                        if (j1 < LA) wr[j1] = a1;
                    }
                }
            }
        }
        __syncthreads();
    }
    dbg_stamp(51);
    // solution rows: row i holds component ks = sInv[i], scaled by its pivot
    double* Mg = Mall + (size_t)bl * mstride;
    for (int e = tid; e < NC * (NC + 1); e += IB_NT) {
        const int i = e / (NC + 1), j = e - i * (NC + 1);
        const int ks = sInv[i];
        const double v = Wk[(size_t)i * LA + NC + j] * fast_rcp(sPivVal[ks]);
        if (j < NC) Mg[(size_t)ks * MP + j] = v;
        else Mg[(size_t)MP * MP + ks] = v;
    }
    double* Pc = Pcall + (size_t)bl * ystride;
    if (!zero_copy) {
        for (int k = wave; k < MP; k += IB_NT / WAVE) {
            const int gk = sCol[k];
            const bool real = k < ncol;
            for (int r = lane; r < n; r += WAVE) Pc[r + (size_t)k * ld] = real ? P[r + (size_t)gk * ld] : 0.0;
        }
    }
    if (tid == 0) { m_out[bl] = ncol; nc_out[bl] = ncol; pc_base_out[bl] = zero_copy ? sCol[0] : -1; }
    dbg_stamp(52);
}
#endif  // INGVIO_ALT_KERNELS

// ---------------------------------------------------------------------------------------------
// K8/K9/K11 in symmetric form across the whole GPU (round 2): the same mathematics as kernels_solve.hip,
//     Pcc = L L^T,  W = L^T A L + s^2 I = L2 L2^T,  M = (L^-T L2^-T)(A L L2^-T)^T,  t = (L^-T L2^-T)(b^T L L2^-T)^T,
// built from the batched blocks of kernels_chol.hip (two Cholesky sweeps with carried rows, three GEMMs) instead of ONE
// workgroup eliminating 216 pivots.  Workspace per filter, column-major, n32 = 6 c_max rounded up to 32 (padding: identity in
// Pcc, zero in A, so every window size up to the capacity runs the same launches):
//     AB [n32+32][n32]   A, then the row b^T           X1, Y1 [2 n32][n32]      [Pcc; I] -> [L; L^-T]
//     X2, Y2 [3 n32+32][n32]   [W; A L; b^T L; L^-T] -> [L2; R1; r1b; R2]
// (X1, X2 are working copies: the right-looking sweep updates their trailing parts in place)
// ---------------------------------------------------------------------------------------------
struct BigWs {
    int n32, ld1, ld2, ldab;
    size_t oAB, oX1, oY1, oX2, oY2, oT, oRef, oU, total;  // oRef: the gauge reference slot of the filter (as a double); oU: factor tiles of launch_lm_chol
    __host__ __device__ explicit BigWs(int n32_) : n32(n32_), ld1(2 * n32_), ld2(3 * n32_ + 32), ldab(n32_ + 32)
    {
        oAB = 0; oX1 = oAB + (size_t)ldab * n32; oY1 = oX1 + (size_t)ld1 * n32; oX2 = oY1 + (size_t)ld1 * n32;
        oY2 = oX2 + (size_t)ld2 * n32; oT = oY2 + (size_t)ld2 * n32; oRef = oT + (size_t)(n32 / 32) * 1024 + n32; oU = oRef + 2;
        const int ntm = (n32 + 15) / 16 <= 12 ? 12 : ((n32 + 15) / 16 <= 14 ? 14 : 16);      // as pick_ntm (kernels_lmchol.hip)
        total = oU + (size_t)(ntm * (ntm + 1) / 2 + ntm) * 256 + (size_t)16 * ntm;
    }
};

// Round 5: the set-up of the solve in two kernels, so that everything that does NOT depend on the measurements - the gauge reference,
// [Pdd; I], its Cholesky sweep (7 of the chain's 22 dependent launches), the Pc copy - can run on a second stream UNDER the gate and
// the Gram kernel (launch_bigwin stage 5).  The reference clone used to be the one with the largest translation information (from
// A); it is now the clone with the smallest prior translation variance (from P alone).  Any reference is exact algebra; the
// conditioning sweep (tests/test_gpu_pinning.py::test_large_window_sigma_scale_sweep) reads 4e-11 against the oracle with either.
__global__ __launch_bounds__(256) void k_big_prep_P(
    CovView cv, FrameView fv, int b0, double* __restrict__ Pcall, int ystride, const int* __restrict__ marg_idx, int* __restrict__ pc_base_out,
    double* __restrict__ ws_all, size_t ws_stride, int n32, int gauge)
{
    constexpr int NC = BIG_NC, MP = NC;
    __shared__ int sCol[NC];
    __shared__ double sTr[BIG_CMAX];
    __shared__ int sRefSlot;
    const BigWs w(n32);
    const int bl = blockIdx.y, b = b0 + bl, tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const int C = fv.n_clones[b], ncol = 6 * C, n = cv.n[b], ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    for (int c = tid; c < NC; c += 256) {
        const int cc = c < ncol ? c : 0;
        sCol[c] = fv.clone_idx[(size_t)b * fv.cmax + cc / 6] + cc % 6;
    }
    if (tid < BIG_CMAX) {
        double t = 1e300;
        if (tid < C) {
            const int i0 = fv.clone_idx[(size_t)b * fv.cmax + tid] + 3;
            t = (P[i0 + (size_t)i0 * ld] + P[i0 + 1 + (size_t)(i0 + 1) * ld]) + P[i0 + 2 + (size_t)(i0 + 2) * ld];
        }
        sTr[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        int best = -1;
        if (gauge) {
            double tb = 1e300;
            for (int c = 0; c < C; ++c) if (sTr[c] < tb) { tb = sTr[c]; best = c; }      // first minimum: deterministic
        }
        sRefSlot = best;
    }
    __syncthreads();
    const int ref6 = sRefSlot >= 0 ? 6 * sRefSlot : (1 << 20);
    auto inref = [&](int i) { return i >= ref6 && i < ref6 + 6; };
    double* ws = ws_all + (size_t)bl * ws_stride;
    double* X1 = ws + w.oX1;
    const int gt = wg * 256 + tid, gn = nwg * 256;
    if (gt == 0) ws[w.oRef] = (double)sRefSlot;
    // [Pdd; I], identity in the padding
    for (int e = gt; e < w.ld1 * n32; e += gn) {
        const int i = e % w.ld1, j = e / w.ld1;
        double v;
        if (i < n32) {
            if (i < ncol && j < ncol && !inref(i) && !inref(j)) {
                v = P[sCol[i] + (size_t)sCol[j] * ld];
                if (sRefSlot >= 0) {
                    const int ri = sCol[ref6 + i % 6], rj = sCol[ref6 + j % 6];
                    v = (v - P[ri + (size_t)sCol[j] * ld]) - (P[sCol[i] + (size_t)rj * ld] - P[ri + (size_t)rj * ld]);
                }
            } else v = i == j ? 1.0 : 0.0;
        } else v = (i - n32 == j) ? 1.0 : 0.0;
        X1[e] = v;
    }
    const bool fused = marg_idx && marg_idx[bl] >= 0;
    bool contig = true;
    for (int c = 0; c < ncol; ++c) contig = contig && (sCol[c] == sCol[0] + c);
    const bool zero_copy = fused && contig && sCol[0] + MP <= ld;
    if (!zero_copy) {
        double* Pc = Pcall + (size_t)bl * ystride;
        for (int e = gt; e < MP * n; e += gn) {
            const int r = e % n, k = e / n;
            Pc[r + (size_t)k * ld] = k < ncol ? P[r + (size_t)sCol[k] * ld] : 0.0;
        }
    }
    if (wg == 0 && tid == 0) pc_base_out[bl] = zero_copy ? sCol[0] : -1;
}

// ... and the part that needs the measurements: [A; b^T] from the chunk partials (reference block zeroed), the activity flag m_out
__global__ __launch_bounds__(256) void k_big_prep_A(
    CovView cv, FrameView fv, int b0, const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
    double* __restrict__ dx_all, int* __restrict__ m_out, int* __restrict__ nc_out, double* __restrict__ ws_all, size_t ws_stride, int n32)
{
    const BigWs w(n32);
    const int bl = blockIdx.y, b = b0 + bl, tid = threadIdx.x, nwg = gridDim.x, wg = blockIdx.x;
    const int C = fv.n_clones[b], ncol = 6 * C, n = cv.n[b], ld = cv.ldp;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        double* dx = dx_all + (size_t)b * ld;
        for (int r = wg * 256 + tid; r < n; r += nwg * 256) dx[r] = 0.0;
        if (wg == 0 && tid == 0) { m_out[bl] = 0; nc_out[bl] = ncol; }
        return;
    }
    double* ws = ws_all + (size_t)bl * ws_stride;
    const int ref = (int)ws[w.oRef];
    const int ref6 = ref >= 0 ? 6 * ref : (1 << 20);
    auto inref = [&](int i) { return i >= ref6 && i < ref6 + 6; };
    double* AB = ws + w.oAB;
    const int gt = wg * 256 + tid, gn = nwg * 256;
    // A symmetric: element (i, j) read as partial row j, column i - coalesced along i; all chunks' loads of an element in flight together
    for (int e = gt; e < w.ldab * n32; e += gn) {
        const int i = e % w.ldab, j = e / w.ldab;
        double s = 0.0;
        if (j < ncol && (i < ncol || i == n32) && !inref(i) && !inref(j)) {
            const size_t src = (size_t)j * (ncol + 1) + (i == n32 ? ncol : i);
            for (int g0 = 0; g0 < G; g0 += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int g = g0 + u;
                    t[u] = (g < G && chunk_used[bl * G + g]) ? Apart[((size_t)bl * G + g) * rstride + src] : 0.0;
                }
                s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
        }
        AB[e] = s;
    }
    if (wg == 0 && tid == 0) { m_out[bl] = ncol; nc_out[bl] = ncol; }
}

// The reference clone's block row / column of [M | t]: minus the sums over the other clones' blocks (every block row and column
// of the n x n solution sums to zero: M = T^-T diag(0, Mr) T^-1, see k_info_solve), the 6 x 6 corner = + the sum of ALL other blocks.
// ONE launch, every load coalesced along a row of M (round 6; before: a thread per output walked its 30 blocks at a stride of six
// doubles or six rows - 64 different lines per load instruction, 45 k line requests from the corner's one workgroup alone: 27-31 us
// per 32 filters for 2.6 k sums):
//   workgroups 0..5 (k): thread J sums column J over the rows 6 c + k, c != ref  ->  block row (ref, k) = - that sum; the same column
//     sums, added over the columns 6 c + l through LDS, are the corner's row k (the double sum, without waiting for anybody);
//     thread MP does the same for t;
//   workgroups 6.. : GF_ROWS rows of M each, staged whole in LDS (coalesced), thread (row, k) sums its 30 entries  ->  block column.
// Rows / columns of the reference block itself are read by nobody who uses them (masked), so the writers race with no reader.
#define GF_ROWS 18
__global__ __launch_bounds__(256) void k_big_gauge_fix(FrameView fv, int b0, double* __restrict__ Mall, int mstride, const double* __restrict__ ws_all,
                                                       size_t ws_stride, size_t oref, const int* __restrict__ active)
{
    constexpr int MP = BIG_NC;
    __shared__ double sRow[GF_ROWS][MP + 1];
    const int bl = blockIdx.y, tid = threadIdx.x;
    if (active && !active[bl]) return;
    const int ref = (int)ws_all[(size_t)bl * ws_stride + oref];
    if (ref < 0) return;
    const int C = fv.n_clones[b0 + bl], ref6 = 6 * ref;
    double* Mg = Mall + (size_t)bl * mstride;
    if (blockIdx.x < 6) {
        const int k = blockIdx.x, J = tid;
        double* sCS = &sRow[0][0];
        double s = 0.0;
        if (J <= MP) {
            const double* col = J < MP ? Mg + (size_t)k * MP + J : Mg + (size_t)MP * MP + k;      // element of row 6 c + k: + c * step
            const size_t step = J < MP ? (size_t)6 * MP : 6;
            double v[BIG_CMAX];
#pragma unroll
            for (int c = 0; c < BIG_CMAX; ++c) v[c] = col[min(c, C - 1) * step];                   // all loads in flight, masked by a factor
#pragma unroll
            for (int c = 0; c < BIG_CMAX; ++c) s += (c < C && c != ref ? 1.0 : 0.0) * v[c];
            const bool inref = J >= ref6 && J < ref6 + 6;
            if (J == MP) Mg[(size_t)MP * MP + ref6 + k] = -s;
            else if (!inref) Mg[(size_t)(ref6 + k) * MP + J] = -s;
            if (J < MP) sCS[J] = inref ? 0.0 : s;
        }
        __syncthreads();
        if (tid < 6) {
            double r = 0.0;
            for (int c = 0; c < C; ++c) r += (c != ref ? 1.0 : 0.0) * sCS[6 * c + tid];
            Mg[(size_t)(ref6 + k) * MP + ref6 + tid] = r;
        }
        return;
    }
    const int J0 = (blockIdx.x - 6) * GF_ROWS;
    constexpr int PER = (GF_ROWS * MP + 255) / 256;
    double v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int e = min(tid + 256 * u, GF_ROWS * MP - 1); v[u] = Mg[(size_t)(J0 + e / MP) * MP + e % MP]; }
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int e = tid + 256 * u; if (e < GF_ROWS * MP) sRow[e / MP][e % MP] = v[u]; }
    __syncthreads();
    if (tid < 6 * GF_ROWS) {
        const int r = tid / 6, k = tid - 6 * r, J = J0 + r;
        if (J < ref6 || J >= ref6 + 6) {
            double sum = 0.0;
            for (int c = 0; c < C; ++c) sum += (c != ref ? 1.0 : 0.0) * sRow[r][6 * c + k];
            Mg[(size_t)J * MP + ref6 + k] = -sum;
        }
    }
}

static int big_n32(int ncol_cap) { const int n = ncol_cap > 0 ? ncol_cap : BIG_NC; return (n + 31) / 32 * 32; }

// the measurement-independent front of the solve (stage 5): gauge reference, [Pdd; I] and its Cholesky sweep -> [L; L^-T], the Pc copy.
// No activity flags yet (m_out is written by k_big_prep_A): every filter of the range runs.
// part 0: both; 1: the set-up kernel only (gauge reference, [Pdd; I], Pc - stage 6); 2: the sweep only (stage 7)
static void launch_big_solve_front(const FactoredLaunch& L, hipStream_t st, int part = 0)
{
    const int n32 = big_n32(L.ncol_cap);
    const BigWs w(n32);
    const size_t wss = bigwin_wk_doubles();
    double* ws = L.big_wk;
#ifdef INGVIO_ALT_KERNELS
    static const bool no_gauge = [] { const char* e = getenv("INGVIO_INFO_GAUGE"); return e && !strcmp(e, "off"); }();
#else
    constexpr bool no_gauge = false;
#endif
    const int gauge = (!L.op.selected_variant && !no_gauge) ? 1 : 0;
    if (part != 2)
        hipLaunchKernelGGL(k_big_prep_P, dim3(32, L.nb), dim3(256), 0, st, L.cv, L.fv, L.b0, L.Pc, L.ystride, L.marg_idx, L.pc_base, ws, wss, n32, gauge);
    if (part == 1) return;
    CholArgs c1 = {};
    c1.W = ws + w.oX1; c1.Y = ws + w.oY1; c1.xs = wss; c1.ld = w.ld1; c1.rows = 2 * n32; c1.ncols = n32;
    c1.status = L.status + L.b0; c1.fail_bit = 4; c1.active = nullptr; c1.batch = L.nb; c1.Tb = ws + w.oT; c1.ts = wss; c1.t_slots = n32 / 32;
    // L^-T (the carried rows) goes straight into the second sweep's working copy as well: no k_copy_rows launch on the critical path
    c1.Y2 = ws + w.oX2; c1.y2s = wss; c1.ld_y2 = w.ld2; c1.y2_row0 = 2 * n32 + 32;
    launch_chol_sweep(c1, st);
}

// part 0: all of it; 1: [A; b^T] from the chunk partials only (stage 8: needs the Gram kernel and the gauge reference of stage 6, not the
// sweep of stage 7); 2: the rest (stage 9)
// The three products have a triangular operand each (L, L^T, R2 = L^-T L2^-T): the chunks of K in front of a block's first row /
// column are exact zeros and are skipped (GemmArgs::k_from; round 6) - about a third of the products' MFMAs at 192-224 columns.
#ifndef BIG_GEMM_TRI
#define BIG_GEMM_TRI 1
#endif
static void launch_big_solve(const FactoredLaunch& L, hipStream_t st, int part = 0)
{
    const int n32 = big_n32(L.ncol_cap), MP = BIG_NC;
    const BigWs w(n32);
    const size_t wss = bigwin_wk_doubles();
    double* ws = L.big_wk;
#ifdef INGVIO_ALT_KERNELS
    static const bool no_gauge = [] { const char* e = getenv("INGVIO_INFO_GAUGE"); return e && !strcmp(e, "off"); }();
#else
    constexpr bool no_gauge = false;
#endif
    const int gauge = (!L.op.selected_variant && !no_gauge) ? 1 : 0;
#ifdef INGVIO_ALT_KERNELS
    static const bool tri_off = [] { const char* e = getenv("INGVIO_BIG_GEMM"); return e && !strcmp(e, "full"); }();      // the products over all of K, for comparison
#else
    constexpr bool tri_off = false;
#endif
    const bool tri = BIG_GEMM_TRI && !tri_off;
    if (part != 2)
        hipLaunchKernelGGL(k_big_prep_A, dim3(32, L.nb), dim3(256), 0, st, L.cv, L.fv, L.b0, L.Apart, L.chunk_used, L.G, L.rstride, L.dx, L.m_out,
                           L.nc_out, ws, wss, n32);
    if (part == 1) return;
    const int* act = L.m_out;                                            // 0 = nothing accepted: every later launch skips the filter
    CholArgs c1 = {};
    c1.W = ws + w.oX1; c1.Y = ws + w.oY1; c1.xs = wss; c1.ld = w.ld1; c1.rows = 2 * n32; c1.ncols = n32;
    c1.status = L.status + L.b0; c1.fail_bit = 4; c1.active = act; c1.batch = L.nb; c1.Tb = ws + w.oT; c1.ts = wss; c1.t_slots = n32 / 32;
    GemmArgs g = {};
    // [A; b^T] L -> X2 rows n32 ..
    g.A = ws + w.oAB; g.sa = wss; g.lda = w.ldab; g.modeA = 0;
    g.B = ws + w.oY1; g.sb = wss; g.ldb = w.ld1; g.modeB = 1;
    g.C = ws + w.oX2 + n32; g.sc = wss; g.rs = 1; g.cs = w.ld2;
    g.M = n32 + 32; g.N = n32; g.K = n32; g.m_lim = n32 + 32; g.n_lim = n32; g.ksplit = 1; g.active = act; g.batch = L.nb;
    g.k_from = tri ? 2 : 0;                                     // L[k][j] = 0 for k < j
    launch_gemm(g, st);
    // W = L^T (A L) + s^2 I -> X2 rows 0 .. (lower blocks)
    g = GemmArgs{};
    g.A = ws + w.oY1; g.sa = wss; g.lda = w.ld1; g.modeA = 1;
    g.B = ws + w.oX2 + n32; g.sb = wss; g.ldb = w.ld2; g.modeB = 1;
    g.C = ws + w.oX2; g.sc = wss; g.rs = 1; g.cs = w.ld2;
    g.M = n32; g.N = n32; g.K = n32; g.m_lim = n32; g.n_lim = n32; g.ksplit = 1; g.lower = 1; g.diag_add_vec = L.noise;
    g.active = act; g.batch = L.nb;
    g.k_from = tri ? 1 : 0;                                     // (L^T)[i][k] = L[k][i] = 0 for k < i
    launch_gemm(g, st);
    CholArgs c2 = c1;                                                   // (rows 2 n32 + 32 .. of X2 = L^-T: written by the first sweep itself, CholArgs::Y2)
    c2.W = ws + w.oX2; c2.Y = ws + w.oY2; c2.ld = w.ld2; c2.rows = 3 * n32 + 32;
    launch_chol_sweep(c2, st);
    // M = R2 R1^T (row-major, MP wide), t = R2 r1b^T
    g = GemmArgs{};
    g.A = ws + w.oY2 + 2 * n32 + 32; g.sa = wss; g.lda = w.ld2; g.modeA = 0;
    g.B = ws + w.oY2 + n32; g.sb = wss; g.ldb = w.ld2; g.modeB = 0;
    g.C = L.T; g.sc = L.mstride; g.rs = MP; g.cs = 1;
    // t rides on the same launch: r1b is the row of Y2 right below R1, i.e. column n32 of the product, stored as the vector after M
    g.M = n32; g.N = n32 + 1; g.K = n32; g.m_lim = MP; g.n_lim = n32 < MP ? n32 : MP; g.ksplit = 1; g.active = act; g.batch = L.nb;
    g.Cx = L.T + (size_t)MP * MP; g.scx = L.mstride; g.cx_col = n32;
    g.k_from = tri ? 1 : 0;                                     // R2 = L^-T L2^-T is upper triangular: R2[i][k] = 0 for k < i
    launch_gemm(g, st);
    // one launch for the reference clone's block row / column AND its 6 x 6 corner (the corner from M itself: a double sum)
    if (gauge) hipLaunchKernelGGL(k_big_gauge_fix, dim3(6 + BIG_NC / GF_ROWS, L.nb), dim3(256), 0, st, L.fv, L.b0, L.T, L.mstride, ws, wss, w.oRef, act);
}

// ---------------------------------------------------------------------------------------------
// K10 (+ fused K12), see k_info_apply: one wave per 16-row tile row, T row in LDS, K streamed.
// ---------------------------------------------------------------------------------------------
struct ABShared {
    static constexpr int MP = BIG_NC;
    double sT[4][16][MP + 2];
    double sB[MP][16];
    double sV[4][16][17];
};

__global__ __launch_bounds__(256, 1) void k_info_apply_big(
    CovView cv, int b0, int nb, int wgpf, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
    int ystride, const int* __restrict__ m_all, double* __restrict__ dx_all, int* __restrict__ status,
    const int* __restrict__ marg_idx, int msize, const int* __restrict__ pc_base)
{
    constexpr int MP = BIG_NC, K4 = MP / 4, JT = (MP + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    ABShared& sh = *reinterpret_cast<ABShared*>(smem_raw);
    const int wg = blockIdx.x, xcd = wg & 7, tq = wg >> 3;
    const int bl = xcd + 8 * (tq / wgpf), part = tq % wgpf;
    if (bl >= nb) return;
    const int b = b0 + bl;
    const bool upd = m_all[bl] != 0;
    const int midx = marg_idx ? marg_idx[bl] : -1;
    const bool fused = midx >= 0;
    if (!upd && !fused) return;
    const int n = cv.n[b], ld = cv.ldp, nt = (n + 15) >> 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (part * 4 >= nt) return;
    const int ti = part * 4 + wave;
    const bool wave_on = ti < nt;
    const int tjmax = min(nt - 1, part * 4 + 3);
    const double* P = cov_ptr(cv, b);
    double* dst = fused ? cov_alt_ptr(cv, b) : cov_ptr(cv, b);
    const int pcb = upd ? pc_base[bl] : -1;
    const double* Pc = pcb >= 0 ? P + (size_t)pcb * ld : Pcall + (size_t)bl * ystride;
    const double* M = Mall + (size_t)bl * mstride;
    const double* tvec = M + (size_t)MP * MP;
    const int l15 = lane & 15, kq = lane >> 4;
    auto alive = [&](int i) { return !(fused && i >= midx && i < midx + msize); };
    auto remap = [&](int i) { return (fused && i >= midx) ? i - msize : i; };

    if (upd && wave_on) {
        const int ra = min(ti * 16 + l15, n - 1);
        double afrag[K4];
#pragma unroll
        for (int k4 = 0; k4 < K4; ++k4) afrag[k4] = (Pc + (size_t)(4 * k4) * ld)[ra + kq * ld];
        {
            double d = 0.0;
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) d += afrag[k4] * tvec[4 * k4 + kq];
            d += __shfl_xor(d, 16, WAVE);
            d += __shfl_xor(d, 32, WAVE);
            if (kq == 0 && ti * 16 + l15 < n) dx_all[(size_t)b * ld + ra] = d;
        }
#pragma unroll 1
        for (int jt = 0; jt < JT; ++jt) {
            double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
            const int jc = min(jt * 16 + l15, MP - 1);
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) {
                const double bf = (M + (size_t)(4 * k4) * MP)[kq * MP + jc];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[k4], bf, acc, 0, 0, 0);
            }
            if (jt * 16 + l15 < MP) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sh.sT[wave][kq + 4 * r][jc] = acc[r];
            }
        }
    }
    __syncthreads();
    for (int tj = 0; tj <= tjmax; ++tj) {
        // stage the B tile Pc[16 tj .. +16][0..MP) once for the four waves
        if (upd) {
            for (int e = tid; e < MP * 16; e += 256) {
                const int k = e >> 4, r = e & 15;
                sh.sB[k][r] = Pc[min(16 * tj + r, n - 1) + (size_t)k * ld];
            }
        }
        __syncthreads();
        if (wave_on && tj <= ti) {
            double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
            if (upd) {
#pragma unroll 6
                for (int k4 = 0; k4 < K4; ++k4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.sT[wave][l15][4 * k4 + kq], sh.sB[4 * k4 + kq][l15], acc, 0, 0, 0);
            }
            const int col = tj * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + kq + 4 * r;
                double v = 0.0;
                if (row < n && col < n && row >= col) {
                    v = P[col + (size_t)row * ld] - acc[r];
                    if (alive(row) && alive(col)) dst[remap(col) + (size_t)remap(row) * ld] = v;
                    if (upd && row == col && v < 0.0) atomicOr(&status[b], 2);
                }
                sh.sV[wave][kq + 4 * r][l15] = v;
            }
            __builtin_amdgcn_wave_barrier();
            const int row2 = ti * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col2 = tj * 16 + kq + 4 * r;
                if (row2 < n && col2 < n && row2 > col2 && alive(row2) && alive(col2))
                    dst[remap(row2) + (size_t)remap(col2) * ld] = sh.sV[wave][l15][kq + 4 * r];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// K10 (+ fused K12) in two flat launches (round 2): T = Pc M (k_apply_T, one workgroup per 32 x 32 block of T), then
// P <- P - T Pc^T over the lower 32 x 32 blocks (k_apply_sym), both with the four waves of a workgroup splitting K = 6C and
// reducing through LDS.  k_info_apply_big above walks a whole tile row per wave - one filter keeps 13 workgroups busy for
// 0.48 ms; here the same filter spreads over ~350 workgroups.
// ---------------------------------------------------------------------------------------------
struct ApplyPtrs { const double* P; double* dst; const double* Pc; const double* M; int n, ld, midx; bool upd, fused; };

__device__ __forceinline__ bool apply_setup(const CovView& cv, int b0, int bl, const double* Mall, int mstride, const double* Pcall, int ystride,
                                            const int* m_all, const int* marg_idx, const int* pc_base, ApplyPtrs& q)
{
    const int b = b0 + bl;
    q.upd = m_all[bl] != 0;
    q.midx = marg_idx ? marg_idx[bl] : -1;
    q.fused = q.midx >= 0;
    if (!q.upd && !q.fused) return false;
    q.n = cv.n[b]; q.ld = cv.ldp;
    q.P = cov_ptr(cv, b);
    q.dst = q.fused ? cov_alt_ptr(cv, b) : cov_ptr(cv, b);
    const int pcb = q.upd ? pc_base[bl] : -1;
    q.Pc = pcb >= 0 ? q.P + (size_t)pcb * q.ld : Pcall + (size_t)bl * ystride;
    q.M = Mall + (size_t)bl * mstride;
    return true;
}

// four waves split K = MP (padded to 16-chunks), partial tiles summed through LDS: returns this thread's four sums of the
// 32 x 32 block in (r = tid & 31, c = (tid >> 5) + 8 q) order.  opA(i, k) = A[i + k lda] (i < ilim), opB(k, j) = B[j sb_j + k sb_k].
template <class FB>
__device__ __forceinline__ void block_gemm_k216(const double* __restrict__ A, int lda, int i0, int ilim, FB opB, double (&out)[4],
                                                double (*sPart)[4][4][64])
{
    constexpr int MP = BIG_NC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    double4_f c00 = { 0, 0, 0, 0 }, c01 = c00, c10 = c00, c11 = c00;
    const int ia = min(i0 + l15, ilim - 1), ib = min(i0 + 16 + l15, ilim - 1);
    constexpr int CH = (MP + 15) / 16;                                   // 14 chunks of 16
    const int w_lo = wave * 4 < CH ? wave * 4 : CH, w_hi = min(CH, w_lo + 4);
    double a0[4][4], a1[4][4], b0[4][4], b1[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = (w_lo + u) * 16 + 4 * kq + s4;
            const bool on = w_lo + u < w_hi && k < MP;
            const int kk = on ? k : 0;
            a0[u][s4] = on ? A[(size_t)ia + (size_t)kk * lda] : 0.0;
            a1[u][s4] = on ? A[(size_t)ib + (size_t)kk * lda] : 0.0;
            b0[u][s4] = on ? opB(kk, l15) : 0.0;
            b1[u][s4] = on ? opB(kk, 16 + l15) : 0.0;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u][s4], b0[u][s4], c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u][s4], b1[u][s4], c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u][s4], b0[u][s4], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u][s4], b1[u][s4], c11, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sPart[wave][0][r][lane] = c00[r]; sPart[wave][1][r][lane] = c01[r];
        sPart[wave][2][r][lane] = c10[r]; sPart[wave][3][r][lane] = c11[r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = tid & 31, c = (tid >> 5) + 8 * q;
        const int tile = (r >> 4) * 2 + (c >> 4), rr = r & 15, cc = c & 15, sl = (rr & 3) * 16 + cc, sr = rr >> 2;
        out[q] = sPart[0][tile][sr][sl] + sPart[1][tile][sr][sl] + sPart[2][tile][sr][sl] + sPart[3][tile][sr][sl];
    }
}

// T[rows, cols] = Pc[rows, :] M[:, cols] (T column-major, ldt), and dx = Pc t by the workgroups of column block 0
__global__ __launch_bounds__(256) void k_apply_T(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                 int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx,
                                                 const int* __restrict__ pc_base, double* __restrict__ Tall, size_t tstride, int ldt,
                                                 double* __restrict__ dx_all)
{
    constexpr int MP = BIG_NC;
    __shared__ double sPart[4][4][4][64];
    const int bl = blockIdx.z, bi = blockIdx.x, bj = blockIdx.y;
    ApplyPtrs q;
    if (!apply_setup(cv, b0, bl, Mall, mstride, Pcall, ystride, m_all, marg_idx, pc_base, q) || !q.upd) return;
    if (32 * bi >= q.n) return;
    const double* M = q.M;
    double out[4];
    block_gemm_k216(q.Pc, q.ld, 32 * bi, q.n, [&](int k, int jj) { const int j = 32 * bj + jj; return j < MP ? M[(size_t)k * MP + j] : 0.0; }, out, sPart);
    double* T = Tall + (size_t)bl * tstride;
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 32 * bi + (tid & 31), c = 32 * bj + (tid >> 5) + 8 * u;
        if (r < q.n && c < MP) T[(size_t)r + (size_t)c * ldt] = out[u];
    }
    if (bj == 0 && tid < 32) {
        const int r = 32 * bi + tid;
        if (r < q.n) {
            const double* tv = M + (size_t)MP * MP;
            double d0 = 0.0, d1 = 0.0;
            for (int k = 0; k < MP; k += 2) { d0 += q.Pc[(size_t)r + (size_t)k * q.ld] * tv[k]; d1 += q.Pc[(size_t)r + (size_t)(k + 1) * q.ld] * tv[k + 1]; }
            dx_all[(size_t)(b0 + bl) * q.ld + r] = d0 + d1;
        }
    }
}

// lower 32 x 32 block (bi >= bj) of P - T Pc^T, written (with the fused marginalisation's index shift) to both triangles
__global__ __launch_bounds__(256) void k_apply_sym(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                   int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx, int msize,
                                                   const int* __restrict__ pc_base, const double* __restrict__ Tall, size_t tstride, int ldt,
                                                   int* __restrict__ status)
{
    __shared__ double sPart[4][4][4][64];
    __shared__ double sV[32][33];
    const int bl = blockIdx.y;
    int t = blockIdx.x, bi = 0;
    while (t >= bi + 1) { t -= bi + 1; ++bi; }
    const int bj = t;
    ApplyPtrs q;
    if (!apply_setup(cv, b0, bl, Mall, mstride, Pcall, ystride, m_all, marg_idx, pc_base, q)) return;
    if (32 * bi >= q.n) return;
    const int n = q.n, ld = q.ld, midx = q.midx, tid = threadIdx.x;
    const bool fused = q.fused;
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
    if (q.upd) {
        const double* Pc = q.Pc;
        const int j0 = 32 * bj;
        block_gemm_k216(Tall + (size_t)bl * tstride, ldt, 32 * bi, n,
                        [&](int k, int jj) { return Pc[(size_t)min(j0 + jj, n - 1) + (size_t)k * ld]; }, acc, sPart);
    }
    auto alive = [&](int i) { return !(fused && i >= midx && i < midx + msize); };
    auto remap = [&](int i) { return (fused && i >= midx) ? i - msize : i; };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rr = tid & 31, cc = (tid >> 5) + 8 * u;
        const int row = 32 * bi + rr, col = 32 * bj + cc;
        double v = 0.0;
        if (row < n && col < n && (bi > bj || row >= col)) {
            v = q.P[(size_t)row + (size_t)col * ld] - acc[u];
            if (alive(row) && alive(col)) q.dst[(size_t)remap(row) + (size_t)remap(col) * ld] = v;
            if (q.upd && row == col && v < 0.0) atomicOr(&status[b0 + bl], 2);
        }
        sV[rr][cc] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                        // mirror: consecutive threads along the block's columns
        const int cc = tid & 31, rr = (tid >> 5) + 8 * u;
        const int row = 32 * bi + rr, col = 32 * bj + cc;
        if (row < n && col < n && row > col && alive(row) && alive(col)) q.dst[(size_t)remap(col) + (size_t)remap(row) * ld] = sV[rr][cc];
    }
}

#ifdef INGVIO_ALT_KERNELS      // round-4 generation of the 64 x 64 apply (INGVIO_BIG_APPLY=6), variant builds only
// ---- the same two launches on 64 x 64 blocks with the operand panels staged through LDS (block64.h): an element of Pc / M / T is
//      read from L2 once per workgroup instead of once per 32 x 32 block and wave
__global__ __launch_bounds__(256) void k_apply_T64(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                   int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx,
                                                   const int* __restrict__ pc_base, double* __restrict__ Tall, size_t tstride, int ldt,
                                                   double* __restrict__ dx_all, int MP)
{
    __shared__ Block64Lds sAB;
    const int bl = blockIdx.z, bi = blockIdx.x, bj = blockIdx.y;
    ApplyPtrs q;
    if (!apply_setup(cv, b0, bl, Mall, mstride, Pcall, ystride, m_all, marg_idx, pc_base, q) || !q.upd) return;
    if (64 * bi >= q.n) return;
    const double* M = q.M;
    const double* Pc = q.Pc;
    const int n = q.n, ld = q.ld;
    b64_d4 c[4];
    block64_mma(sAB, (MP + 15) & ~15,
                [&](int r, int k) { return k < MP ? Pc[(size_t)min(64 * bi + r, n - 1) + (size_t)k * ld] : 0.0; },
                [&](int r, int k) { const int j = 64 * bj + r; return (k < MP && j < MP) ? M[(size_t)k * MP + j] : 0.0; }, true, c);
    double* T = Tall + (size_t)bl * tstride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4, wi = wave >> 1, wj = wave & 1;
#pragma unroll
    for (int h = 0; h < 4; ++h) {                                        // tile h of the quadrant: rows 16 (h >> 1), columns 16 (h & 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 64 * bi + 32 * wi + 16 * (h >> 1) + kq + 4 * r, col = 64 * bj + 32 * wj + 16 * (h & 1) + l15;
            if (row < n && col < MP) T[(size_t)row + (size_t)col * ldt] = c[h][r];
        }
    }
    if (bj == 0 && tid < 64) {
        const int r = 64 * bi + tid;
        if (r < n) {
            const double* tv = M + (size_t)MP * MP;
            double d0 = 0.0, d1 = 0.0;
            for (int k = 0; k < MP; k += 2) { d0 += Pc[(size_t)r + (size_t)k * ld] * tv[k]; d1 += Pc[(size_t)r + (size_t)(k + 1) * ld] * tv[k + 1]; }
            dx_all[(size_t)(b0 + bl) * ld + r] = d0 + d1;
        }
    }
}

__global__ __launch_bounds__(256) void k_apply_sym64(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                     int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx, int msize,
                                                     const int* __restrict__ pc_base, const double* __restrict__ Tall, size_t tstride, int ldt,
                                                     int* __restrict__ status, int MP)
{
    __shared__ Block64Lds sAB;
    __shared__ double sV[4][32][33];
    const int bl = blockIdx.y;
    int t = blockIdx.x, bi = 0;
    while (t >= bi + 1) { t -= bi + 1; ++bi; }
    const int bj = t;
    ApplyPtrs q;
    if (!apply_setup(cv, b0, bl, Mall, mstride, Pcall, ystride, m_all, marg_idx, pc_base, q)) return;
    if (64 * bi >= q.n) return;
    const int n = q.n, ld = q.ld, midx = q.midx, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wi = wave >> 1, wj = wave & 1;
    const bool fused = q.fused;
    const bool quad_on = !(bi == bj && wj > wi);
    b64_d4 c[4];
    if (q.upd) {
        const double* T = Tall + (size_t)bl * tstride;
        const double* Pc = q.Pc;
        block64_mma(sAB, (MP + 15) & ~15,
                    [&](int r, int k) { return k < MP ? T[(size_t)min(64 * bi + r, n - 1) + (size_t)k * ldt] : 0.0; },
                    [&](int r, int k) { return k < MP ? Pc[(size_t)min(64 * bj + r, n - 1) + (size_t)k * ld] : 0.0; }, quad_on, c);
    } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) c[h] = b64_d4{ 0.0, 0.0, 0.0, 0.0 };
    }
    if (!quad_on) return;
    block64_to_lds(c, sV[wave]);
    auto alive = [&](int i) { return !(fused && i >= midx && i < midx + msize); };
    auto remap = [&](int i) { return (fused && i >= midx) ? i - msize : i; };
    const int r0 = 64 * bi + 32 * wi, c0 = 64 * bj + 32 * wj;
    for (int e = lane; e < 1024; e += 64) {
        const int rr = e & 31, cc = e >> 5;
        const int row = r0 + rr, col = c0 + cc;
        double v = 0.0;
        if (row < n && col < n && row >= col) {
            v = NT_LOAD(&q.P[(size_t)row + (size_t)col * ld]) - sV[wave][rr][cc];
            if (alive(row) && alive(col)) NT_STORE(&q.dst[(size_t)remap(row) + (size_t)remap(col) * ld], v);
            if (q.upd && row == col && v < 0.0) atomicOr(&status[b0 + bl], 2);
        }
        sV[wave][rr][cc] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 1024; e += 64) {
        const int cc = e & 31, rr = e >> 5;
        const int row = r0 + rr, col = c0 + cc;
        if (row < n && col < n && row > col && alive(row) && alive(col)) NT_STORE(&q.dst[(size_t)remap(col) + (size_t)remap(row) * ld], sV[wave][rr][cc]);
    }
}

#endif  // INGVIO_ALT_KERNELS

// ---- round 5: the two launches again with (a) the K range cut to the context's window class (kc = 6 c_max columns instead of
//      the 216 of the largest class: 12 instead of 14 chunks and 3 instead of 4 column blocks of T at 30 clones), (b) operands
//      prefetched two chunks ahead through a double-buffered LDS stage (block64_mma2), (c) dx = Pc t as column kc of the T product
//      instead of a 216-step serial loop on 64 threads, (d) the prior's block of k_apply_sym64 requested BEFORE the product: its
//      16 loads per lane used to start after the last MFMA, one dependent load -> subtract -> store chain per element.
__global__ __launch_bounds__(256, 3) void k_apply_T64b(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                    int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx,
                                                    const int* __restrict__ pc_base, double* __restrict__ Tall, size_t tstride, int ldt,
                                                    double* __restrict__ dx_all, int MP, int kc, int nb, int nbi, int nbj)
{
    __shared__ Block64Lds2 sAB;
    // XCD-aware order (workgroup w runs on XCD w % 8): XCD x walks through the filters x, x + 8, ... ONE AFTER THE OTHER, all blocks
    // of a filter on the same XCD - its Pc (1.2 MB at N = 807) and M (0.26 MB) are then served by that XCD's 4 MB L2 instead of
    // travelling from MALL / HBM once per block (536 MB per 32 filters in k_apply_sym64)
    const int per = nbi * nbj, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bl = xcd + 8 * (slot / per), blk = slot % per, bj = blk / nbi, bi = blk - bj * nbi;
    if (bl >= nb) return;
    ApplyPtrs q;
    if (!apply_setup(cv, b0, bl, Mall, mstride, Pcall, ystride, m_all, marg_idx, pc_base, q) || !q.upd) return;
    if (64 * bi >= q.n) return;
    const double* M = q.M;
    const double* tv = M + (size_t)MP * MP;
    const double* Pc = q.Pc;
    const int n = q.n, ld = q.ld;
    b64_d4 c[4];
    const int tid = threadIdx.x;
    const int arow = min(64 * bi + (tid & 63), n - 1), jcol = 64 * bj + (tid & 63);
    const double* Arow = Pc + arow;
    const double* Bcol = jcol < kc ? M + jcol : tv;                      // column kc of the product is dx = Pc t
    const size_t bstep = jcol < kc ? (size_t)MP : 1;
    const bool bon = jcol <= kc;
    // UNCONDITIONAL loads from clamped (valid, finite) addresses, zeroed by a factor: `k < kc ? p[k] : 0` compiles into one exec-masked
    // basic block per load, and the compiler - unable to count the loads in flight across those branches - drains the whole queue
    // (s_waitcnt vmcnt(0)) before every LDS stage: the two-chunk prefetch collapsed to none
    const double bmask = bon ? 1.0 : 0.0;
    block64_mma2(sAB, (kc + 15) & ~15,
                 [&](int, int k) { return Arow[(size_t)min(k, kc - 1) * ld]; },
                 [&](int, int k) { return (k < kc ? bmask : 0.0) * Bcol[(size_t)min(k, kc - 1) * bstep]; }, true, c);
    double* T = Tall + (size_t)bl * tstride;
    const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4, wi = wave >> 1, wj = wave & 1;
#pragma unroll
    for (int h = 0; h < 4; ++h) {                                        // tile h of the quadrant: rows 16 (h >> 1), columns 16 (h & 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 64 * bi + 32 * wi + 16 * (h >> 1) + kq + 4 * r, col = 64 * bj + 32 * wj + 16 * (h & 1) + l15;
            if (row < n) {
                if (col < kc) T[(size_t)row + (size_t)col * ldt] = c[h][r];
                else if (col == kc) dx_all[(size_t)(b0 + bl) * ld + row] = c[h][r];
            }
        }
    }
}

__global__ __launch_bounds__(256, 3) void k_apply_sym64b(CovView cv, int b0, const double* __restrict__ Mall, int mstride, const double* __restrict__ Pcall,
                                                      int ystride, const int* __restrict__ m_all, const int* __restrict__ marg_idx, int msize,
                                                      const int* __restrict__ pc_base, const double* __restrict__ Tall, size_t tstride, int ldt,
                                                      int* __restrict__ status, int kc, int nb, int per)
{
    __shared__ union { Block64Lds2 ab; double sV[4][32][33]; } sh;      // the transposition scratch reuses the operand stage
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;              // XCD-aware order, see k_apply_T64b: row-major over the lower blocks,
    const int bl = xcd + 8 * (slot / per);                               // so consecutive workgroups of an XCD share the T panel of a block row
    if (bl >= nb) return;
    int t = slot % per, bi = 0;
    while (t >= bi + 1) { t -= bi + 1; ++bi; }
    const int bj = t;
    ApplyPtrs q;
    if (!apply_setup(cv, b0, bl, Mall, mstride, Pcall, ystride, m_all, marg_idx, pc_base, q)) return;
    if (64 * bi >= q.n) return;
    const int n = q.n, ld = q.ld, midx = q.midx, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wi = wave >> 1, wj = wave & 1;
    const bool fused = q.fused;
    const bool quad_on = !(bi == bj && wj > wi);
    const int r0 = 64 * bi + 32 * wi, c0 = 64 * bj + 32 * wj;
    // the prior's quadrant: 16 values per lane, in flight during the product
    double pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, row = r0 + (e & 31), col = c0 + (e >> 5);
        pv[i] = (quad_on && row < n && col < n && row >= col) ? NT_LOAD(&q.P[(size_t)row + (size_t)col * ld]) : 0.0;
    }
    b64_d4 c[4];
    if (q.upd) {
        const double* Arow = Tall + (size_t)bl * tstride + min(64 * bi + (tid & 63), n - 1);
        const double* Brow = q.Pc + min(64 * bj + (tid & 63), n - 1);
        block64_mma2(sh.ab, (kc + 15) & ~15,
                     [&](int, int k) { return Arow[(size_t)min(k, kc - 1) * ldt]; },
                     [&](int, int k) { return (k < kc ? 1.0 : 0.0) * Brow[(size_t)min(k, kc - 1) * ld]; }, quad_on, c);
    } else {
#pragma unroll
        for (int h = 0; h < 4; ++h) c[h] = b64_d4{ 0.0, 0.0, 0.0, 0.0 };
    }
    if (!quad_on) return;
    block64_to_lds(c, sh.sV[wave]);
    auto alive = [&](int i) { return !(fused && i >= midx && i < midx + msize); };
    auto remap = [&](int i) { return (fused && i >= midx) ? i - msize : i; };
    bool neg = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, rr = e & 31, cc = e >> 5;
        const int row = r0 + rr, col = c0 + cc;
        double v = 0.0;
        if (row < n && col < n && row >= col) {
            v = pv[i] - sh.sV[wave][rr][cc];
            if (alive(row) && alive(col)) NT_STORE(&q.dst[(size_t)remap(row) + (size_t)remap(col) * ld], v);
            neg = neg || (q.upd && row == col && v < 0.0);
        }
        sh.sV[wave][rr][cc] = v;
    }
    if (neg) atomicOr(&status[b0 + bl], 2);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, cc = e & 31, rr = e >> 5;
        const int row = r0 + rr, col = c0 + cc;
        if (row < n && col < n && row > col && alive(row) && alive(col)) NT_STORE(&q.dst[(size_t)remap(col) + (size_t)remap(row) * ld], sh.sV[wave][rr][cc]);
    }
}

// ---------------------------------------------------------------------------------------------
int dbg_read_bigwin(long long* out, int n) { return dbg_read_local(out, n); }
size_t bigwin_sg_doubles(int G) { return (size_t)G * BIG_CMAX * BIG_CMAX * GB_SW; }
size_t bigwin_wk_doubles() { const size_t gj = (size_t)BIG_NC * (2 * BIG_NC + 1), ch = BigWs(big_n32(BIG_NC)).total; return gj > ch ? gj : ch; }
int bigwin_rec_size() { return rec_size(BIG_CMAX); }
int bigwin_cmax() { return BIG_CMAX; }

// T = Pc M and P - T Pc^T on 64 x 64 blocks; mp = row length of M (row-major mp x mp, then t), T [n][mp] column-major (ldt)
void launch_apply64(const FactoredLaunch& L, hipStream_t st, int mp, double* T, size_t tstride, int ldt)
{
    const int nb64 = (ldt + 63) / 64;
#ifdef INGVIO_ALT_KERNELS
    static const bool old64 = [] { const char* e = getenv("INGVIO_BIG_APPLY"); return e && e[0] == '6'; }();      // round-4 kernels, for comparison
#endif
#ifdef INGVIO_ALT_KERNELS
    if (old64) {
        hipLaunchKernelGGL(k_apply_T64, dim3(nb64, (mp + 63) / 64, L.nb), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride,
                           L.m_out, L.marg_idx, L.pc_base, T, tstride, ldt, L.dx, mp);
        hipLaunchKernelGGL(k_apply_sym64, dim3(nb64 * (nb64 + 1) / 2, L.nb), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride,
                           L.m_out, L.marg_idx, L.marg_size, L.pc_base, T, tstride, ldt, L.status, mp);
        return;
    }
#endif
    {
        const int kc = (L.ncol_cap > 0 && L.ncol_cap < mp) ? L.ncol_cap : mp;      // the window class of the context: M, t are zero beyond it
        const int fpx = (L.nb + 7) / 8, nbj = (kc + 1 + 63) / 64, per = nb64 * (nb64 + 1) / 2;      // filters per XCD
        hipLaunchKernelGGL(k_apply_T64b, dim3(8 * fpx * nb64 * nbj), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride,
                           L.m_out, L.marg_idx, L.pc_base, T, tstride, ldt, L.dx, mp, kc, L.nb, nb64, nbj);
        hipLaunchKernelGGL(k_apply_sym64b, dim3(8 * fpx * per), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride,
                           L.m_out, L.marg_idx, L.marg_size, L.pc_base, T, tstride, ldt, L.status, kc, L.nb, per);
    }
}

int launch_bigwin(const FactoredLaunch& L, hipStream_t st)
{
    if (L.fv.cmax > BIG_CMAX) return -1;
    if (L.stage == 0) {
        const int cm = L.ncol_cap > 0 ? L.ncol_cap / 6 : BIG_CMAX;                 // the context's c_max picks the class
        if (cm <= 24) { if (L.stereo) launch_gate_big<true, 24>(L, st); else launch_gate_big<false, 24>(L, st); }
        // round 5: stereo classes 28 and 30.  The two-wave gate needs 231 VGPRs (2 waves per SIMD = 4 workgroups per CU at most) and its
        // LDS is dominated by the packed triangle of K_r, 3 (C - 1) (3 (C - 1) + 1) / 2 doubles: the 32 class takes 45 KB = THREE
        // workgroups per CU, the 30 class 40.1 KB and the 28 class 36 KB = four.  A 30-clone window (BASELINE config 5) and the
        // reference's 25 / 27-pose configs (+ the clone of the frame) no longer pay for two clones they do not have.
        else if (cm <= 28 && L.stereo) launch_gate_big<true, 28>(L, st);
        else if (cm <= 30 && L.stereo) launch_gate_big<true, 30>(L, st);
        else if (cm <= 32) { if (L.stereo) launch_gate_big<true, 32>(L, st); else launch_gate_big<false, 32>(L, st); }
        else { if (L.stereo) launch_gate_big<true, BIG_CMAX>(L, st); else launch_gate_big<false, BIG_CMAX>(L, st); }
        return 0;
    }
    if (L.stage == 5) { launch_big_solve_front(L, st); return 0; }      // the measurement-independent front of stage 2 (may run on a side stream)
    if (L.stage == 6 || L.stage == 7) { launch_big_solve_front(L, st, L.stage - 5); return 0; }      // ... in two pieces: set-up kernel, sweep
    if (L.stage == 8 || L.stage == 9) { launch_big_solve(L, st, L.stage - 7); return 0; }           // stage 2 in two pieces: [A; b^T], the rest
    if (L.stage == 1) {
        const size_t sm = ((sizeof(GBBatch) + 15) / 16) * 16 + 2 * sizeof(int) * (size_t)L.fv.fmax;
        static size_t attr_sm = 0;
        if (sm > attr_sm) { hipFuncSetAttribute((const void*)k_feat_gram_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_sm = sm; }
        hipLaunchKernelGGL(k_feat_gram_big, dim3(L.G, L.nb), dim3(GB_NT), sm, st, L.fv, L.op, L.b0, L.accept, L.used, L.rec,
                           L.Apart, L.chunk_used, L.G, L.rstride, L.big_sg);
        return 0;
    }
    if (L.stage == 2) {
#ifdef INGVIO_ALT_KERNELS
        static const bool use_gj = [] { const char* e = getenv("INGVIO_BIG_SOLVE"); return e && e[0] == 'g'; }();
        if (use_gj) {
            hipLaunchKernelGGL(k_info_update_big, dim3(L.nb), dim3(IB_NT), 0, st, L.cv, L.fv, L.b0, L.Apart, L.chunk_used, L.G,
                               L.rstride, L.noise, L.T, L.mstride, L.Pc, L.ystride, L.dx, L.m_out, L.nc_out, L.status, L.marg_idx,
                               L.pc_base, L.big_wk);
            return 0;
        }
#endif
        launch_big_solve(L, st);
        return 0;
    }
#ifdef INGVIO_ALT_KERNELS
    static const bool old_apply = [] { const char* e = getenv("INGVIO_BIG_APPLY"); return e && e[0] == 'o'; }();
#else
    constexpr bool old_apply = false;
#endif
    if (!old_apply) {
        // T lives in the solve workspace's X2/Y2 region (free once M has been extracted)
        const BigWs w(big_n32(L.ncol_cap));
        const int ldt = (L.n_cap + 31) / 32 * 32, nbr = ldt / 32;
        if ((size_t)ldt * BIG_NC <= 2 * (size_t)w.ld2 * w.n32) {
            double* T = L.big_wk + w.oX2;
            const size_t wss = bigwin_wk_doubles();
#ifdef INGVIO_ALT_KERNELS
            static const bool blk32 = [] { const char* e = getenv("INGVIO_BIG_APPLY"); return e && e[0] == '3'; }();
#else
            constexpr bool blk32 = false;
#endif
            if (blk32 || L.nb < 4) {                                          // a few filters: more, smaller workgroups
                hipLaunchKernelGGL(k_apply_T, dim3(nbr, (BIG_NC + 31) / 32, L.nb), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride,
                                   L.m_out, L.marg_idx, L.pc_base, T, wss, ldt, L.dx);
                hipLaunchKernelGGL(k_apply_sym, dim3(nbr * (nbr + 1) / 2, L.nb), dim3(256), 0, st, L.cv, L.b0, L.T, L.mstride, L.Pc, L.ystride,
                                   L.m_out, L.marg_idx, L.marg_size, L.pc_base, T, wss, ldt, L.status);
                return 0;
            }
            launch_apply64(L, st, BIG_NC, T, wss, ldt);
            return 0;
        }
    }
    const int nt = (L.n_cap + 15) / 16, wgpf = (nt + 3) / 4, nb8 = (L.nb + 7) / 8 * 8;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)k_info_apply_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ABShared));
        attr_done = true;
    }
    hipLaunchKernelGGL(k_info_apply_big, dim3(nb8 * wgpf), dim3(256), sizeof(ABShared), st, L.cv, L.b0, L.nb, wgpf, L.T, L.mstride,
                       L.Pc, L.ystride, L.m_out, L.dx, L.status, L.marg_idx, L.marg_size, L.pc_base);
    return 0;
}
