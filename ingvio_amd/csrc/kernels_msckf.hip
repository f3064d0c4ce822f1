// kernels_msckf.hip — K3 (per-feature Jacobian), K4 (left-nullspace by 3 Householder reflectors),
// K5 (chi^2 gate against the prior), K6/K7 (stack + tall-skinny QR compression as a TSQR tree).
//
// Layout ("column owner"): a workgroup of NT = 64*ceil((6C+1)/64) lanes holds one feature's
// stacked block [Hx | r] with lane c owning column c (c < 6C: Jacobian column of window slot c/6,
// component c%6; c == 6C: the residual) in RR = rows-per-obs * CMAX FP64 registers.  Householder
// reflectors are then pure broadcast (the reflector vector through LDS) + per-lane FMAs: no
// cross-lane reductions in the hot loops.  The running R factor of the TSQR lives in LDS in packed
// upper-trapezoid form (row k holds columns k..6C).  gfx950 only.
#include "dev_common.h"

#include "feat_build.h"

// ---------------------------------------------------------------------------------------------
// K3 + K4.  RemoveLostUpdate.cpp:407-523 (stereo) / :169-273 (mono); selected-timestamp twins
// SwMargUpdate.cpp:499-700, KeyframeUpdate.cpp:330-415 (selected_variant: anchor block assigned,
// quirk Q10).  On return lane c holds the projected block column in B[3 .. rows), rows = RPO*nobs.
// Must be called by all NT lanes.  Returns rows (0 if the feature has no usable observation).
// ---------------------------------------------------------------------------------------------
template <int CMAX, bool STEREO>
__device__ __forceinline__ int build_feature(const FrameView& fv, const MsckfOpts& op, int b, int j, int C,
                                             FeatShared<CMAX, STEREO>& sh,
                                             double (&B)[FeatCfg<CMAX, STEREO>::RR])
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RPO = Cfg::RPO, RR = Cfg::RR;
    const int tid = threadIdx.x;
    const int ncol = 6 * C;
    const int a = fv.anchor[(size_t)b * fv.fmax + j];
    const int rows = feat_phase1<CMAX, STEREO>(fv, op, b, j, C, sh);
    const int nobs = sh.nobs;
    feat_phase2<CMAX, STEREO>(sh, rows);

    // ---- phase 3: every lane builds its column and applies the reflectors ----------------
    const int c = tid;
    const int s = c / 6, comp = c % 6;
    const bool is_col = c < ncol, is_res = c == ncol;
    const bool anchor_col = is_col && (s == a);
#pragma unroll
    for (int o = 0; o < CMAX; ++o) {
#pragma unroll
        for (int t = 0; t < RPO; ++t) {
            double val = 0.0;
            if (o < nobs) {
                const int so = sh.slot[o];
                if (is_res) val = sh.res[o][t];
                else if (is_col) {
                    if (op.selected_variant && anchor_col) {               // SwMargUpdate.cpp:302 (Q10)
                        if (comp < 3 && so != a) val = -sh.GX[o][t][comp];
                    } else if (s == so) {
                        if (comp < 3) { if (so != a) val = sh.GX[o][t][comp]; }      // :478
                        else val = -sh.G[o][t][comp - 3];                             // :482
                    } else if (anchor_col && comp < 3) val = -sh.GX[o][t][comp];      // :479
                }
            }
            B[RPO * o + t] = val;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double w = 0.0;
#pragma unroll
        for (int i = 0; i < RR; ++i) w += sh.V[k][i] * B[i];      // rows >= `rows` are zero in both
        w *= sh.tau[k];
#pragma unroll
        for (int i = 0; i < RR; ++i) B[i] -= w * sh.V[k][i];
    }
    return rows;
}

// ---------------------------------------------------------------------------------------------
// K5: UpdateBase::testChiSquared -> whitenResidual (Update.cpp:104-124, 36-56) with
// getMarginalCov (StateManager.cpp:128-153) gathered straight from the prior P (all features gate
// against the same prior).  One workgroup per (feature, filter).  gamma is computed by a bordered
// LDL^T elimination of [[S, r],[r^T, 0]]: the corner ends at -r^T S^-1 r.
// ---------------------------------------------------------------------------------------------
template <int CMAX, bool STEREO>
struct GateShared {
    using Cfg = FeatCfg<CMAX, STEREO>;
    FeatShared<CMAX, STEREO> f;
    double H[Cfg::NCOLMAX][Cfg::RRH];       // H_j, column-major
    double T[Cfg::NCOLMAX][Cfg::RRH];       // H_j * Pcc
    double S[(Cfg::RRH + 1) * (Cfg::RRH + 2)];
    double rj[Cfg::RRH];
};

template <int CMAX, bool STEREO>
__global__ __launch_bounds__((FeatCfg<CMAX, STEREO>::NT)) void k_msckf_gate(
    CovView cv, FrameView fv, MsckfOpts op, int b0, double* __restrict__ gamma_out, int* __restrict__ accept_out)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RR = Cfg::RR, RRH = Cfg::RRH, NT = Cfg::NT, LS = RRH + 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GateShared<CMAX, STEREO>& sh = *reinterpret_cast<GateShared<CMAX, STEREO>*>(smem_raw);

    const int b = b0 + blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    if (j >= fv.n_feat[b]) return;
    const int C = fv.n_clones[b], ncol = 6 * C, ld = cv.ldp;
    const double* P = cov_ptr(cv, b);
    load_gidx<CMAX, STEREO>(fv, b, C, sh.f);

    double B[RR];
    const int rows = build_feature<CMAX, STEREO>(fv, op, b, j, C, sh.f, B);
    const int rho = rows - 3;
    const size_t oidx = (size_t)b * fv.fmax + j;
    if (rho <= 0) {                        // uniform
        if (tid == 0) { gamma_out[oidx] = __builtin_nan(""); accept_out[oidx] = 0; }
        return;
    }
    if (tid < ncol) {
#pragma unroll
        for (int i = 0; i < RRH; ++i) sh.H[tid][i] = (i < rho) ? B[3 + i] : 0.0;
    } else if (tid == ncol) {
#pragma unroll
        for (int i = 0; i < RRH; ++i) sh.rj[i] = (i < rho) ? B[3 + i] : 0.0;
    }
    __syncthreads();
    // T = H_j * Pcc : lane c accumulates column c; Pcc[k][c] read as P[gidx[c], gidx[k]] (symmetric)
    if (tid < ncol) {
        double acc[RRH];
#pragma unroll
        for (int i = 0; i < RRH; ++i) acc[i] = 0.0;
        const int gc = sh.f.gidx[tid];
        for (int k = 0; k < ncol; ++k) {
            const double p = P[gc + (size_t)sh.f.gidx[k] * ld];
#pragma unroll
            for (int i = 0; i < RRH; ++i) acc[i] += sh.H[k][i] * p;
        }
#pragma unroll
        for (int i = 0; i < RRH; ++i) sh.T[tid][i] = acc[i];
    }
    __syncthreads();
    // S = T H_j^T + sigma^2 I (lower triangle), bordered with r_j
    for (int e = tid; e < rho * rho; e += NT) {
        const int i = e % rho, i2 = e / rho;
        if (i < i2) continue;
        double acc = 0.0;
        for (int c = 0; c < ncol; ++c) acc += sh.T[c][i] * sh.H[c][i2];
        sh.S[i * LS + i2] = acc + (i == i2 ? op.var : 0.0);
    }
    for (int e = tid; e <= rho; e += NT) sh.S[rho * LS + e] = (e < rho) ? sh.rj[e] : 0.0;
    __syncthreads();
    // bordered elimination: S[i][k] -= S[i][j] S[k][j] / S[j][j]  for j < k <= i <= rho
    for (int jj = 0; jj < rho; ++jj) {
        const double inv = 1.0 / sh.S[jj * LS + jj];
        const int w = rho - jj;                    // remaining rows jj+1 .. rho
        for (int e = tid; e < w * w; e += NT) {
            const int i = jj + 1 + e % w, k = jj + 1 + e / w;
            if (k > i) continue;
            sh.S[i * LS + k] -= sh.S[i * LS + jj] * sh.S[k * LS + jj] * inv;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double g = -sh.S[rho * LS + rho];
        const int dof = fv.dof[oidx];
        const bool ok = dof >= 1 && dof < op.chi2_len && g < op.chi2[dof];      // Update.cpp:120
        gamma_out[oidx] = g;
        accept_out[oidx] = ok ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// K6 + K7: TSQR.  Packed upper-trapezoid R in LDS: row k holds columns k..ncol (ncol = rhs).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int roff(int k, int ncol) { return k * (ncol + 1) - (k * (k - 1)) / 2; }

template <int RR>
struct FoldShared {
    double v[2][RR];
    double tau[2];
};

// Folds the column-owner block B (all RR register rows; unused rows must be zero) into the packed
// R (Householder "append rows" update, one barrier per column).  All NT lanes must call.
template <int RR>
__device__ __forceinline__ void fold_rows(double (&B)[RR], int ncol, double* sR, FoldShared<RR>& fs, int k0 = 0)
{
    const int tid = threadIdx.x;
    for (int k = k0; k < ncol; ++k) {     // columns < k0 of the block are known to be zero
        const int buf = k & 1;
        const int rk = roff(k, ncol);
        if (tid == k) {
            double nrm2 = 0.0;
#pragma unroll
            for (int i = 0; i < RR; ++i) nrm2 += B[i] * B[i];
            double tau = 0.0;
            if (nrm2 > 0.0) {
                const double x0 = sR[rk];
                const double nrm = sqrt(x0 * x0 + nrm2);
                const double alpha = x0 >= 0.0 ? -nrm : nrm;
                const double v0 = x0 - alpha;
                tau = -v0 / alpha;
                const double iv0 = 1.0 / v0;
#pragma unroll
                for (int i = 0; i < RR; ++i) { fs.v[buf][i] = B[i] * iv0; B[i] = 0.0; }
                sR[rk] = alpha;
            }
            fs.tau[buf] = tau;
        }
        __syncthreads();
        const double tau = fs.tau[buf];
        if (tau != 0.0 && tid > k && tid <= ncol) {
            double w = sR[rk + tid - k];
#pragma unroll
            for (int i = 0; i < RR; ++i) w += fs.v[buf][i] * B[i];
            w *= tau;
            sR[rk + tid - k] -= w;
#pragma unroll
            for (int i = 0; i < RR; ++i) B[i] -= w * fs.v[buf][i];
        }
    }
    __syncthreads();
}

// grid = (G chunks, nb).  Chunk g folds the used features j = g, g+G, ... of filter b and writes
// its n x (n+1) partial factor (dense, row-major, zeros below the diagonal).  Chunk 0 also
// publishes the final accepted mask (chi^2 accept AND rank among accepted < max_accept,
// RemoveLostUpdate.cpp:357-359) and the per-filter counts.
template <int CMAX, bool STEREO>
struct FoldKShared {
    using Cfg = FeatCfg<CMAX, STEREO>;
    FeatShared<CMAX, STEREO> f;
    FoldShared<Cfg::RR> fs;
    int nused_chunk;
    // followed by: int use[fmax] ; double R[packed]
};

template <int CMAX, bool STEREO>
__global__ __launch_bounds__((FeatCfg<CMAX, STEREO>::NT)) void k_msckf_fold(
    FrameView fv, MsckfOpts op, int b0, const int* __restrict__ accept_in, int* __restrict__ used_out,
    double* __restrict__ Rpart, int* __restrict__ chunk_used, int G, int rstride)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RR = Cfg::RR, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    FoldKShared<CMAX, STEREO>& sh = *reinterpret_cast<FoldKShared<CMAX, STEREO>*>(smem_raw);
    const int bl = blockIdx.y, b = b0 + bl, g = blockIdx.x, tid = threadIdx.x;
    const int F = fv.n_feat[b], C = fv.n_clones[b], ncol = 6 * C;
    int* sUse = reinterpret_cast<int*>(smem_raw + ((sizeof(FoldKShared<CMAX, STEREO>) + 15) / 16) * 16);
    double* sR = reinterpret_cast<double*>(reinterpret_cast<char*>(sUse) + ((sizeof(int) * fv.fmax + 15) / 16) * 16);

    // accepted-feature cap: use[j] = accept[j] && #accepted before j < max_accept
    for (int j = tid; j < F; j += NT) {
        int use = accept_in[(size_t)b * fv.fmax + j];
        if (use && op.max_accept > 0) {
            int rank = 0;
            for (int q = 0; q < j; ++q) rank += accept_in[(size_t)b * fv.fmax + q];
            if (rank >= op.max_accept) use = 0;
        }
        sUse[j] = use;
        if (g == 0) used_out[(size_t)b * fv.fmax + j] = use;
    }
    const int psize = roff(ncol, ncol);
    for (int e = tid; e < psize; e += NT) sR[e] = 0.0;
    if (tid == 0) sh.nused_chunk = 0;
    __syncthreads();

    double B[RR];
    int nused = 0;
    for (int j = g; j < F; j += G) {
        if (!sUse[j]) continue;                                  // uniform
        build_feature<CMAX, STEREO>(fv, op, b, j, C, sh.f, B);
        B[0] = 0.0; B[1] = 0.0; B[2] = 0.0;                      // rows 0..2 span range(Hf): projected out
        fold_rows<RR>(B, ncol, sR, sh.fs);
        ++nused;
    }
    double* out = Rpart + ((size_t)bl * G + g) * rstride;
    for (int k = 0; k < ncol; ++k) {
        if (tid <= ncol) out[(size_t)k * (ncol + 1) + tid] = (tid >= k) ? sR[roff(k, ncol) + tid - k] : 0.0;
    }
    if (tid == 0) chunk_used[bl * G + g] = nused;
}

// grid = nb.  Merges the G partial factors of a filter (row blocks of RR rows folded into
// partial 0), then writes H_thin (n x n upper triangular, column-major, ld = mld), r_thin, the
// column map and m = n for the Kalman-update kernel (m = 0 if nothing was accepted).
template <int CMAX, bool STEREO>
__global__ __launch_bounds__((FeatCfg<CMAX, STEREO>::NT)) void k_msckf_merge(
    FrameView fv, int b0, const double* __restrict__ Rpart, const int* __restrict__ chunk_used, int G, int rstride,
    double* __restrict__ Hout, double* __restrict__ res_out, int* __restrict__ colmap, int* __restrict__ m_out,
    int* __restrict__ nc_out, int mld, int hstride, int cstride)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RR = Cfg::RR, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    FoldShared<RR>& fs = *reinterpret_cast<FoldShared<RR>*>(smem_raw);
    double* sR = reinterpret_cast<double*>(smem_raw + ((sizeof(FoldShared<RR>) + 15) / 16) * 16);
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int C = fv.n_clones[b], ncol = 6 * C;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        if (tid == 0) { m_out[bl] = 0; nc_out[bl] = ncol; }
        return;
    }
    const int psize = roff(ncol, ncol);
    for (int e = tid; e < psize; e += NT) sR[e] = 0.0;
    __syncthreads();
    double B[RR];
    bool first = true;
    for (int g = 0; g < G; ++g) {
        if (chunk_used[bl * G + g] == 0) continue;
        const double* part = Rpart + ((size_t)bl * G + g) * rstride;
        if (first) {
            for (int k = 0; k < ncol; ++k)
                if (tid >= k && tid <= ncol) sR[roff(k, ncol) + tid - k] = part[(size_t)k * (ncol + 1) + tid];
            first = false;
            __syncthreads();
            continue;
        }
        for (int rb = 0; rb < ncol; rb += RR) {
            const int cnt = min(RR, ncol - rb);
#pragma unroll
            for (int i = 0; i < RR; ++i)
                B[i] = (i < cnt && tid <= ncol) ? part[(size_t)(rb + i) * (ncol + 1) + tid] : 0.0;
            fold_rows<RR>(B, ncol, sR, fs, rb);                  // partial factors are upper triangular
        }
    }
    double* H = Hout + (size_t)bl * hstride;
    for (int k = 0; k < ncol; ++k) {
        if (tid < ncol) H[k + (size_t)tid * mld] = (tid >= k) ? sR[roff(k, ncol) + tid - k] : 0.0;
        if (tid == ncol) res_out[(size_t)bl * mld + k] = sR[roff(k, ncol) + ncol - k];
    }
    if (tid < ncol) colmap[(size_t)bl * cstride + tid] = fv.clone_idx[(size_t)b * fv.cmax + tid / 6] + tid % 6;
    if (tid == 0) { m_out[bl] = ncol; nc_out[bl] = ncol; }
}

// Dense TSQR leaf for ingvio_qr_compress (the SPQR call sites on an explicit H_large): chunk g
// folds row blocks g, g+G, ... of the m x ncol matrix [H | res] (column-major, ld = ldh).
template <int CMAX, bool STEREO>
__global__ __launch_bounds__((FeatCfg<CMAX, STEREO>::NT)) void k_fold_dense(
    const double* __restrict__ H, const double* __restrict__ res, int ldh, int m, int ncol,
    double* __restrict__ Rpart, int* __restrict__ chunk_used, int G, int rstride)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    constexpr int RR = Cfg::RR, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    FoldShared<RR>& fs = *reinterpret_cast<FoldShared<RR>*>(smem_raw);
    double* sR = reinterpret_cast<double*>(smem_raw + ((sizeof(FoldShared<RR>) + 15) / 16) * 16);
    const int g = blockIdx.x, tid = threadIdx.x;
    const int psize = roff(ncol, ncol);
    for (int e = tid; e < psize; e += NT) sR[e] = 0.0;
    __syncthreads();
    double B[RR];
    int nblk = 0;
    const int nrb = (m + RR - 1) / RR;
    for (int rbi = g; rbi < nrb; rbi += G) {
        const int rb = rbi * RR, cnt = min(RR, m - rb);
#pragma unroll
        for (int i = 0; i < RR; ++i) {
            double val = 0.0;
            if (i < cnt) {
                if (tid < ncol) val = H[(size_t)(rb + i) + (size_t)tid * ldh];
                else if (tid == ncol) val = res[rb + i];
            }
            B[i] = val;
        }
        fold_rows<RR>(B, ncol, sR, fs);
        ++nblk;
    }
    double* out = Rpart + (size_t)g * rstride;
    for (int k = 0; k < ncol; ++k)
        if (tid <= ncol) out[(size_t)k * (ncol + 1) + tid] = (tid >= k) ? sR[roff(k, ncol) + tid - k] : 0.0;
    if (tid == 0) chunk_used[g] = nblk;
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (explicit instantiations live here so the C ABI file stays template-free)
// ---------------------------------------------------------------------------------------------
#include "launch_msckf.h"

template <int CMAX, bool STEREO>
static void launch_t(const MsckfLaunch& L, hipStream_t st)
{
    using Cfg = FeatCfg<CMAX, STEREO>;
    const int ncolmax = 6 * L.fv.cmax;      // runtime C <= cmax <= CMAX
    (void)ncolmax;
    if (L.stage == 0) {
        const size_t sm = sizeof(GateShared<CMAX, STEREO>);
        hipFuncSetAttribute((const void*)k_msckf_gate<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL((k_msckf_gate<CMAX, STEREO>), dim3(L.fmax_used, L.nb), dim3(Cfg::NT), sm, st,
                           L.cv, L.fv, L.op, L.b0, L.gamma, L.accept);
    } else if (L.stage == 1) {
        const int nc = 6 * CMAX;
        const size_t sm = ((sizeof(FoldKShared<CMAX, STEREO>) + 15) / 16) * 16 + ((sizeof(int) * L.fv.fmax + 15) / 16) * 16 +
                          sizeof(double) * (size_t)(nc * (nc + 1) - (nc * (nc - 1)) / 2);
        hipFuncSetAttribute((const void*)k_msckf_fold<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL((k_msckf_fold<CMAX, STEREO>), dim3(L.G, L.nb), dim3(Cfg::NT), sm, st,
                           L.fv, L.op, L.b0, L.accept, L.used, L.Rpart, L.chunk_used, L.G, L.rstride);
    } else if (L.stage == 2) {
        const int nc = 6 * CMAX;
        const size_t sm = ((sizeof(FoldShared<Cfg::RR>) + 15) / 16) * 16 +
                          sizeof(double) * (size_t)(nc * (nc + 1) - (nc * (nc - 1)) / 2);
        hipFuncSetAttribute((const void*)k_msckf_merge<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL((k_msckf_merge<CMAX, STEREO>), dim3(L.nb), dim3(Cfg::NT), sm, st,
                           L.fv, L.b0, L.Rpart, L.chunk_used, L.G, L.rstride, L.Hout, L.res_out, L.colmap, L.m_out,
                           L.nc_out, L.mld, L.hstride, L.cstride);
    } else {
        const int nc = 6 * CMAX;
        const size_t sm = ((sizeof(FoldShared<Cfg::RR>) + 15) / 16) * 16 +
                          sizeof(double) * (size_t)(nc * (nc + 1) - (nc * (nc - 1)) / 2);
        hipFuncSetAttribute((const void*)k_fold_dense<CMAX, STEREO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL((k_fold_dense<CMAX, STEREO>), dim3(L.G), dim3(Cfg::NT), sm, st,
                           L.dH, L.dres, L.ldh, L.m, L.ncol, L.Rpart, L.chunk_used, L.G, L.rstride);
    }
}

int msckf_cmax_class(int cmax) { return cmax <= 6 ? 6 : (cmax <= 11 ? 11 : (cmax <= 16 ? 16 : -1)); }

int launch_msckf(const MsckfLaunch& L, hipStream_t st)
{
    const int cls = msckf_cmax_class(L.fv.cmax);
    if (cls < 0) return -1;
#define DISPATCH(CM)                                                         \
    if (cls == CM) { if (L.stereo) launch_t<CM, true>(L, st); else launch_t<CM, false>(L, st); return 0; }
    DISPATCH(6)
    DISPATCH(11)
    DISPATCH(16)
#undef DISPATCH
    return -1;
}
