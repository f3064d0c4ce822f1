// kernels_ekf.hip — K8-K11: StateManager::ekfUpdate (StateManager.cpp:359-423) and
// UpdateBase::whitenResidual (Update.cpp:36-79).
//
//   k_ekf_core  (one workgroup per filter):  PH^T = P[:, cols] H^T  (K8, :381-397)
//                                            S = H Pcc H^T + R      (K9, :399-403) in LDS
//                                            Cholesky S = L L^T, Y = PH^T L^-T, z = L^-1 res
//                                            dx = Y z (= K res, :423)
//   k_downdate  (MFMA FP64 16x16x4 tiles):   P <- P - Y Y^T  (= P - K (PH^T)^T, :407-411), lower
//                                            tiles computed once and mirrored, which is the
//                                            reference's 0.5 (P + P^T) without a second pass.
// S is SPD (R is positive definite), so Cholesky + triangular solves give the same K as the
// reference's general LU `S.inverse()`.  gfx950 only.
#include "dev_common.h"
#include "launch_ekf.h"
#include "block64.h"

#define EKF_THREADS 512
#define IB 16

__global__ __launch_bounds__(EKF_THREADS) void k_ekf_core(
    CovView cv, int b0, const double* __restrict__ Hall, const double* __restrict__ res_all,
    const int* __restrict__ colmap_all, int* __restrict__ m_all, const int* __restrict__ nc_all,
    const double* __restrict__ noise_all, int r_kind, int mld, int hstride, int cstride, int nstride,
    double* __restrict__ Yall, int ystride, double* __restrict__ dx_all, int* __restrict__ status,
    const double* __restrict__ chi2, int chi2_len, int gate_max_rows,
    const double* __restrict__ Wall, size_t wstride, int ypad, const int* __restrict__ marg_idx, int marg_size)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sS = reinterpret_cast<double*>(smem_raw);
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int m = m_all[bl], nc = nc_all[bl];
    const int n = cv.n[b], ld = cv.ldp;
    double* dx = dx_all + (size_t)b * ld;
    if (m == 0) {
        for (int r = tid; r < ld; r += EKF_THREADS) dx[r] = 0.0;      // the whole row: callers fetch ldp entries per filter
        return;
    }
    const double* P = cov_ptr(cv, b);
    const double* W = Wall ? Wall + (size_t)bl * wstride : nullptr;        // in-frame update: the var_order columns of the posterior to be
    const int midx = marg_idx ? marg_idx[bl] : -1;                         // dx after the frame's marginalisation
    const double* H = Hall + (size_t)bl * hstride;
    const double* res = res_all + (size_t)bl * mld;
    const int* cm = colmap_all + (size_t)bl * cstride;
    const double* noise = noise_all + (size_t)bl * nstride;
    double* Y = Yall + (size_t)bl * ystride;
    const int LS = m + 1;
    int* sCol = reinterpret_cast<int*>(sS + (size_t)(m + 1) * LS);

    for (int c = tid; c < nc; c += EKF_THREADS) sCol[c] = cm[c];
    __syncthreads();
    // ---- K8: PH^T -> Y (n x m, column-major, ld) ------------------------------------------
    // Small updates (the in-frame GNSS update: 16 rows, 15 columns) used to walk these loops one dependent global round trip at a time -
    // a load of P (or W) and IB of H per column here, two loads per column for S, and in the substitution at the end every x_k read
    // back from the Y it had just been stored to (m^2 / 2 serialized loads per state row): 42 us per 512 filters for 0.1 MFLOP each.
    // With nc <= 16 / m <= 16 a thread's operands are requested together and kept in registers; the sums run in the same order
    // (bit-identical results).  Larger updates take the loops as they were.
    const bool small_c = nc <= 16, small_m = m <= 16;
    for (int r = tid; r < n; r += EKF_THREADS) {
        double pr[16];
        if (small_c) {
#pragma unroll
            for (int c = 0; c < 16; ++c) { const int cc = c < nc ? c : 0; pr[c] = W ? W[r + (size_t)cc * ld] : P[r + (size_t)sCol[cc] * ld]; }
        }
        for (int ib = 0; ib < m; ib += IB) {
            double acc[IB];
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) acc[ii] = 0.0;
            if (small_c) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c < nc) {
                        const double* hc = H + (size_t)c * mld + ib;
#pragma unroll
                        for (int ii = 0; ii < IB; ++ii) acc[ii] += pr[c] * hc[ii];
                    }
                }
            } else
            for (int c = 0; c < nc; ++c) {
                const double p = W ? W[r + (size_t)c * ld] : P[r + (size_t)sCol[c] * ld];
                const double* hc = H + (size_t)c * mld + ib;      // rows ib.. of column c (zero padded to mld)
#pragma unroll
                for (int ii = 0; ii < IB; ++ii) acc[ii] += p * hc[ii];
            }
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) if (ib + ii < m) Y[r + (size_t)(ib + ii) * ld] = acc[ii];
        }
    }
    __syncthreads();
    // ---- K9: S = H * PHT[cols, :] + R, lower triangle, bordered by res ---------------------
    for (int e = tid; e < m * m; e += EKF_THREADS) {
        const int i = e % m, i2 = e / m;
        if (i < i2) continue;
        double acc = 0.0;
        if (small_c) {
            double hv[16], yv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) { const int cc = c < nc ? c : 0; hv[c] = H[i + (size_t)cc * mld]; yv[c] = Y[sCol[cc] + (size_t)i2 * ld]; }
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c < nc) acc += hv[c] * yv[c];
        } else
        for (int c = 0; c < nc; ++c) acc += H[i + (size_t)c * mld] * Y[sCol[c] + (size_t)i2 * ld];
        if (r_kind == 0) { if (i == i2) acc += noise[0]; }
        else if (r_kind == 1) { if (i == i2) acc += noise[i]; }
        else acc += noise[i + (size_t)i2 * m];
        sS[i * LS + i2] = acc;
    }
    for (int e = tid; e <= m; e += EKF_THREADS) sS[m * LS + e] = (e < m) ? res[e] : 0.0;
    __syncthreads();
    // ---- right-looking elimination (unscaled columns), one barrier per column --------------
    for (int j = 0; j < m; ++j) {
        const double inv = 1.0 / sS[j * LS + j];
        const int w = m - j;
        for (int e = tid; e < w * w; e += EKF_THREADS) {
            const int i = j + 1 + e % w, k = j + 1 + e / w;
            if (k > i) continue;
            sS[i * LS + k] -= sS[i * LS + j] * sS[k * LS + j] * inv;
        }
        __syncthreads();
    }
    // scale: L[i][j] = S^(j)[i][j] / sqrt(S^(j)[j][j]);  row m becomes z = L^-1 res
    bool bad = false;
    for (int e = tid; e < (m + 1) * m; e += EKF_THREADS) {
        const int i = e / m, j = e % m;
        if (i <= j) continue;
        const double d = sS[j * LS + j];
        if (!(d > 0.0)) bad = true;
        sS[i * LS + j] = sS[i * LS + j] / sqrt(d);
    }
    __syncthreads();
    for (int j = tid; j < m; j += EKF_THREADS) sS[j * LS + j] = sqrt(sS[j * LS + j]);
    __syncthreads();
    if (__syncthreads_or(bad)) {                                                    // S = H P H^T + R is not positive definite:
        for (int r = tid; r < ld; r += EKF_THREADS) dx[r] = 0.0;                   // no update (INGVIO_E_NOT_PD), k_downdate skips the filter
        if (tid == 0) { atomicOr(&status[b], 4); if (W) m_all[bl] = 0; }            // in-frame: the write-back must not fold a gain in
        return;
    }
    // ---- optional block gate on the prior (GnssUpdate.cpp:286: `rows <= 14 && strong_reject && !testChiSquared(.., R, rows)`):
    //      gamma = res^T S^-1 res = |z|^2 is already in the border of the factorisation
    if (gate_max_rows > 0 && m <= gate_max_rows) {
        double g = 0.0;
        for (int j = 0; j < m; ++j) g += sS[m * LS + j] * sS[m * LS + j];          // uniform: every thread sums the same LDS row
        const bool pass = m < chi2_len && g < chi2[m];                              // Update.cpp:160
        if (!pass) {
            for (int r = tid; r < ld; r += EKF_THREADS) dx[r] = 0.0;
            if (tid == 0) { m_all[bl] = 0; atomicOr(&status[b], 8); }               // k_downdate sees m = 0: state untouched
            return;
        }
    }
    // ---- Y = PHT L^-T (row-wise forward substitution), dx = Y z -----------------------------
    for (int r = tid; r < n; r += EKF_THREADS) {
        double d = 0.0;
        if (small_m) {
            double y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = Y[r + (size_t)(j < m ? j : 0) * ld];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < m) {
                    double x = y[j];
#pragma unroll
                    for (int k = 0; k < j; ++k) x -= sS[j * LS + k] * y[k];
                    x /= sS[j * LS + j];
                    y[j] = x;
                    Y[r + (size_t)j * ld] = x;
                    d += x * sS[m * LS + j];
                }
            }
        } else
        for (int j = 0; j < m; ++j) {
            double x = Y[r + (size_t)j * ld];
            for (int k = 0; k < j; ++k) x -= sS[j * LS + k] * Y[r + (size_t)k * ld];
            x /= sS[j * LS + j];
            Y[r + (size_t)j * ld] = x;
            d += x * sS[m * LS + j];
        }
        if (midx < 0) dx[r] = d;
        else if (r < midx) dx[r] = d;
        else if (r >= midx + marg_size) dx[r - marg_size] = d;
        const int kp = ypad > m ? ypad : ((m + 3) & ~3);
        for (int j = m; j < kp; ++j) Y[r + (size_t)j * ld] = 0.0;                   // pad K dim for MFMA
    }
    if (midx >= 0) for (int r = n - marg_size + tid; r < ld; r += EKF_THREADS) dx[r] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// Per-row whitenResidual gates of a stacked update against the prior, then in-place compaction of the accepted rows
// (GnssUpdate.cpp:190,259: every pseudo-range / Doppler row is tested on its own, dof 1, before it joins the stack).
// Row i only has non-zero entries in the columns of its own sub_order, so h_i Pvv h_i^T over the stacked var_order
// equals the reference's 11-column sub-block product.  One workgroup per filter; m <= 256 candidate rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_gate(
    CovView cv, int b0, const double* __restrict__ Hin_all, const double* __restrict__ res_in_all, const double* __restrict__ noise_in_all,
    const int* __restrict__ m_in_all, const int* __restrict__ colmap_all, const int* __restrict__ nc_all, int in_hstride, int in_cstride,
    double* __restrict__ Hall, double* __restrict__ res_all, double* __restrict__ noise_all, int* __restrict__ m_all,
    int* __restrict__ colmap_out, int* __restrict__ nc_out, int mld, int hstride, int cstride,
    double thr, double* __restrict__ gamma_all, int* __restrict__ keep_all, const double* __restrict__ Wall, size_t wstride)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x;
    const int m = m_in_all[bl], nc = nc_all[bl], ld = cv.ldp;
    const double* W = Wall ? Wall + (size_t)bl * wstride : nullptr;
    double* sP = reinterpret_cast<double*>(smem_raw);             // nc x nc marginal
    double* sH = sP + (size_t)nc * nc;                            // m x nc candidate rows (row-major)
    __shared__ int sPos[257];
    const double* P = cov_ptr(cv, b);
    const double* Hin = Hin_all + (size_t)bl * in_hstride;
    const double* res_in = res_in_all + (size_t)bl * mld;
    const double* noise_in = noise_in_all + (size_t)bl * mld;
    const int* cm = colmap_all + (size_t)bl * in_cstride;
    double* H = Hall + (size_t)bl * hstride;
    double* res = res_all + (size_t)bl * mld;
    double* noise = noise_all + (size_t)bl * mld;
    for (int e = tid; e < nc * nc; e += 256) { const int c = e % nc, c2 = e / nc; sP[e] = W ? W[cm[c] + (size_t)c2 * ld] : P[cm[c] + (size_t)cm[c2] * ld]; }
    for (int e = tid; e < m * nc; e += 256) { const int i = e % m, c = e / m; sH[i * nc + c] = Hin[i + (size_t)c * mld]; }
    for (int c = tid; c < nc; c += 256) colmap_out[(size_t)bl * cstride + c] = cm[c];
    sPos[tid] = 0;
    if (tid == 0) { sPos[256] = 0; nc_out[bl] = nc; }
    __syncthreads();
    bool keep = false;
    double r_i = 0.0, n_i = 0.0;
    if (tid < m) {
        const double* h = sH + tid * nc;
        double s = 0.0;
        for (int c = 0; c < nc; ++c) {
            double t = 0.0;
            for (int c2 = 0; c2 < nc; ++c2) t += sP[c + c2 * nc] * h[c2];
            s += h[c] * t;
        }
        r_i = res_in[tid]; n_i = noise_in[tid];
        const double g = r_i * r_i / (s + n_i);                   // Update.cpp:36-56 with a 1 x 1 S
        keep = g < thr;                                           // :93-97, dof = res.rows() = 1 (thr = +inf: gate off)
        gamma_all[(size_t)bl * mld + tid] = g;
        keep_all[(size_t)bl * mld + tid] = keep ? 1 : 0;
        sPos[tid + 1] = keep ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) for (int i = 0; i < m; ++i) sPos[i + 1] += sPos[i];      // m <= 256: a serial scan is cheaper than its barriers
    __syncthreads();
    const int mk = sPos[m];
    if (tid < m && keep) {
        const int d = sPos[tid];
        res[d] = r_i; noise[d] = n_i;
    }
    const int mpad = (m + 15) & ~15;                                       // k_ekf_core reads whole 16-row groups: rows >= mk are zero
    for (int e = tid; e < mpad * nc; e += 256) {
        const int i = e % mpad, c = e / mpad;
        if (i < m && sPos[i + 1] != sPos[i]) H[sPos[i] + (size_t)c * mld] = sH[i * nc + c];
        if (i >= mk && i < mld) H[i + (size_t)c * mld] = 0.0;
    }
    if (tid == 0) m_all[bl] = mk;
}

// ---------------------------------------------------------------------------------------------
// K10: P <- P - Y Yb^T (Yb == Y: the symmetric Cholesky form; Yb = Pc: the information form) with
// v_mfma_f64_16x16x4_f64.  One wave per 16x16 tile of the lower
// triangle; A[i][k] = Y[ri+i][k], B[k][j] = Y[rj+j][k] are read straight from L2 (Y is N x m,
// <= 130 KB per filter).  f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg.
// ---------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_downdate(CovView cv, int b0, const double* __restrict__ Yall,
                                                  const double* __restrict__ Yball,
                                                  const int* __restrict__ m_all, size_t ystride, int* __restrict__ status, int ldy)
{
    const int bl = blockIdx.y, b = b0 + bl;
    const int m = m_all[bl];
    if (m == 0 || (status[b] & 4)) return;                               // bit 4: S not positive definite, the state stays untouched
    const int n = cv.n[b], ld = cv.ldp;
    const int nt = (n + 15) >> 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= nt * (nt + 1) / 2) return;
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    double* P = cov_ptr(cv, b);
    const double* Y = Yall + (size_t)bl * ystride;
    const double* Yb = Yball + (size_t)bl * ystride;      // == Y for the symmetric Y Y^T form
    const int ra = ti * 16 + (lane & 15), rb = tj * 16 + (lane & 15), kq = lane >> 4;
    const bool va = ra < n, vb = rb < n;
    double4_t acc = { 0.0, 0.0, 0.0, 0.0 };
    const int mp = (m + 3) & ~3;
    for (int k0 = 0; k0 < mp; k0 += 4) {
        const double a = va ? Y[ra + (size_t)(k0 + kq) * ldy] : 0.0;
        const double bb = vb ? Yb[rb + (size_t)(k0 + kq) * ldy] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
    }
    const int col = tj * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + (lane >> 4) + 4 * r;
        if (row < n && col < n && row >= col) {
            // read through the mirrored (coalesced) address; P is symmetric on entry
            const double v = P[col + (size_t)row * ld] - acc[r];
            P[col + (size_t)row * ld] = v;
            P[row + (size_t)col * ld] = v;
            if (row == col && v < 0.0) atomicOr(&status[b], 2);          // StateManager.cpp:413-421
        }
    }
}

// whitenResidual for small blocks (GNSS per-row / landmark / block gating): one workgroup per block.
__device__ __forceinline__ void gamma_body(const CovView& cv, int b, const double* __restrict__ H, const double* __restrict__ res,
                                           const int* __restrict__ colmap, int m, int nc, const double* __restrict__ noise, int r_kind,
                                           int mld, double* __restrict__ gamma_out, double* sT)
{
    double* sS = sT + (size_t)nc * m;                           // sT: nc x m = Pcc H^T ; sS: (m+1) x (m+1)
    const int tid = threadIdx.x, ld = cv.ldp, LS = m + 1;
    const double* P = cov_ptr(cv, b);
    for (int e = tid; e < nc * m; e += 256) {
        const int c = e % nc, i = e / nc;
        double acc = 0.0;
        for (int c2 = 0; c2 < nc; ++c2) acc += P[colmap[c] + (size_t)colmap[c2] * ld] * H[i + (size_t)c2 * mld];
        sT[c + (size_t)i * nc] = acc;
    }
    __syncthreads();
    for (int e = tid; e < m * m; e += 256) {
        const int i = e % m, i2 = e / m;
        if (i < i2) continue;
        double acc = 0.0;
        for (int c = 0; c < nc; ++c) acc += H[i + (size_t)c * mld] * sT[c + (size_t)i2 * nc];
        if (r_kind == 0) { if (i == i2) acc += noise[0]; }
        else if (r_kind == 1) { if (i == i2) acc += noise[i]; }
        else acc += noise[i + (size_t)i2 * m];
        sS[i * LS + i2] = acc;
    }
    for (int e = tid; e <= m; e += 256) sS[m * LS + e] = (e < m) ? res[e] : 0.0;
    __syncthreads();
    for (int j = 0; j < m; ++j) {
        const double inv = 1.0 / sS[j * LS + j];
        const int w = m - j;
        for (int e = tid; e < w * w; e += 256) {
            const int i = j + 1 + e % w, k = j + 1 + e / w;
            if (k > i) continue;
            sS[i * LS + k] -= sS[i * LS + j] * sS[k * LS + j] * inv;
        }
        __syncthreads();
    }
    if (tid == 0) *gamma_out = -sS[m * LS + m];
}

__global__ __launch_bounds__(256) void k_gamma(CovView cv, int b, const double* __restrict__ H, const double* __restrict__ res,
                                               const int* __restrict__ colmap, int m, int nc, const double* __restrict__ noise,
                                               int r_kind, int mld, double* __restrict__ gamma_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    gamma_body(cv, b, H, res, colmap, m, nc, noise, r_kind, mld, gamma_out, reinterpret_cast<double*>(smem_raw));
}

// Many independent gates against the same prior in one launch (all landmarks of a frame, all GNSS rows): block g reads
// its descriptor {offset of H, of res, of colmap (in ints), m, nc} from desc[5 g ..]; H is m x nc, tight (ld = m).
__global__ __launch_bounds__(256) void k_gamma_multi(CovView cv, int b, const double* __restrict__ dbuf, const int* __restrict__ ibuf,
                                                     const int* __restrict__ desc, const double* __restrict__ noise,
                                                     double* __restrict__ gamma_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int* d = desc + 5 * blockIdx.x;
    gamma_body(cv, b, dbuf + d[0], dbuf + d[1], ibuf + d[2], d[3], d[4], noise, 0, d[3], gamma_out + blockIdx.x,
               reinterpret_cast<double*>(smem_raw));
}

void launch_ekf_core(const EkfLaunch& L, hipStream_t st)
{
    const size_t sm = sizeof(double) * (size_t)(L.m_cap + 1) * (L.m_cap + 1) + sizeof(int) * (size_t)L.nc_cap + 16;
    hipFuncSetAttribute((const void*)k_ekf_core, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(k_ekf_core, dim3(L.nb), dim3(EKF_THREADS), sm, st, L.cv, L.b0, L.H, L.res, L.colmap, L.m, L.nc,
                       L.noise, L.r_kind, L.mld, L.hstride, L.cstride, L.nstride, L.Y, L.ystride, L.dx, L.status,
                       L.chi2, L.chi2_len, L.gate_max_rows, L.W, L.wstride, L.ypad, L.marg_idx, L.marg_size);
}

// per-row gates + compaction from the staged (pristine) rows into the working rows; returns non-zero when they do not fit in LDS
int launch_rows_gate(const EkfLaunch& L, const RowsGateIn& in, double thr, double* gamma, int* keep, hipStream_t st)
{
    const size_t sm = sizeof(double) * ((size_t)L.nc_cap * L.nc_cap + (size_t)L.m_cap * L.nc_cap);
    if (L.m_cap > 256 || sm > 150 * 1024) return -1;
    hipFuncSetAttribute((const void*)k_rows_gate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(k_rows_gate, dim3(L.nb), dim3(256), sm, st, L.cv, L.b0, in.H, in.res, in.noise, in.m, in.colmap, in.nc,
                       in.hstride, in.cstride, const_cast<double*>(L.H), const_cast<double*>(L.res), const_cast<double*>(L.noise), L.m,
                       const_cast<int*>(L.colmap), const_cast<int*>(L.nc), L.mld, L.hstride, L.cstride, thr, gamma, keep, L.W, L.wstride);
    return 0;
}
// ---------------------------------------------------------------------------------------------
// The same downdate for many rows (m >= 64: the stacked landmark update, generic updates through the dense-H route): one
// workgroup per 64 x 64 block of the lower triangle, the two 64-row panels of Y staged through LDS 16 columns at a time, so
// every element of Y is read once per block row/column pair instead of once per 16 x 16 tile (4x less L2 traffic).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_downdate64(CovView cv, int b0, const double* __restrict__ Yall, const int* __restrict__ m_all,
                                                    size_t ystride, int* __restrict__ status, int ldy, const int* __restrict__ marg_idx, int marg_size, int nb, int nblk)
{
    __shared__ Block64Lds sAB;
    __shared__ double sV[4][32][33];
    // workgroup -> (filter, block) so that the blocks of one filter run on ONE XCD (workgroups are dealt round-robin to the 8 XCDs):
    // its rows of Y are then fetched from HBM once and served to the other blocks by that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bl = (slot / nblk) * 8 + xcd, blk = slot % nblk;
    if (bl >= nb) return;
    const int b = b0 + bl, m = m_all[bl];
    // fused marginalisation (marg_idx[bl] >= 0): the updated covariance is written into the OTHER ping-pong half without the
    // marg_size rows / columns at marg_idx (k_post_marg flips the halves afterwards) - the frame's landmark update is followed
    // by the marginalisation of the oldest clone (IngvioFilter.cpp:296-322), a full extra read + write of every covariance when
    // it runs as a kernel of its own.  Then a filter without rows (or with a failed solve) is still copied.
    const int midx = marg_idx ? marg_idx[bl] : -1;
    const bool fuse = midx >= 0;
    const bool upd = m > 0 && !(status[b] & 4);                          // bit 4: S not positive definite, the state stays untouched
    if (!upd && !fuse) return;
    const int n = cv.n[b], ld = cv.ldp;
    int t = blk, bi = 0;
    while (t >= bi + 1) { t -= bi + 1; ++bi; }
    const int bj = t;
    if (64 * bi >= n) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = wave >> 1, wj = wave & 1;
    const bool quad_on = !(bi == bj && wj > wi);                         // the upper quadrant of a diagonal block comes from the mirror
    const double* P = cov_ptr(cv, b);
    double* D = fuse ? cov_alt_ptr(cv, b) : cov_ptr(cv, b);
    const double* Y = Yall + (size_t)bl * ystride;
    // this lane's elements of the prior block, requested now: they arrive under the MFMA loop (a read-modify-write that starts its
    // loads after the loop leaves the matrix cores idle for a memory round trip per workgroup)
    const int r0 = 64 * bi + 32 * wi, q0c = 64 * bj + 32 * wj;
    double pold[16];
    if (quad_on) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = lane + 64 * u, row = r0 + (e & 31), col = q0c + (e >> 5);
            pold[u] = (row < n && col < n && row >= col) ? NT_LOAD(&P[(size_t)row + (size_t)col * ld]) : 0.0;      // read once: streaming
        }
    }
    b64_d4 c[4];
    block64_mma(sAB, upd ? (m + 15) & ~15 : 0,
                [&](int r, int k) { return k < m ? Y[(size_t)min(64 * bi + r, n - 1) + (size_t)k * ldy] : 0.0; },
                [&](int r, int k) { return k < m ? Y[(size_t)min(64 * bj + r, n - 1) + (size_t)k * ldy] : 0.0; }, quad_on, c);
    if (!quad_on) return;                                                // no barrier below
    // this wave's 32 x 32 quadrant through LDS, then row-fast read-modify-write of P and its mirror
    block64_to_lds(c, sV[wave]);
    const int mhi = fuse ? midx + marg_size : 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = lane + 64 * u, rr = e & 31, cc = e >> 5;
        const int row = r0 + rr, col = q0c + cc;
        if (row < n && col < n && row >= col) {
            const double v = pold[u] - sV[wave][rr][cc];
            sV[wave][rr][cc] = v;
            if (row == col && v < 0.0) atomicOr(&status[b], 2);
            if (!fuse) NT_STORE(&D[(size_t)row + (size_t)col * ld], v);                     // the posterior: streaming stores (see k_info_apply)
            else if ((row < midx || row >= mhi) && (col < midx || col >= mhi))
                NT_STORE(&D[(size_t)(row < midx ? row : row - marg_size) + (size_t)(col < midx ? col : col - marg_size) * ld], v);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < 1024; e += 64) {
        const int cc = e & 31, rr = e >> 5;
        const int row = r0 + rr, col = q0c + cc;
        if (row < n && col < n && row > col) {
            if (!fuse) NT_STORE(&D[(size_t)col + (size_t)row * ld], sV[wave][rr][cc]);
            else if ((row < midx || row >= mhi) && (col < midx || col >= mhi))
                NT_STORE(&D[(size_t)(col < midx ? col : col - marg_size) + (size_t)(row < midx ? row : row - marg_size) * ld], sV[wave][rr][cc]);
        }
    }
}

bool launch_downdate(const EkfLaunch& L, int n_cap, hipStream_t st, const double* Yb, int ldy, size_t ystride, const int* marg_idx, int marg_size)
{
    const int nt = (n_cap + 15) / 16;
    const int tiles = nt * (nt + 1) / 2;
    if (ldy && L.m_cap >= 64 && !Yb) {                                     // the dense-H route with many rows: the LDS-blocked variant
        const int nb64 = (n_cap + 63) / 64;
        const int nblk = nb64 * (nb64 + 1) / 2;
        hipLaunchKernelGGL(k_downdate64, dim3((L.nb + 7) / 8 * 8 * nblk), dim3(256), 0, st, L.cv, L.b0, L.Y, L.m,
                           ystride ? ystride : (size_t)L.ystride, L.status, ldy ? ldy : L.cv.ldp, marg_idx, marg_size, L.nb, nblk);
        return marg_idx != nullptr;
    }
    hipLaunchKernelGGL(k_downdate, dim3((tiles + 3) / 4, L.nb), dim3(256), 0, st, L.cv, L.b0, L.Y, Yb ? Yb : L.Y, L.m, ystride ? ystride : (size_t)L.ystride, L.status, ldy ? ldy : L.cv.ldp);
    return false;
}

void launch_gamma_multi(CovView cv, int b, int nblk, const double* dbuf, const int* ibuf, const int* desc, const double* noise,
                        double* gamma_out, size_t lds_bytes, hipStream_t st)
{
    hipFuncSetAttribute((const void*)k_gamma_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(k_gamma_multi, dim3(nblk), dim3(256), lds_bytes, st, cv, b, dbuf, ibuf, desc, noise, gamma_out);
}

void launch_gamma(CovView cv, int b, const double* H, const double* res, const int* colmap, int m, int nc,
                  const double* noise, int r_kind, int mld, double* gamma_out, hipStream_t st)
{
    const size_t sm = sizeof(double) * ((size_t)nc * m + (size_t)(m + 1) * (m + 1));
    hipFuncSetAttribute((const void*)k_gamma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL(k_gamma, dim3(1), dim3(256), sm, st, cv, b, H, res, colmap, m, nc, noise, r_kind, mld, gamma_out);
}
