// kernels_qr.hip — general stacked-QR compression (K7) for shapes beyond the 96-column TSQR of kernels_msckf.hip:
// H (m x n) | res  ->  H_thin (n x n upper triangular) | r_thin with H_thin^T H_thin = H^T H, H_thin^T r_thin = H^T res
// (the SPQR call sites RemoveLostUpdate.cpp:376-397, SwMargUpdate.cpp:336-357, KeyframeUpdate.cpp:707-728, and the
// BASELINE "stress" shape 6000 x 800).  Blocked Householder QR with panels of QR_NB = 8 columns, no Q is ever formed.
// The working copy D is PANEL-major: column panel p is a contiguous [row][8] array, so
//   * the panel kernel's row-owner threads read 64 contiguous bytes each and a wave 4 KB in one piece,
//   * the trailing kernels map a wave to 8 columns x 8 rows = 512 contiguous bytes per load.
//   k_qr_panel  one workgroup keeps the whole panel (m x 8) in REGISTERS and runs the 8 reflectors with one block reduction
//               each: the dots of column k with the columns right of it give the norm AND every v^T a_c at once.  Same
//               reflector convention as the oracle (alpha = -sign(x0) |x|, v0 = 1).
//   k_qr_w      W = V^T [V | C] for the trailing columns C (one workgroup per column panel and 512-row chunk); its first 8
//               columns are V^T V.
//   k_qr_apply  sums the row-chunk partials of its 8 columns, solves (striu(V^T V) + diag(1/tau))^T Z = W - the compact-WY
//               factor through its inverse (T^-1 = striu(V^T V) + diag(1/tau)), so no T is ever built - and does
//               C -= V Z, one read-modify-write of the trailing matrix per panel.
// FP64 VALU throughout (FMA rate = MFMA rate on gfx950; the products here are 8-wide, below an MFMA tile).
// gfx950 only.
#include "dev_common.h"
#include "launch_qr.h"
#include "launch_chol.h"

#define QR_NB 8
#define QR_NT 512                // 8 waves = 2 per SIMD: 256 VGPRs per lane, the 12 x 8 panel slice of a thread stays in registers
#define QR_RPT 12                // rows per thread in the panel kernel: m - j0 <= 6144
#define QR_RC 512                // rows per workgroup in k_qr_w / k_qr_apply

namespace {

// element (r, c) of the panel-major working copy; mp8 = 8 * (rows rounded up to 8)
__device__ __forceinline__ size_t didx(int r, int c, size_t mp8) { return (size_t)(c >> 3) * mp8 + (size_t)r * 8 + (c & 7); }

// column-major H (ldh) | res  ->  panel-major D, column n = res; padding columns of the last panel are zero.
// Rows [src0, src0 + cnt) of the input go to rows [dst0, dst0 + cnt) of the working copy.
__global__ __launch_bounds__(256) void k_qr_load(const double* __restrict__ H, int ldh, const double* __restrict__ res, int n,
                                                  double* __restrict__ D, size_t mp8, int src0, int dst0, int cnt)
{
    __shared__ double t[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int cc = ty; cc < 32; cc += 8) {
        const int r = r0 + tx, c = c0 + cc;
        t[cc][tx] = (r < cnt && c <= n) ? (c < n ? H[(size_t)(src0 + r) + (size_t)c * ldh] : res[src0 + r]) : 0.0;
    }
    __syncthreads();
    const int ncp = (n + 1 + 7) & ~7;
    for (int rr = ty; rr < 32; rr += 8) {
        const int r = r0 + rr, c = c0 + tx;
        if (r < cnt && c < ncp) D[didx(dst0 + r, c, mp8)] = t[tx][rr];
    }
}

// between two row chunks of a tall matrix: the top n rows keep R (and Q^T res in column n), the reflectors stored below the
// diagonal are cleared so that [R ; next rows] is the next matrix to factorise
__global__ __launch_bounds__(256) void k_qr_clear_lower(double* __restrict__ D, size_t mp8, int n)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    if (i > j) D[didx(i, j, mp8)] = 0.0;
}

__global__ __launch_bounds__(QR_NT) void k_qr_panel(double* __restrict__ D, size_t mp8, int m, int j0, int nbp, double* __restrict__ tau_out)
{
    __shared__ double red[QR_NT / WAVE][QR_NB];
    __shared__ double sg[QR_NB], bc[QR_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* const Pn = D + (size_t)(j0 >> 3) * mp8;                    // this panel: [row][8]
    double x[QR_RPT][QR_NB];
#pragma unroll
    for (int i = 0; i < QR_RPT; ++i) {
        const int r = j0 + tid + QR_NT * i;
        const double2* src = reinterpret_cast<const double2*>(Pn + (size_t)(r < m ? r : 0) * 8);
#pragma unroll
        for (int c2 = 0; c2 < QR_NB / 2; ++c2) {
            const double2 v = src[c2];
            x[i][2 * c2] = (r < m && 2 * c2 < nbp) ? v.x : 0.0;
            x[i][2 * c2 + 1] = (r < m && 2 * c2 + 1 < nbp) ? v.y : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < QR_NB; ++k) {
        if (k < nbp) {                                                   // uniform
            double p[QR_NB];
#pragma unroll
            for (int c = 0; c < QR_NB; ++c) p[c] = 0.0;
#pragma unroll
            for (int i = 0; i < QR_RPT; ++i) {
                const bool act = tid + QR_NT * i >= k;                   // rows j0+k and below
                const double xk = act ? x[i][k] : 0.0;
#pragma unroll
                for (int c = k; c < QR_NB; ++c) p[c] += xk * x[i][c];
            }
#pragma unroll
            for (int c = k; c < QR_NB; ++c) {
                double v = p[c];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
                if (lane == 0) red[wave][c] = v;
            }
            if (tid == k) {
#pragma unroll
                for (int c = 0; c < QR_NB; ++c) bc[c] = x[0][c];       // row j0+k of the panel
            }
            __syncthreads();
            if (wave == 0) {                                             // second stage: 8 waves x QR_NB values
                const int c = lane & (QR_NB - 1), wg = lane / QR_NB;     // wg: 0..7 = the wave whose partial this lane takes
                double v = red[wg][c];
#pragma unroll
                for (int off = QR_NB; off < WAVE; off <<= 1) v += __shfl_xor(v, off, WAVE);
                if (lane < QR_NB) sg[lane] = v;
            }
            __syncthreads();
            const double x0 = bc[k], nrm = sqrt(sg[k]);
            double tau = 0.0;
            if (nrm != 0.0) {                                            // oracle: a zero column is skipped
                const double alpha = x0 >= 0.0 ? -nrm : nrm, v0 = x0 - alpha, iv0 = 1.0 / v0;
                tau = -v0 / alpha;
                double w[QR_NB];
#pragma unroll
                for (int c = k + 1; c < QR_NB; ++c) w[c] = tau * ((sg[c] - x0 * bc[c]) * iv0 + bc[c]);
#pragma unroll
                for (int i = 0; i < QR_RPT; ++i) {
                    const int rr = tid + QR_NT * i;
                    if (rr > k) {
                        const double v = x[i][k] * iv0;
                        x[i][k] = v;
#pragma unroll
                        for (int c = k + 1; c < QR_NB; ++c) x[i][c] -= w[c] * v;
                    } else if (rr == k) {
                        x[i][k] = alpha;
#pragma unroll
                        for (int c = k + 1; c < QR_NB; ++c) x[i][c] -= w[c];
                    }
                }
            }
            if (tid == 0) tau_out[k] = tau;
            __syncthreads();                                             // red / bc / sg are reused by the next reflector
        }
    }
#pragma unroll
    for (int i = 0; i < QR_RPT; ++i) {
        const int r = j0 + tid + QR_NT * i;
        if (r < m) {
            double2* dst = reinterpret_cast<double2*>(Pn + (size_t)r * 8);
#pragma unroll
            for (int c2 = 0; c2 < QR_NB / 2; ++c2) {                     // columns >= nbp of a short last panel keep their (loaded) value
                if (2 * c2 + 1 < nbp) dst[c2] = make_double2(x[i][2 * c2], x[i][2 * c2 + 1]);
                else if (2 * c2 < nbp) Pn[(size_t)r * 8 + 2 * c2] = x[i][2 * c2];
            }
        }
    }
}

// V as the trailing kernels see it: unit lower-trapezoidal (the panel stores R on and above the diagonal)
__device__ __forceinline__ double vmask(double d, int rrel, int k) { return rrel > k ? d : (rrel == k ? 1.0 : 0.0); }

// stages V rows [r0, r0+nr) of panel j0 into LDS, masked
__device__ __forceinline__ void stage_v(const double* __restrict__ D, size_t mp8, int j0, int nbp, int r0, int nr, double (*sV)[QR_NB])
{
    const double* Pn = D + (size_t)(j0 >> 3) * mp8;
    for (int e = threadIdx.x; e < QR_RC * QR_NB; e += 256) {
        const int rr = e >> 3, k = e & 7;
        sV[rr][k] = (rr < nr && k < nbp) ? vmask(Pn[(size_t)(r0 + rr) * 8 + k], r0 + rr - j0, k) : 0.0;
    }
}

// Wp[chunk][k][cx] = sum over the chunk's rows of V[r][k] * X[r][cx],  X = D[:, j0 ...): V itself (cx < 8), then the trailing
// columns.  Workgroup = one 8-column panel x 512 rows; lane = (column cl, row lane rl): a wave reads 8 rows x 64 B in one piece.
__global__ __launch_bounds__(256) void k_qr_w(const double* __restrict__ D, size_t mp8, int m, int ncx, int j0, int nbp, double* __restrict__ Wp)
{
    __shared__ double sV[QR_RC][QR_NB];
    __shared__ double sAcc[4][QR_NB][8];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3, wave = threadIdx.x >> 6, pb = blockIdx.x, chunk = blockIdx.y;
    const int r0 = j0 + chunk * QR_RC, nr = min(QR_RC, m - r0);
    stage_v(D, mp8, j0, nbp, r0, nr, sV);
    __syncthreads();
    const double* Px = D + (size_t)((j0 >> 3) + pb) * mp8;
    double acc[QR_NB];
#pragma unroll
    for (int k = 0; k < QR_NB; ++k) acc[k] = 0.0;
#pragma unroll 4
    for (int rr = rl; rr < nr; rr += 32) {
        double xv = Px[(size_t)(r0 + rr) * 8 + cl];
        if (pb == 0) xv = cl < nbp ? vmask(xv, r0 + rr - j0, cl) : xv;   // the V columns of X
#pragma unroll
        for (int k = 0; k < QR_NB; ++k) acc[k] += sV[rr][k] * xv;
    }
#pragma unroll
    for (int k = 0; k < QR_NB; ++k) {
        double v = acc[k];
        v += __shfl_xor(v, 8, WAVE); v += __shfl_xor(v, 16, WAVE); v += __shfl_xor(v, 32, WAVE);      // the wave's 8 row lanes
        if ((threadIdx.x & 63) < 8) sAcc[wave][k][cl] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int k = threadIdx.x >> 3, c8 = threadIdx.x & 7, cx = 8 * pb + c8;
        if (cx < ncx) Wp[((size_t)chunk * QR_NB + k) * ncx + cx] = (sAcc[0][k][c8] + sAcc[1][k][c8]) + (sAcc[2][k][c8] + sAcc[3][k][c8]);
    }
}

// C -= V Z for one 8-column panel x 512 rows.  Z for these 8 columns is solved here (every row-chunk workgroup repeats the
// 8 x 8 forward substitution: cheaper than a kernel of its own): W = sum of the row-chunk partials,
// (striu(V^T V) + diag(1/tau))^T Z = W, V^T V = the first 8 columns of W.  tau_k = 0 (skipped reflector) gives z_k = 0.
__global__ __launch_bounds__(256) void k_qr_apply(double* __restrict__ D, size_t mp8, int m, int ncx, int j0, int nbp, const double* __restrict__ Wp,
                                                  int nchunks, const double* __restrict__ tau)
{
    __shared__ double sV[QR_RC][QR_NB];
    __shared__ double sW[QR_NB][8], sG[QR_NB][8], sZ[QR_NB][8];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3, pb = blockIdx.x, cx = 8 * pb + cl;
    const int r0 = j0 + blockIdx.y * QR_RC, nr = min(QR_RC, m - r0);
    stage_v(D, mp8, j0, nbp, r0, nr, sV);
    if (threadIdx.x < 128) {
        const int t = threadIdx.x & 63, k = t >> 3, c8 = t & 7, col = threadIdx.x < 64 ? 8 * pb + c8 : c8;
        double s = 0.0;
        if (col < ncx) for (int ch = 0; ch < nchunks; ++ch) s += Wp[((size_t)ch * QR_NB + k) * ncx + col];
        if (threadIdx.x < 64) sW[k][c8] = s; else sG[k][c8] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double z[QR_NB];
#pragma unroll
        for (int k = 0; k < QR_NB; ++k) {
            double s = sW[k][cl];
#pragma unroll
            for (int l = 0; l < k; ++l) s -= sG[l][k] * z[l];
            z[k] = k < nbp ? tau[k] * s : 0.0;
            sZ[k][cl] = z[k];
        }
    }
    __syncthreads();
    if (cx >= ncx || cx < nbp) return;                                   // (a short last panel shares its 8-group with trailing columns)
    double z[QR_NB];
#pragma unroll
    for (int k = 0; k < QR_NB; ++k) z[k] = sZ[k][cl];
    double* Px = D + (size_t)((j0 >> 3) + pb) * mp8;
#pragma unroll 4
    for (int rr = rl; rr < nr; rr += 32) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < QR_NB; ++k) s += sV[rr][k] * z[k];
        Px[(size_t)(r0 + rr) * 8 + cl] -= s;
    }
}

// R (n x n upper, column-major ldt) and Q^T res (first n entries) out of the working copy
__global__ __launch_bounds__(256) void k_qr_extract(const double* __restrict__ D, size_t mp8, int m, int n, double* __restrict__ Ht, int ldt,
                                                     double* __restrict__ rt)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n * n) {
        const int i = e / n, j = e - i * n;                              // row i, column j
        Ht[i + (size_t)j * ldt] = (i <= j && i < m) ? D[didx(i, j, mp8)] : 0.0;
    }
    if (e < n) rt[e] = e < m ? D[didx(e, n, mp8)] : 0.0;
}

// ---- Cholesky-QR for tall stacks (launch_qr_chol) ----------------------------------------------------------------------
// lower triangle of G = sum of the split-K partials of [H | r]^T [H | r]; identity on the padding diagonal
__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ part, size_t pstride, int ks, int n1, int n32, double* __restrict__ X)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)n32 * n32; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % n32), j = (int)(e / n32);
        double v = 0.0;
        if (i >= j) {
            if (i < n1) { for (int s = 0; s < ks; ++s) v += part[(size_t)s * pstride + e]; }
            else v = i == j ? 1.0 : 0.0;
        }
        X[e] = v;
    }
}

// R = L^T (n x n upper triangular, column-major ldt), Q^T r = row n of L
__global__ __launch_bounds__(256) void k_chol_extract(const double* __restrict__ Y, int n32, int n, double* __restrict__ Ht, int ldt, double* __restrict__ rt)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n * n) {
        const int j = e % n, i = e / n;                                     // consecutive threads read along a column of L
        Ht[i + (size_t)j * ldt] = j >= i ? Y[(size_t)j + (size_t)i * n32] : 0.0;
    }
    if (e < n) rt[e] = Y[(size_t)n + (size_t)e * n32];
}

}  // namespace

static int qrc_n32(int n) { return (n + 1 + 31) / 32 * 32; }
static int qrc_ksplit(int m, int n) { return gram_ksplit(m, n); }
size_t qr_chol_workspace_doubles(int m, int n)
{
    const size_t n32 = qrc_n32(n);
    return ((size_t)qrc_ksplit(m, n) + 2) * n32 * n32 + 2048 + n32;
}

// Cholesky-QR: R^T R = H^T H and R^T z = H^T r are all the Kalman update reads from the thin QR (StateManager.cpp:359-411 is
// invariant under any orthogonal transformation of the stacked rows), so R is taken as the Cholesky factor of the Gram matrix
// of [H | r] - one FP64 MFMA GEMM over the m rows and a blocked Cholesky of n+1 columns (kernels_chol.hip) instead of n
// dependent Householder reflectors.  Diagonal of R positive; a rank-deficient H (quirk Q9) gives zero rows where a pivot falls
// below 1e-13 of its column's squared norm.  Entry-wise accuracy of R is cond(H)^2 eps (Householder: cond(H) eps); R^T R is
// exact to eps |H|^2 either way.
int launch_qr_chol(const double* dH, int ldh, const double* dres, int m, int n, double* ws, double* dHt, int ldt, double* drt, hipStream_t st)
{
    const int n1 = n + 1, n32 = qrc_n32(n), ks = qrc_ksplit(m, n);
    const size_t nn = (size_t)n32 * n32;
    double *part = ws, *X = ws + (size_t)ks * nn, *Y = X + nn, *Tb = Y + nn;
    launch_gram(dH, ldh, dres, m, n, part, nn, n32, ks, st);
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((nn + 1023) / 1024)), dim3(256), 0, st, part, nn, ks, n1, n32, X);
    CholArgs c = {};
    c.W = X; c.Y = Y; c.xs = 0; c.ld = n32; c.Tb = Tb; c.ts = 0; c.rows = n32; c.ncols = n32; c.clamp = 1; c.clamp_rel = 1e-13; c.batch = 1;
    launch_chol_sweep(c, st);
    hipLaunchKernelGGL(k_chol_extract, dim3((n * n + 255) / 256), dim3(256), 0, st, Y, n32, n, dHt, ldt, drt);
    return 0;
}

static size_t qr_mp8(int m) { return (size_t)8 * (((size_t)m + 7) & ~(size_t)7); }


// one factorisation of the top m_eff rows of the working copy (all columns incl. res)
static void qr_factor_rows(double* D, size_t mp8, int m_eff, int n, double* Wp, double* tau, hipStream_t st)
{
    const int n1 = n + 1;
    const int nref = m_eff - 1 < n ? m_eff - 1 : n;                      // reflectors: min(m-1, n), as the oracle
    for (int j0 = 0; j0 < nref; j0 += QR_NB) {
        const int nbp = nref - j0 < QR_NB ? nref - j0 : QR_NB;
        hipLaunchKernelGGL(k_qr_panel, dim3(1), dim3(QR_NT), 0, st, D, mp8, m_eff, j0, nbp, tau);
        const int ncx = n1 - j0;                                         // X = columns j0 .. n (incl. res)
        if (ncx > nbp) {
            const int nch = (m_eff - j0 + QR_RC - 1) / QR_RC, npx = (ncx + 7) / 8;
            hipLaunchKernelGGL(k_qr_w, dim3(npx, nch), dim3(256), 0, st, D, mp8, m_eff, ncx, j0, nbp, Wp);
            hipLaunchKernelGGL(k_qr_apply, dim3(npx, nch), dim3(256), 0, st, D, mp8, m_eff, ncx, j0, nbp, Wp, nch, tau);
        }
    }
}

#define QR_ROW_CAP (QR_NT * QR_RPT)      // rows the register-resident panel kernel can hold

size_t qr_dense_workspace_doubles(int m, int n)
{
    const size_t mw = m < QR_ROW_CAP ? m : QR_ROW_CAP;
    const size_t npan = ((size_t)n + 1 + 7) / 8, nch = (mw + QR_RC - 1) / QR_RC, ncp = npan * 8;
    return npan * qr_mp8((int)mw) + nch * QR_NB * ncp + (size_t)QR_NB * ncp + 64;
}

// Any m: a matrix taller than the panel kernel's 6144 rows is factorised in row chunks (a sequential TSQR): the first 6144 rows
// give R_0, then [R_s ; next 6144 - n rows] gives R_s+1 ... - e.g. BASELINE config 5's literal stacked shape 35100 x 180
// (RemoveLostUpdate.cpp:376-397 at F = 300, C = 30) takes 6 factorisations.  Returns -1 when n leaves no room for a chunk.
int launch_qr_dense(const double* dH, int ldh, const double* dres, int m, int n, double* ws, double* dHt, int ldt, double* drt, hipStream_t st)
{
    if (m > QR_ROW_CAP && n + 64 > QR_ROW_CAP) return -1;
    const int n1 = n + 1, npan = (n1 + 7) / 8, ncp = npan * 8;
    const int mw = m < QR_ROW_CAP ? m : QR_ROW_CAP;
    const size_t mp8 = qr_mp8(mw);
    double* D = ws;
    double* Wp = D + (size_t)npan * mp8;
    const int nch_max = (mw + QR_RC - 1) / QR_RC;
    double* tau = Wp + (size_t)nch_max * QR_NB * ncp;
    hipLaunchKernelGGL(k_qr_load, dim3((mw + 31) / 32, (ncp + 31) / 32), dim3(256), 0, st, dH, ldh, dres, n, D, mp8, 0, 0, mw);
    qr_factor_rows(D, mp8, mw, n, Wp, tau, st);
    int done = mw, m_last = mw;
    while (done < m) {
        const int cnt = m - done < QR_ROW_CAP - n ? m - done : QR_ROW_CAP - n;
        hipLaunchKernelGGL(k_qr_clear_lower, dim3((n * n + 255) / 256), dim3(256), 0, st, D, mp8, n);
        hipLaunchKernelGGL(k_qr_load, dim3((cnt + 31) / 32, (ncp + 31) / 32), dim3(256), 0, st, dH, ldh, dres, n, D, mp8, done, n, cnt);
        m_last = n + cnt;
        qr_factor_rows(D, mp8, m_last, n, Wp, tau, st);
        done += cnt;
    }
    hipLaunchKernelGGL(k_qr_extract, dim3((n * n + 255) / 256), dim3(256), 0, st, D, mp8, m_last, n, dHt, ldt, drt);
    return 0;
}
