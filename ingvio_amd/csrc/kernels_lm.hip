// kernels_lm.hip — SURVEY.md §8(f) row f-2: the covariance operations of the SLAM-landmark path.
//   k_delayed_qr    StateManager::addVariableDelayed, Givens part (StateManager.cpp:577-589): H_new (m x s) to upper
//                   triangular, the same rotations on res and H_old.  The rotation list depends on H_new only, so one
//                   lane derives it (s*(m-1) rotations on <= s columns) and then every column of [H_old | res] is
//                   swept by its own lane with a running carry: each element is read and written once per H_new
//                   column, no barrier inside the sweep.
//   k_delayed_add   StateManager::addVariableDelayedInvertible (:461-543): PH^T, S = H Pcc H^T + sigma^2 I,
//                   H_new^-1 (s <= 6, partial-pivot Gauss-Jordan by one lane, what Eigen's inverse() does for
//                   dynamic sizes), new rows/columns -PH^T H_new^-T, corner H_new^-1 S H_new^-T, n += s.
//   k_replace_var   StateManager::replaceVarLinear (:639-693): rows/columns of the target variable <- P H^T (computed
//                   from the untouched P), diagonal block <- H Pcc H^T.
// All three are one workgroup per call: a landmark is 3 columns of a <= 1000-column matrix, the work is O(N s nc).
// gfx950 only.
#include "dev_common.h"
#include "launch_lm.h"

#define LM_NT 256
#define LM_SMAX 6

namespace {

// Eigen::JacobiRotation<double>::makeGivens (real case)
__device__ __forceinline__ void make_givens(double p, double q, double& c, double& s)
{
    if (q == 0.0) { c = p < 0.0 ? -1.0 : 1.0; s = 0.0; }
    else if (p == 0.0) { c = 0.0; s = q < 0.0 ? 1.0 : -1.0; }
    else if (fabs(p) > fabs(q)) {
        const double t = q / p;
        double u = sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        c = 1.0 / u; s = -t * c;
    } else {
        const double t = p / q;
        double u = sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        s = -1.0 / u; c = -t * s;
    }
}

__global__ __launch_bounds__(LM_NT) void k_delayed_qr(double* __restrict__ H_old, double* __restrict__ res, double* __restrict__ H_new,
                                                       int m, int s, int nc, int mld)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sCS = reinterpret_cast<double*>(smem_raw);                  // [s][m][2]: (c, s) of the rotation on rows (r-1, r) of column col
    double* sHn = sCS + 2 * (size_t)s * m;                              // [s][m]: H_new staged in LDS for the serial part
    const int tid = threadIdx.x;
    for (int e = tid; e < s * m; e += LM_NT) sHn[e] = H_new[(e % m) + (size_t)(e / m) * mld];
    __syncthreads();
    if (tid == 0) {
        for (int col = 0; col < s; ++col)
            for (int r = m - 1; r > col; --r) {
                double c, sn;
                make_givens(sHn[(r - 1) + col * m], sHn[r + col * m], c, sn);
                sCS[2 * (col * m + r)] = c; sCS[2 * (col * m + r) + 1] = sn;
                for (int j = col; j < s; ++j) {                          // applyOnTheLeft(G.adjoint()): x' = c x - s y, y' = s x + c y
                    const double x = sHn[(r - 1) + j * m], y = sHn[r + j * m];
                    sHn[(r - 1) + j * m] = c * x - sn * y;
                    sHn[r + j * m] = sn * x + c * y;
                }
            }
    }
    __syncthreads();
    for (int e = tid; e < s * m; e += LM_NT) H_new[(e % m) + (size_t)(e / m) * mld] = sHn[e];
    for (int j = tid; j <= nc; j += LM_NT) {
        double* A = j < nc ? H_old + (size_t)j * mld : res;
        for (int col = 0; col < s; ++col) {
            double carry = A[m - 1];
            for (int r = m - 1; r > col; --r) {
                const double c = sCS[2 * (col * m + r)], sn = sCS[2 * (col * m + r) + 1];
                const double x = A[r - 1];
                A[r] = sn * x + c * carry;
                carry = c * x - sn * carry;
            }
            A[col] = carry;
        }
    }
}

// s x s inverse by Gauss-Jordan with partial pivoting, in LDS, one lane
__device__ void inv_small(double* A, double* Ai, int s)
{
    for (int i = 0; i < s * s; ++i) Ai[i] = (i % (s + 1) == 0) ? 1.0 : 0.0;
    for (int j = 0; j < s; ++j) {
        int p = j; double mx = fabs(A[j + j * s]);
        for (int i = j + 1; i < s; ++i) if (fabs(A[i + j * s]) > mx) { mx = fabs(A[i + j * s]); p = i; }
        if (p != j)
            for (int c = 0; c < s; ++c) {
                double t = A[j + c * s]; A[j + c * s] = A[p + c * s]; A[p + c * s] = t;
                t = Ai[j + c * s]; Ai[j + c * s] = Ai[p + c * s]; Ai[p + c * s] = t;
            }
        const double d = 1.0 / A[j + j * s];
        for (int c = 0; c < s; ++c) { A[j + c * s] *= d; Ai[j + c * s] *= d; }
        for (int i = 0; i < s; ++i) {
            if (i == j) continue;
            const double f = A[i + j * s];
            for (int c = 0; c < s; ++c) { A[i + c * s] -= f * A[j + c * s]; Ai[i + c * s] -= f * Ai[j + c * s]; }
        }
    }
}

__global__ __launch_bounds__(LM_NT) void k_delayed_add(CovView cv, int b, const double* __restrict__ Hx, const double* __restrict__ Hf,
                                                        const int* __restrict__ colmap, int s, int nc, int mld, double var,
                                                        double* __restrict__ Y /* n x s scratch, ld = ldp */)
{
    __shared__ double sHf[LM_SMAX * LM_SMAX], sHi[LM_SMAX * LM_SMAX], sS[LM_SMAX * LM_SMAX], sT[LM_SMAX * LM_SMAX];
    const int tid = threadIdx.x, n = cv.n[b], ld = cv.ldp;
    double* P = cov_ptr(cv, b);
    // PH^T (:490-505)
    for (int r = tid; r < n; r += LM_NT) {
        double acc[LM_SMAX];
#pragma unroll
        for (int j = 0; j < LM_SMAX; ++j) acc[j] = 0.0;
        for (int c = 0; c < nc; ++c) {
            const double p = P[r + (size_t)colmap[c] * ld];
#pragma unroll
            for (int j = 0; j < LM_SMAX; ++j) if (j < s) acc[j] += p * Hx[j + (size_t)c * mld];
        }
#pragma unroll
        for (int j = 0; j < LM_SMAX; ++j) if (j < s) Y[r + (size_t)j * ld] = acc[j];
    }
    if (tid < s * s) sHf[tid] = Hf[(tid % s) + (size_t)(tid / s) * mld];
    __syncthreads();
    // S = Hx PHT[cols] + var I (:507-514), H_new^-1 (:516)
    if (tid < s * s) {
        const int i = tid % s, j = tid / s;
        double acc = 0.0;
        for (int c = 0; c < nc; ++c) acc += Hx[i + (size_t)c * mld] * Y[colmap[c] + (size_t)j * ld];
        sS[i + j * s] = acc + (i == j ? var : 0.0);
    }
    if (tid == 0) inv_small(sHf, sHi, s);
    __syncthreads();
    // T = Hi S ; corner = T Hi^T (:518), symmetrised as :534 does for the whole matrix
    if (tid < s * s) {
        const int i = tid % s, j = tid / s;
        double acc = 0.0;
        for (int l = 0; l < s; ++l) acc += sHi[i + l * s] * sS[l + j * s];
        sT[i + j * s] = acc;
    }
    __syncthreads();
    if (tid < s * s) {
        const int i = tid % s, j = tid / s;
        double a = 0.0, bt = 0.0;
        for (int l = 0; l < s; ++l) { a += sT[i + l * s] * sHi[j + l * s]; bt += sT[j + l * s] * sHi[i + l * s]; }
        P[(n + i) + (size_t)(n + j) * ld] = 0.5 * (a + bt);
    }
    // cross = -PH^T Hi^T (:526-528)
    for (int r = tid; r < n; r += LM_NT) {
        double y[LM_SMAX];
#pragma unroll
        for (int l = 0; l < LM_SMAX; ++l) y[l] = l < s ? Y[r + (size_t)l * ld] : 0.0;
        for (int j = 0; j < s; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < LM_SMAX; ++l) if (l < s) acc += y[l] * sHi[j + l * s];
            P[r + (size_t)(n + j) * ld] = -acc;
            P[(n + j) + (size_t)r * ld] = -acc;
        }
    }
    __syncthreads();
    if (tid == 0) cv.n[b] = n + s;
}

__global__ __launch_bounds__(LM_NT) void k_replace_var(CovView cv, int b, const double* __restrict__ H, const int* __restrict__ colmap,
                                                        int tidx, int ts, int nc, int mld, double* __restrict__ Y)
{
    __shared__ double sB[LM_SMAX * LM_SMAX];
    const int tid = threadIdx.x, n = cv.n[b], ld = cv.ldp;
    double* P = cov_ptr(cv, b);
    for (int r = tid; r < n; r += LM_NT) {
        double acc[LM_SMAX];
#pragma unroll
        for (int j = 0; j < LM_SMAX; ++j) acc[j] = 0.0;
        for (int c = 0; c < nc; ++c) {
            const double p = P[r + (size_t)colmap[c] * ld];
#pragma unroll
            for (int j = 0; j < LM_SMAX; ++j) if (j < ts) acc[j] += p * H[j + (size_t)c * mld];
        }
#pragma unroll
        for (int j = 0; j < LM_SMAX; ++j) if (j < ts) Y[r + (size_t)j * ld] = acc[j];
    }
    __syncthreads();                                                      // PH^T complete before P is touched
    double hph = 0.0;
    if (tid < ts * ts) {                                                  // H Pcc H^T (:683-685) from PHT[cols]
        const int i = tid % ts, j = tid / ts;
        for (int c = 0; c < nc; ++c) hph += H[i + (size_t)c * mld] * Y[colmap[c] + (size_t)j * ld];
        sB[i + j * ts] = hph;
    }
    __syncthreads();
    // the reference stores H Pcc H^T as computed (symmetric up to rounding); the device keeps P exactly symmetric
    if (tid < ts * ts) hph = 0.5 * (sB[tid % ts + (tid / ts) * ts] + sB[tid / ts + (tid % ts) * ts]);
    for (int e = tid; e < n * ts; e += LM_NT) {
        const int r = e % n, j = e / n;
        if (r >= tidx && r < tidx + ts) continue;
        const double v = Y[r + (size_t)j * ld];
        P[r + (size_t)(tidx + j) * ld] = v;                               // :687
        P[(tidx + j) + (size_t)r * ld] = v;                               // :689
    }
    if (tid < ts * ts) P[(tidx + tid % ts) + (size_t)(tidx + tid / ts) * ld] = hph;      // :691
}

}  // namespace

int launch_delayed_qr(double* H_old, double* res, double* H_new, int m, int s, int nc, int mld, hipStream_t st)
{
    const size_t sm = sizeof(double) * 3 * (size_t)s * m;
    if (s < 1 || s > LM_SMAX || sm > 60 * 1024) return -1;
    hipLaunchKernelGGL(k_delayed_qr, dim3(1), dim3(LM_NT), sm, st, H_old, res, H_new, m, s, nc, mld);
    return 0;
}

int launch_delayed_add(CovView cv, int b, const double* Hx, const double* Hf, const int* colmap, int s, int nc, int mld, double var,
                       double* Y, hipStream_t st)
{
    if (s < 1 || s > LM_SMAX) return -1;
    hipLaunchKernelGGL(k_delayed_add, dim3(1), dim3(LM_NT), 0, st, cv, b, Hx, Hf, colmap, s, nc, mld, var, Y);
    return 0;
}

int launch_replace_var(CovView cv, int b, const double* H, const int* colmap, int tidx, int ts, int nc, int mld, double* Y, hipStream_t st)
{
    if (ts < 1 || ts > LM_SMAX) return -1;
    hipLaunchKernelGGL(k_replace_var, dim3(1), dim3(LM_NT), 0, st, cv, b, H, colmap, tidx, ts, nc, mld, Y);
    return 0;
}
