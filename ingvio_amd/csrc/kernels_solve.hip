// kernels_solve.hip — K8 / K9 / K11 of the factored MSCKF path in SYMMETRIC form, on the FP64 matrix cores.
//
// The information-form update needs  M = (A Pcc + s^2 I)^-1 A  and  t = (A Pcc + s^2 I)^-1 b   (StateManager.cpp:399-405 after the
// push-through identity, see kernels_factored.hip).  A Pcc + s^2 I is a product of two symmetric matrices plus a shift: its
// eigenvalues are those of the SPD innovation matrix, but it is far from normal (A has rank n - 6), and Gauss-Jordan on it loses
// cond(A^1/2) more digits than the reference's SPD solve (measured: dx off by 3e-5 at a 100x inflated prior, tests/test_gpu_pinning.py).
// With  Pcc = L D L^T  (unit lower L):
//        A Pcc + s^2 I = L^-T (L^T A L + s^2 D^-1) D L^T,      W := L^T A L + s^2 D^-1   SPD, eigenvalues >= s^2 / max(D)
//        M = (L^-T D^-1) W^-1 (L^T A),   t = (L^-T D^-1) W^-1 (L^T b)
// and with  W = L2 D2 L2^T:   M = R2' D2^-1 R1'^T,   R2' = (L^-T D^-1) L2^-T,   R1' = (A L) L2^-T.
// Both factorisations are blocked LDL^T (panels of 4 pivots) on 16x16 tiles in the MFMA C/D layout, as in the gate kernel
// (gate_kernel.h), with CARRIED ROWS: rows C appended below a matrix that is being factorised come out as C L^-T D^-1 (their
// "L entries") / C L^-T (before the D scaling) — so carrying the identity through the first factorisation yields L^-T D^-1, and
// carrying [A L ; b^T L] and L^-T D^-1 through the second yields R1' D2^-1 and R2' without any triangular solve.
// The three products (A L, L^T (A L), R2' D2^-1 R1'^T) are MFMA GEMMs on operands staged in LDS.  No pivot search, no atomics, one
// workgroup barrier per panel (the panel buffer is double-buffered).
// One workgroup of 4 waves per filter; tiles are dealt round-robin to the waves.  gfx950 only.
#include "launch_factored.h"

typedef double double4_f __attribute__((ext_vector_type(4)));

namespace {

template <int NC>
struct SolveCfg {
    static constexpr int NT = (NC + 15) / 16, NP = 16 * NT;          // tile rows / padded size of the n x n matrices
    static constexpr int NR1 = (NC + 1 + 15) / 16, R1ROWS = 16 * NR1; // rows of [A L ; b^T L]: the b row sits at row NC
    static constexpr int NLT = NT * (NT + 1) / 2;
    static constexpr int NW = 4;
    static constexpr int P1 = 2 * NLT, P2 = NLT + NR1 * NT + NLT;    // tiles of the two factorisations (matrix + carried)
    static constexpr int S1 = (P1 + NW - 1) / NW, S2 = (P2 + NW - 1) / NW;
    static constexpr int MROWS = R1ROWS > NP ? R1ROWS : NP, LDM = NP + 1;
    static constexpr int PANROWS = NP + R1ROWS + NP;
    static constexpr int MP = (NC + 3) & ~3;
    static constexpr size_t lds_bytes() { return sizeof(double) * (2 * (size_t)MROWS * LDM + 2 * (size_t)PANROWS * 4 + 2 * NP) + sizeof(int) * NP; }
};

__device__ __forceinline__ void tri_decode(int q, int& hi, int& lo)      // q -> (hi >= lo), row-major over the lower triangle
{
    hi = 0;
    while ((hi + 1) * (hi + 2) / 2 <= q) ++hi;
    lo = q - hi * (hi + 1) / 2;
}

struct Slot { int arow, tcol, rt; bool valid; };      // pan row base of the tile's rows, its column tile, carried-upper row tile or -1

// One blocked LDL^T sweep over the tiles of T (S slots per wave).  emit(row, col, x, x * dinv) is called once for every (pan row,
// pivot column) with the row's L^-T-transformed entry; dsave(k, q, 1/d) once per pivot.
template <int S, int PANROWS, int NW, class Emit, class DSave>
__device__ __forceinline__ void ldl_sweep(double4_f (&T)[S], const Slot (&sl)[S], int npan, int np_rows, int nrowtiles, double (*pan)[PANROWS][4],
                                          int wave, int lane, Emit emit, DSave dsave, int* bad)
{
    const int kq = lane >> 4, l15 = lane & 15;
    for (int k = 0; k < npan; ++k) {
        const int buf = k & 1, tj0 = k >> 2, cb = 4 * (k & 3);
        if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
            for (int u = 0; u < S; ++u)
                if (sl[u].valid && sl[u].tcol == tj0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) pan[buf][sl[u].arow + kq + 4 * r][l15 - cb] = T[u][r];
                }
        }
        __syncthreads();
        double a[4][4];
#pragma unroll
        for (int ra = 0; ra < 4; ++ra) {
            const double2* pr = reinterpret_cast<const double2*>(pan[buf][4 * k + ra]);
            const double2 u0 = pr[0], u1 = pr[1];
            a[ra][0] = u0.x; a[ra][1] = u0.y; a[ra][2] = u1.x; a[ra][3] = u1.y;
        }
        // 4x4 LDL^T of the diagonal block (every lane, uniform data)
        const double p0 = a[0][0], r0 = fast_rcp(p0);
        const double l10 = a[1][0] * r0, l20 = a[2][0] * r0, l30 = a[3][0] * r0;
        const double p1 = a[1][1] - l10 * a[1][0], r1 = fast_rcp(p1);
        const double t21 = a[2][1] - l20 * a[1][0], t31 = a[3][1] - l30 * a[1][0];
        const double l21 = t21 * r1, l31 = t31 * r1;
        const double p2 = a[2][2] - l20 * a[2][0] - l21 * t21, r2 = fast_rcp(p2);
        const double t32 = a[3][2] - l30 * a[2][0] - l31 * t21;
        const double l32 = t32 * r2;
        const double p3 = a[3][3] - l30 * a[3][0] - l31 * t31 - l32 * t32, r3 = fast_rcp(p3);
        const double dsel = kq == 0 ? r0 : (kq == 1 ? r1 : (kq == 2 ? r2 : r3));
        if (!(p0 > 0.0) || !(p1 > 0.0) || !(p2 > 0.0) || !(p3 > 0.0)) *bad = 1;
        if (wave == 0 && lane < 4) dsave(k, lane, lane == 0 ? r0 : (lane == 1 ? r1 : (lane == 2 ? r2 : r3)));
        auto xrow = [&](int row) {              // entry kq of (pan row) L_d^-T
            const double2* pr = reinterpret_cast<const double2*>(pan[buf][row]);
            const double2 u0 = pr[0], u1 = pr[1];
            const double x0 = u0.x;
            const double x1 = u0.y - l10 * x0;
            const double x2 = u1.x - l20 * x0 - l21 * x1;
            const double x3 = u1.y - l30 * x0 - l31 * x1 - l32 * x2;
            return kq == 0 ? x0 : (kq == 1 ? x1 : (kq == 2 ? x2 : x3));
        };
        // the factor entries of this panel: row tiles are dealt to the waves
        for (int t = wave; t < nrowtiles; t += NW) {
            const int row = 16 * t + l15;
            if (row < np_rows && row < 4 * k) continue;      // finished matrix rows: L is zero above the diagonal
            const double x = xrow(row);
            emit(t, row, 4 * k + kq, tj0, x, x * dsel);
        }
        // trailing update: T(ti, tj) -= X_i D^-1 X_j^T
#pragma unroll
        for (int u = 0; u < S; ++u) {
            if (sl[u].valid && sl[u].tcol >= tj0 && (sl[u].rt < 0 || sl[u].rt <= tj0)) {
                const int ra = sl[u].arow + l15, rb = 16 * sl[u].tcol + l15;
                double xa = xrow(ra), xb = -xrow(rb) * dsel;
                if (ra < np_rows && ra <= 4 * k + 3) xa = 0.0;      // pivot rows and everything above: finished
                if (rb <= 4 * k + 3) xb = 0.0;
                T[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, T[u], 0, 0, 0);
            }
        }
    }
    __syncthreads();
}

template <int NC>
__global__ __launch_bounds__(256) void k_info_solve(
    CovView cv, FrameView fv, int b0, const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
    const double* __restrict__ noise_all, double* __restrict__ Mall, int mstride, double* __restrict__ Pcall, int ystride,
    double* __restrict__ dx_all, int* __restrict__ m_out, int* __restrict__ nc_out, int* __restrict__ status,
    const int* __restrict__ marg_idx, int* __restrict__ pc_base_out)
{
    using Cfg = SolveCfg<NC>;
    constexpr int NT = Cfg::NT, NP = Cfg::NP, NR1 = Cfg::NR1, R1ROWS = Cfg::R1ROWS, NLT = Cfg::NLT, NW = Cfg::NW;
    constexpr int S1 = Cfg::S1, S2 = Cfg::S2, LDM = Cfg::LDM, MROWS = Cfg::MROWS, PANROWS = Cfg::PANROWS, MP = Cfg::MP;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* X = reinterpret_cast<double*>(smem_raw);                 // MROWS x LDM: L, later R1' D2^-1
    double* Y = X + (size_t)MROWS * LDM;                             // MROWS x LDM: L^-T D^-1, then A L, later R2'
    double (*pan)[PANROWS][4] = reinterpret_cast<double (*)[PANROWS][4]>(Y + (size_t)MROWS * LDM);
    double* sD1inv = reinterpret_cast<double*>(pan) + 2 * (size_t)PANROWS * 4;      // NP
    double* sD2inv = sD1inv + NP;                                                    // NP
    int* sCol = reinterpret_cast<int*>(sD2inv + NP);                                 // NP
    __shared__ int sBad;
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, l15 = lane & 15;
    const int C = fv.n_clones[b], ncol = 6 * C, n = cv.n[b], ld = cv.ldp;
    double* dx = dx_all + (size_t)b * ld;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        for (int r = tid; r < n; r += 256) dx[r] = 0.0;
        if (tid == 0) { m_out[bl] = 0; nc_out[bl] = ncol; pc_base_out[bl] = -1; }
        return;
    }
    const double* P = cov_ptr(cv, b);
    const double var = noise_all[bl];
    for (int c = tid; c < NP; c += 256) { const int cc = c < ncol ? c : 0; sCol[c] = fv.clone_idx[(size_t)b * fv.cmax + cc / 6] + cc % 6; }
    for (int e = tid; e < 2 * MROWS * LDM; e += 256) X[e] = 0.0;
    if (tid == 0) sBad = 0;
    __syncthreads();
    const bool fused = marg_idx && marg_idx[bl] >= 0;
    const int contig = __syncthreads_and(tid >= ncol || sCol[tid < NP ? tid : 0] == sCol[0] + tid);
    const bool zero_copy = fused && contig && sCol[0] + MP <= ld;
    int bad = 0;

    // ================= factorisation 1: Pcc = L D L^T, identity carried =================
    {
        Slot sl[S1];
        double4_f T[S1];
#pragma unroll
        for (int u = 0; u < S1; ++u) {
            const int p = u * NW + wave;
            sl[u].valid = p < Cfg::P1;
            int hi = 0, lo = 0;
            tri_decode(p < NLT ? p : (p < Cfg::P1 ? p - NLT : 0), hi, lo);
            const bool carried = p >= NLT;
            sl[u].arow = carried ? NP + 16 * lo : 16 * hi;      // lower tile (hi, lo) / carried upper tile (row lo, col hi)
            sl[u].tcol = carried ? hi : lo;
            sl[u].rt = carried ? lo : -1;
            const int rt = carried ? lo : hi, ct = sl[u].tcol;
            const int col = 16 * ct + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + kq + 4 * r;
                double v = row == col ? 1.0 : 0.0;
                if (!carried && sl[u].valid && row < ncol && col < ncol) v = P[sCol[col] + (size_t)sCol[row] * ld];   // symmetric: coalesced along l15
                T[u][r] = v;
            }
        }
        auto emit = [&](int t, int row, int col, int tj0, double x, double xs) {
            if (row < NP) X[row * LDM + col] = row == col ? 1.0 : (row < col ? 0.0 : xs);   // L (unit lower, exactly)
            else if (t - NT <= tj0) Y[(row - NP) * LDM + col] = xs;                       // L^-T D^-1 (upper)
        };
        auto dsave = [&](int k, int q, double r) { sD1inv[4 * k + q] = r; };
        ldl_sweep<S1, PANROWS, NW>(T, sl, NP / 4, NP, 2 * NT, pan, wave, lane, emit, dsave, &bad);
    }
    // ================= second matrix and its carried rows =================
    Slot sl[S2];
    double4_f T[S2];
#pragma unroll
    for (int u = 0; u < S2; ++u) {
        const int p = u * NW + wave;
        sl[u].valid = p < Cfg::P2;
        int kind = p < NLT ? 0 : (p < NLT + NR1 * NT ? 1 : 2);
        int hi = 0, lo = 0;
        if (kind == 0) tri_decode(p, hi, lo);
        else if (kind == 2) tri_decode(p < Cfg::P2 ? p - NLT - NR1 * NT : 0, hi, lo);
        else { hi = (p - NLT) / NT; lo = (p - NLT) % NT; }
        sl[u].arow = kind == 0 ? 16 * hi : (kind == 1 ? NP + 16 * hi : NP + R1ROWS + 16 * lo);
        sl[u].tcol = kind == 2 ? hi : lo;
        sl[u].rt = kind == 2 ? lo : -1;
        T[u] = double4_f{ 0.0, 0.0, 0.0, 0.0 };
        if (kind == 2 && sl[u].valid) {                       // R2 = L^-T D^-1 from factorisation 1
#pragma unroll
            for (int r = 0; r < 4; ++r) T[u][r] = Y[(16 * lo + kq + 4 * r) * LDM + 16 * hi + l15];
        }
    }
    __syncthreads();                                          // every wave has its R2 tiles: Y may be overwritten
    // ---- R1 = [A ; b^T ; 0] L  (A symmetric, from the gram partials in global memory; L in X) ----
    auto Aext = [&](int row, int col) {                       // row NC = b^T
        double s = 0.0;
        if (col < ncol && (row < ncol || row == NC)) {
            const size_t e = row == NC ? (size_t)col * (ncol + 1) + ncol : (size_t)row * (ncol + 1) + col;
            for (int g = 0; g < G; ++g) if (chunk_used[bl * G + g]) s += Apart[((size_t)bl * G + g) * rstride + e];
        }
        return s;
    };
#pragma unroll
    for (int u = 0; u < S2; ++u) {
        const int p = u * NW + wave;
        if (sl[u].valid && p >= NLT && p < NLT + NR1 * NT) {
            const int i = (p - NLT) / NT, j = (p - NLT) % NT;
            double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
            for (int kt = j; kt < NT; ++kt) {
                double af[4], bf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    af[s] = Aext(16 * i + l15, 16 * kt + 4 * s + kq);
                    bf[s] = X[(16 * kt + 4 * s + kq) * LDM + 16 * j + l15];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bf[s], acc, 0, 0, 0);
            }
            T[u] = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) Y[(16 * i + kq + 4 * r) * LDM + 16 * j + l15] = acc[r];
        }
    }
    __syncthreads();
    // ---- W = L^T (A L) + s^2 D^-1, lower tiles ----
#pragma unroll
    for (int u = 0; u < S2; ++u) {
        const int p = u * NW + wave;
        if (sl[u].valid && p < NLT) {
            const int i = sl[u].arow >> 4, j = sl[u].tcol;
            double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
            for (int kt = i; kt < NT; ++kt) {
                double af[4], bf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kr = 16 * kt + 4 * s + kq;
                    af[s] = X[kr * LDM + 16 * i + l15];                          // A[i'][k'] = L[k][i]
                    bf[s] = (NC < NP && kr >= NC) ? 0.0 : Y[kr * LDM + 16 * j + l15];      // row NC of Y is b^T L, not a row of A L
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bf[s], acc, 0, 0, 0);
            }
            if (i == j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (kq + 4 * r == l15) acc[r] += var * sD1inv[16 * i + l15];
            }
            T[u] = acc;
        }
    }
    __syncthreads();                                          // X (L) and Y (A L) are dead: they become the outputs of sweep 2
    for (int e = tid; e < MROWS * LDM; e += 256) Y[e] = 0.0;   // R2' is upper triangular by tiles
    // ================= factorisation 2: W = L2 D2 L2^T with [A L ; b^T L] and L^-T D^-1 carried =================
    {
        auto emit = [&](int t, int row, int col, int tj0, double x, double xs) {
            if (row < NP) return;                                                          // L2 itself is not needed
            if (row < NP + R1ROWS) X[(row - NP) * LDM + col] = xs;                        // R1' D2^-1
            else if (t - NT - NR1 <= tj0) Y[(row - NP - R1ROWS) * LDM + col] = x;         // R2'
        };
        auto dsave = [&](int k, int q, double r) { sD2inv[4 * k + q] = r; };
        ldl_sweep<S2, PANROWS, NW>(T, sl, NP / 4, NP, NT + NR1 + NT, pan, wave, lane, emit, dsave, &bad);
    }
    if (bad) sBad = 1;
    // ---- [M | t] = R2' (R1' D2^-1)^T : tile (i, j), j over the NR1 row tiles of R1' (column NC carries t) ----
    double* Mg = Mall + (size_t)bl * mstride;
    for (int q = wave; q < NT * NR1; q += NW) {
        const int i = q / NR1, j = q % NR1;
        double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
        for (int kt = i; kt < NT; ++kt) {
            double af[4], bf[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                af[s] = Y[(16 * i + l15) * LDM + 16 * kt + 4 * s + kq];
                bf[s] = X[(16 * j + l15) * LDM + 16 * kt + 4 * s + kq];          // B[k'][j'] = (R1' D2^-1)[j][k]
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bf[s], acc, 0, 0, 0);
        }
        const int col = 16 * j + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * i + kq + 4 * r;
            const double v = (row < NC) ? acc[r] : 0.0;
            if (row < MP && col < MP) Mg[(size_t)row * MP + col] = col < NC ? v : 0.0;
            if (col == NC && row < MP) Mg[(size_t)MP * MP + row] = v;
        }
    }
    if (NC == MP) { /* column NC of M does not exist: nothing to clear */ }
    __syncthreads();
    if (sBad && tid == 0) atomicOr(&status[b], 4);
    // Pc = P[:, clone cols] for the in-place update (the apply kernel must read the PRE-update columns)
    double* Pc = Pcall + (size_t)bl * ystride;
    if (!zero_copy) {
        const int tx = tid & 63, ty = tid >> 6;
        for (int k = ty; k < MP; k += 4) {
            const int gk = k < NP ? sCol[k] : 0;
            const bool real = k < ncol;
            for (int r = tx; r < n; r += 64) Pc[r + (size_t)k * ld] = real ? P[r + (size_t)gk * ld] : 0.0;
        }
    }
    if (tid == 0) { m_out[bl] = ncol; nc_out[bl] = ncol; pc_base_out[bl] = zero_copy ? sCol[0] : -1; }
}

}  // namespace

// returns 0 when the window class is handled here (6 C <= 66), non-zero otherwise (caller falls back to k_info_update)
int launch_info_solve(const FactoredLaunch& L, hipStream_t st)
{
    const int ncm = 6 * L.fv.cmax;
#define SOLVE_DISPATCH(NC)                                                                                                    \
    {                                                                                                                         \
        const size_t sm = SolveCfg<NC>::lds_bytes();                                                                          \
        static bool attr_set = false;                                                                                         \
        if (!attr_set) { hipFuncSetAttribute((const void*)k_info_solve<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_set = true; } \
        hipLaunchKernelGGL(k_info_solve<NC>, dim3(L.nb), dim3(256), sm, st, L.cv, L.fv, L.b0, L.Apart, L.chunk_used, L.G, L.rstride, \
                           L.noise, L.T, L.mstride, L.Pc, L.ystride, L.dx, L.m_out, L.nc_out, L.status, L.marg_idx, L.pc_base);       \
        return 0;                                                                                                             \
    }
    if (ncm <= 36) SOLVE_DISPATCH(36)
    if (ncm <= 66) SOLVE_DISPATCH(66)
#undef SOLVE_DISPATCH
    return 1;
}
