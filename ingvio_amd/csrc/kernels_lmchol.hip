// kernels_lmchol.hip — the SPD solve of a stacked update with up to 256 rows, ONE workgroup per filter, matrix resident in
// registers (gfx950 only).  Used by the batched SLAM-landmark update (LandmarkUpdate.cpp:32-149: S = H P H^T + s^2 I with 4 rows
// per stereo landmark, 208 rows at 52 landmarks) and by generic ekfUpdate calls of that size (StateManager.cpp:359-411).
//
// kernels_chol.hip factorises out of HBM / L2 with one launch per 32-column panel and one workgroup per 32 x 32 block: made for
// a handful of large matrices (config 5).  With hundreds of filters every one of its workgroups is a memory round trip deep and
// does 40 MFMA: 512 filters x 208 rows took 0.34 ms (factor) + 0.35 ms (carried rows) + 0.07 ms (dx).  Here:
//
//   k_lm_factor   S = U^T U by 16-column panels, the upper triangle of S dealt round-robin as 16 x 16 tiles to the 8 waves of the
//                 workgroup and kept in registers in the MFMA C/D layout (lane (kq, l15), register r = element (kq + 4 r, l15)).
//                 That layout IS the B operand of v_mfma_f64_16x16x4 when k-step r covers rows kq + 4 r, and also the A operand
//                 of a product with the tile transposed - so, per panel k:
//                     (A) the wave that owns S_kk factorises it (one lane per row, identity carried: T = L_kk^-T falls out)
//                     (B) U_kb = L_kk^-1 S_kb      : A = L_kk^-1 from LDS, B = the tile's registers           (b > k)
//                     (C) S_ab -= U_ka^T U_kb       : A and B both straight from the panel buffer in LDS       (k < a <= b)
//                 two workgroup barriers per panel, no transposes.  The residual rides along on the vector ALU: z = L^-1 res.
//                 Output: U and the L_kk^-1 as tiles in that register layout (lane-linear: 512-byte rows), z.
//   k_lm_carry    V = L^-1 (P H^T)^T by forward substitution, one wave per 16 state rows: V_k = L_kk^-1 (C_k - sum_p<k U_pk^T V_p)
//                 with the V_p of the block resident in registers (again C/D layout = B operand) and the U tiles streamed as A
//                 operands; writes Y = V^T in the layout k_downdate64 reads, and dx = V^T z.
// Parity: tests/test_landmark_batch.py, tests/test_landmark_path.py, tests/test_gpu_parity.py (dense-H route) - all through the C ABI.
#include "launch_chol.h"
#include "dev_common.h"

typedef double double4_f __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ double rsqrt_full(double p)                  // v_rsq_f64 + two Newton steps
{
    double y = __builtin_amdgcn_rsq(p);
    double e = fma(-p * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);
    e = fma(-p * y, y, 1.0);
    y = fma(y * e, 0.5, y);
    return y;
}

__device__ __forceinline__ double lane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

constexpr int tri_tiles(int ntm) { return ntm * (ntm + 1) / 2; }
// row-major index of tile (a, b), a <= b, in the upper triangle of an ntm x ntm tile grid
__host__ __device__ constexpr int tri_index(int ntm, int a, int b) { return a * ntm - a * (a - 1) / 2 + (b - a); }

// ---------------------------------------------------------------------------------------------
// Factorisation of one 16 x 16 diagonal tile by ONE wave.  sDI = [D; I] (32 x 17): lane l < 16 holds row l of D, lane 16 + i
// row i of the identity carried below it (lanes 32..63 shadow lanes 0..31).  Elimination in LDL^T form, scaled to Cholesky at
// the end, the column update of pivot j - 1 software-pipelined behind the pivot chain of j (the scheme of factor32 in
// kernels_chol.hip).  Returns the scaled rows: lanes 0..15 rows of L, lanes 16..31 rows of T = L^-T.  bad: a pivot of this
// lane's row was not positive (its column is zeroed).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void factor16(double (*sDI)[17], double (*sC)[16], int lane, double (&d)[16], bool& bad)
{
    const int row = lane & 31;
#pragma unroll
    for (int j = 0; j < 16; ++j) d[j] = sDI[row][j];
    double m_prev = 0.0, pv = 1.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double lc[16];
        if (j >= 1) {
#pragma unroll
            for (int c = j + 1; c < 16; ++c) lc[c] = sC[(j - 1) & 1][c];
        }
        if (lane < 16) sC[j & 1][lane] = d[j];
        asm volatile("" ::: "memory");
        const double p = lane_f64(d[j], j);
        const bool ok = p > 0.0;
        const double rinv = ok ? fast_rcp(p) : 0.0;
        const double m = d[j] * rinv;
        if (row == j) pv = d[j];
        if (j < 15) {
            const double x = lane_f64(d[j], j + 1);
            d[j + 1] = fma(-m, x, d[j + 1]);
        }
        asm volatile("" ::: "memory");
        if (j >= 1) {
#pragma unroll
            for (int c = j + 1; c < 16; ++c) {
                d[c] = fma(-m_prev, lc[c], d[c]);
                asm volatile("" : "+v"(d[c]));
            }
        }
        m_prev = m;
    }
    const bool okp = pv > 0.0;
    if (lane < 16) sC[0][lane] = okp ? rsqrt_full(pv) : 0.0;
    asm volatile("" ::: "memory");
    bad = lane < 16 && !okp;
#pragma unroll
    for (int j = 0; j < 16; ++j) d[j] *= sC[0][j];
}

// ---------------------------------------------------------------------------------------------
// grid = filters, 512 threads.
// ---------------------------------------------------------------------------------------------
template <int NTM, bool WL>                                              // WL: also write L in matrix form (a template flag: as a run-time one it cost the landmark instantiation 512 B of spills)
__global__ __launch_bounds__(512) void k_lm_factor(LmCholArgs a)
{
    constexpr int NTILES = tri_tiles(NTM), NW = 8, NS = (NTILES + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) double pan[NTM][4][64];      // row panel k of U: tile (k, b) at pan[b], register layout
    __shared__ double sDI[32][17];
    __shared__ double sT[16][17];                                        // T = L_kk^-T: sT[i][j] = L_kk^-1 [j][i]
    __shared__ __attribute__((aligned(16))) double sC[2][16];
    __shared__ double sR[16 * NTM], sZ[16];
    __shared__ int sBad;
    const int bl = blockIdx.x;
    if (a.m[bl] == 0) return;
    const int m = a.m_fixed > 0 ? a.m_fixed : a.m[bl];
    const int nt = min(NTM, (m + 15) >> 4);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const double* X = a.X + (size_t)bl * a.xs;
    double* Ug = a.U + (size_t)bl * a.us;
    double* zg = Ug + (size_t)(NTILES + NTM) * 256;
    const int ldx = a.ldx;
    // compact row R of the stacked system = row sMap[R] of X (the landmark front gates AFTER it built S for every candidate, so
    // the accepted rows are picked here; no map: identity); a negative entry is a padding row (unit diagonal)
    __shared__ int sMap[16 * NTM];
    for (int e = tid; e < 16 * NTM; e += 512) sMap[e] = e < 16 * nt ? (a.rowmap ? a.rowmap[(size_t)bl * a.rm_stride + e] : e) : -1;
    if (tid == 0) sBad = 0;
    __syncthreads();
    // this wave's tiles: t = wave + 8 s
    int sa[NS], sb[NS];
    double4_f T[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int t = wave + NW * s, ta = 0;
        if (t >= NTILES) { sa[s] = NTM; sb[s] = NTM; }
        else {
            while (t >= NTM - ta) { t -= NTM - ta; ++ta; }
            sa[s] = ta; sb[s] = ta + t;
        }
        // S is symmetric, its LOWER triangle is what every producer fills (the GEMM's lower blocks, k_add_noise, k_lm_front): element
        // (R1, R2) is read at row max, column min (for a < b: 16 lanes contiguous), mirrored inside diagonal tiles
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int R1 = 16 * sa[s] + kq + 4 * r, R2 = 16 * sb[s] + l15;
            double v = R1 == R2 ? 1.0 : 0.0;
            if (sb[s] < nt) {
                const int o1 = sMap[R1], o2 = sMap[R2];
                if (o1 >= 0 && o2 >= 0) v = X[(size_t)max(o1, o2) + (size_t)min(o1, o2) * ldx];
            }
            T[s][r] = v;
        }
    }
    for (int e = tid; e < 16 * NTM; e += 512) sR[e] = (a.res_row >= 0 && sMap[e] >= 0) ? X[(size_t)a.res_row + (size_t)sMap[e] * ldx] : 0.0;
    double* Lg = WL ? a.Y + (size_t)bl * a.xs : nullptr;                 // L in matrix form (rows [0, 16 nt) of Y), for callers that multiply with it
    __syncthreads();
    dbg_stamp(0);
#pragma unroll 1
    for (int k = 0; k < nt; ++k) {
        if (k == 1) dbg_stamp(1);
        // ---- (A) the diagonal tile
        bool mine = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (sa[s] == k && sb[s] == k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sDI[kq + 4 * r][l15] = T[s][r];
                mine = true;
            }
        }
        if (mine) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = lane + 64 * u; sDI[16 + (e >> 4)][e & 15] = (e >> 4) == (e & 15) ? 1.0 : 0.0; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            double d[16]; bool bad;
            factor16(sDI, sC, lane, d, bad);
            if (lane >= 16 && lane < 32) {
#pragma unroll
                for (int j = 0; j < 16; ++j) sT[lane - 16][j] = d[j];
            }
            if (WL && lane < 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j) Lg[(size_t)(16 * k + lane) + (size_t)(16 * k + j) * ldx] = j <= lane ? d[j] : 0.0;
            }
            if (__any(bad) && lane == 0) sBad = 1;
        }
        if (k == 1) dbg_stamp(2);
        lds_barrier();
        if (k == 1) dbg_stamp(3);
        // ---- (B) row panel k
        double A[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) A[r] = sT[kq + 4 * r][l15];          // A[m = l15][k = kq + 4 r] = L_kk^-1 [l15][kq + 4 r]
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Ug[(size_t)(NTILES + k) * 256 + 64 * r + lane] = A[r];
        }
        if (wave == NW - 1 && lane < 16) {
            double z = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) z = fma(sT[j][lane], sR[16 * k + j], z);
            sZ[lane] = z;
            zg[16 * k + lane] = z;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (sa[s] == k && sb[s] > k && sb[s] < nt) {
                double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r], T[s][r], acc, 0, 0, 0);
                T[s] = acc;
                double* ug = Ug + (size_t)tri_index(NTM, k, sb[s]) * 256 + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pan[sb[s]][r][lane] = acc[r]; ug[64 * r] = acc[r]; }
                if (WL) {                                                // L[16 b + j][16 k + i] = U_kb[i][j]; the mirrored block is zero
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Lg[(size_t)(16 * sb[s] + l15) + (size_t)(16 * k + kq + 4 * r) * ldx] = acc[r];
                        Lg[(size_t)(16 * k + l15) + (size_t)(16 * sb[s] + kq + 4 * r) * ldx] = 0.0;
                    }
                }
            }
        }
        lds_barrier();
        if (k == 1) dbg_stamp(4);
        // ---- (C) trailing update
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (sa[s] > k && sb[s] < nt) {
                double4_f acc = T[s];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pan[sa[s]][r][lane], pan[sb[s]][r][lane], acc, 0, 0, 0);
                T[s] = acc;
            }
        }
        if (tid < 16 * (nt - k - 1)) {                                   // res_b -= U_kb^T z_k
            const int b = k + 1 + (tid >> 4), j = tid & 15;
            double v = sR[16 * b + j];
#pragma unroll
            for (int i = 0; i < 16; ++i) v = fma(-pan[b][i >> 2][(i & 3) * 16 + j], sZ[i], v);
            sR[16 * b + j] = v;
        }
        if (k == 1) dbg_stamp(5);
    }
    dbg_stamp(6);
    __syncthreads();
    if (tid == 0 && sBad) atomicOr(&a.status[bl], a.fail_bit);
}

// ---------------------------------------------------------------------------------------------
// grid = filters x blocks of 64 state rows, 256 threads: wave w = 16-row block 4 blk + w.  The four waves need the same
// A operands (the U tiles of column k and L_kk^-1, k + 1 tiles per step): the workgroup stages them through LDS, double-buffered,
// the next step's tiles in flight (one double per thread and tile) under this step's MFMAs; one barrier per step.
// ---------------------------------------------------------------------------------------------
template <int NTM>
__global__ __launch_bounds__(256) void k_lm_carry(LmCholArgs a)
{
    constexpr int NTILES = tri_tiles(NTM);
    __shared__ __attribute__((aligned(16))) double sA[2][NTM + 1][256];
    // workgroup -> (filter, block of 64 state rows) with the blocks of one filter on ONE XCD: its factor tiles come from HBM once
    const bool gen = a.carried_rows > 0;
    const int nblk = ((gen ? a.carried_rows : a.cv.ldp) + 63) / 64;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bl = (slot / nblk) * 8 + xcd, blk = slot % nblk;
    if (bl >= a.nb) return;
    if (a.m[bl] == 0) return;
    const int m = a.m_fixed > 0 ? a.m_fixed : a.m[bl];
    const int b = a.b0 + bl, n = gen ? a.carried_rows : a.cv.n[b], ld = gen ? a.carried_rows : a.cv.ldp;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int cb = 4 * blk + wave, j = 16 * cb + l15;
    double* dx = a.dx ? a.dx + (size_t)b * ld : nullptr;
    if (a.status[bl] & a.fail_bit) {                                     // S not positive definite: no update
        if (dx && kq == 0 && j < ld) dx[j] = 0.0;
        return;
    }
    const bool on = 16 * cb < n;                                         // (a wave beyond the state still helps staging the tiles)
    const int nt = min(NTM, (m + 15) >> 4), ldx = a.ldx;
    const double* C0 = a.X + (size_t)bl * a.xs + a.mc + (on ? j : 0);   // (P H^T)[j][R] at C0[R ldx]
    double* Y0 = a.Y + (size_t)bl * a.xs + a.mc + j;
    const double* Ug = a.U + (size_t)bl * a.us + tid;
    const double* zg = a.U + (size_t)bl * a.us + (size_t)(NTILES + NTM) * 256;
    __shared__ int sMap[16 * NTM];                                       // compact row -> row of X (see k_lm_factor)
    for (int e = tid; e < 16 * NTM; e += 256) sMap[e] = e < 16 * ((m + 15) >> 4) ? (a.rowmap ? a.rowmap[(size_t)bl * a.rm_stride + e] : e) : -1;
    __syncthreads();
    double4_f V[NTM];
    double dxa = 0.0;
    double pre[NTM + 1], cn[4];
    auto fetch_c = [&](int k, double (&c)[4]) {                          // the block's rows of P H^T for step k
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = sMap[16 * k + kq + 4 * r];
            c[r] = o >= 0 ? C0[(size_t)o * ldx] : 0.0;
        }
    };
    auto fetch = [&](int k) {                                            // tiles (p, k), p < k, then L_kk^-1
#pragma unroll
        for (int p = 0; p < NTM; ++p) if (p < k) pre[p] = Ug[(size_t)tri_index(NTM, p, k) * 256];
        pre[NTM] = Ug[(size_t)(NTILES + k) * 256];
        fetch_c(k, cn);
    };
    // The first PRO steps are too short to hide a memory round trip each (step k is 4 (k + 1) MFMAs): their k + 1 tiles and their
    // rows of P H^T are all requested up front and staged in buffer 0 (slot k (k + 1) / 2 + p); from step PRO on the tiles of
    // step k sit in buffer (k - PRO + 1) & 1, requested one step ahead.
    constexpr int PRO = NTM >= 12 ? 4 : (NTM >= 8 ? 3 : 2);
    static_assert(NTM >= PRO && PRO * (PRO + 1) / 2 <= NTM + 1, "prologue slots");
    auto bufidx = [](int k) { return (k - PRO + 1) & 1; };             // step PRO -> buffer 1 (buffer 0 still serves the prologue steps)
    double cpro[PRO][4];
    {
        double ppre[PRO * (PRO + 1) / 2];
#pragma unroll
        for (int k = 0; k < PRO; ++k) {
#pragma unroll
            for (int p = 0; p <= k; ++p)
                ppre[k * (k + 1) / 2 + p] = p < k ? Ug[(size_t)tri_index(NTM, p, k) * 256] : Ug[(size_t)(NTILES + k) * 256];
            fetch_c(k, cpro[k]);
        }
#pragma unroll
        for (int e = 0; e < PRO * (PRO + 1) / 2; ++e) sA[0][e][tid] = ppre[e];
    }
    if (PRO < nt) fetch(PRO);
    lds_barrier();
    // A step's results are stored at the top of the NEXT step, before that step's loads are requested: stores that follow loads in
    // program order turn the wait for the loads into a full drain of the store queue (one counter for both on gfx9).
    auto store_y = [&](int k) {
        if (on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Y0[(size_t)(16 * k + kq + 4 * r) * ldx] = V[k][r];
        }
    };
#pragma unroll
    for (int k = 0; k < NTM; ++k) {
        if (k < nt) {
            double4_f acc, acc2 = { 0.0, 0.0, 0.0, 0.0 };
            if (k < PRO) acc = double4_f{ cpro[k][0], cpro[k][1], cpro[k][2], cpro[k][3] };
            else acc = double4_f{ cn[0], cn[1], cn[2], cn[3] };
            if (k > 0) store_y(k - 1);
            if (k >= PRO && k + 1 < nt) fetch(k + 1);
            const double (*buf)[256] = k < PRO ? sA[0] + k * (k + 1) / 2 : sA[bufidx(k)];
            const int tslot = k < PRO ? k : NTM;
#pragma unroll
            for (int p = 0; p < k; ++p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (p & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-buf[p][64 * r + lane], V[p][r], acc2, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-buf[p][64 * r + lane], V[p][r], acc, 0, 0, 0);
                }
            }
            if (k > 1) acc += acc2;
            double4_f y = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int r = 0; r < 4; ++r) y = __builtin_amdgcn_mfma_f64_16x16x4f64(buf[tslot][64 * r + lane], acc[r], y, 0, 0, 0);
            V[k] = y;
            if (dx && on) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dxa = fma(y[r], zg[16 * k + kq + 4 * r], dxa);
            }
            if (k >= PRO - 1 && k + 1 < nt) {                            // the tiles of step k + 1 into its buffer
#pragma unroll
                for (int p = 0; p < NTM; ++p) if (p < k + 1) sA[bufidx(k + 1)][p][tid] = pre[p];
                sA[bufidx(k + 1)][NTM][tid] = pre[NTM];
                lds_barrier();
            }
            if (k + 1 >= nt) store_y(k);                                 // the last step's own results
        }
    }
    dxa += __shfl_xor(dxa, 16, WAVE);
    dxa += __shfl_xor(dxa, 32, WAVE);
    if (dx && kq == 0 && j < ld) dx[j] = (on && j < n) ? dxa : 0.0;
}

template <int NTM>
void launch_t(const LmCholArgs& a, hipStream_t st)
{
    if (a.write_L) hipLaunchKernelGGL((k_lm_factor<NTM, true>), dim3(a.nb), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_lm_factor<NTM, false>), dim3(a.nb), dim3(512), 0, st, a);
    const int crows = a.carried_rows > 0 ? a.carried_rows : a.cv.ldp;
    hipLaunchKernelGGL((k_lm_carry<NTM>), dim3((a.nb + 7) / 8 * 8 * ((crows + 63) / 64)), dim3(256), 0, st, a);
}

int pick_ntm(int mc)
{
    const int t = (mc + 15) / 16;
    return t <= 4 ? 4 : (t <= 8 ? 8 : (t <= 12 ? 12 : (t <= 14 ? 14 : 16)));
}

}  // namespace

int dbg_read_lmchol(long long* out, int n) { return dbg_read_local(out, n); }

size_t lm_chol_ws_doubles(int mc)
{
    const int ntm = pick_ntm(mc);
    return (size_t)(tri_tiles(ntm) + ntm) * 256 + (size_t)16 * ntm;
}

void launch_lm_chol(const LmCholArgs& a, hipStream_t st)
{
    switch (pick_ntm(a.mc)) {
    case 4: launch_t<4>(a, st); break;
    case 8: launch_t<8>(a, st); break;
    case 12: launch_t<12>(a, st); break;
    case 14: launch_t<14>(a, st); break;
    default: launch_t<16>(a, st); break;
    }
}
