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

__global__ __launch_bounds__(PROP_THREADS) void k_propagate(
    CovView cv, int b0, const double* __restrict__ Phi, const double* __restrict__ G,
    const double* __restrict__ dts, int k, const int* __restrict__ gnss_idx,
    double sg0, double sg1, double sg2, double sg3, int enable_gnss, double scb, double srw,
    const double* __restrict__ augR, int* __restrict__ status_clear)
{
    const int bl = blockIdx.y, b = b0 + bl, tid = threadIdx.x;
    if (status_clear && blockIdx.x == 0 && tid == 0) status_clear[b] = 0;
    const int n = cv.n[b], ld = cv.ldp;
    double* P = cov_ptr(cv, b);

    __shared__ double sPhi[225], sQ[225], sPG[180], sT1[225], sT2[225];
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
        for (int j = 0; j < 6; ++j) qpre[j] = P[tid + (size_t)(15 + j) * ld];
    }
    dbg_stamp(16);
    if (tid < 225) { sPhi[tid] = (tid % 15 == tid / 15) ? 1.0 : 0.0; sQ[tid] = 0.0; }
    const double* PhiB = Phi + (size_t)bl * k * 225;
    const double* GB = G + (size_t)bl * k * 180;
    const double* dtB = dts + (size_t)bl * k;
    for (int s = tid; s < k; s += PROP_THREADS) sDt[s] = dtB[s];
    // the 5x5 clock block's recursion (sequential in the steps, a few FLOP each) rides on thread 255, which has no element of the
    // 15 x 15 products: it used to run after the loop on thread 0 with everybody waiting (13 k cycles)
    int gi[5];
    double qg[5][5], Tg = 0.0;
    if (tid == PROP_THREADS - 1) {
        for (int g = 0; g < 5; ++g) gi[g] = (enable_gnss && gnss_idx) ? gnss_idx[bl * 5 + g] : -1;
        for (int a = 0; a < 5; ++a) for (int c = 0; c < 5; ++c) qg[a][c] = 0.0;
    }
    __syncthreads();
    for (int s = 0; s < k; ++s) {
        if (tid == PROP_THREADS - 1) {
            const double dt = sDt[s];
            if (gi[4] >= 0) {
                Tg += dt;
                for (int g = 0; g < 4; ++g) if (gi[g] >= 0) for (int c = 0; c < 5; ++c) qg[g][c] += dt * qg[4][c];
                for (int g = 0; g < 4; ++g) if (gi[g] >= 0) for (int r = 0; r < 5; ++r) qg[r][g] += dt * qg[r][4];
            }
            for (int a = 0; a < 5; ++a) {
                if (gi[a] < 0) continue;
                for (int c = 0; c < 5; ++c) {
                    if (gi[c] < 0) continue;
                    if (a != 4 && c != 4) qg[a][c] += dt * scb * scb + dt * dt * dt * srw * srw;   // :110
                    else if (a == 4 && c == 4) qg[a][c] += dt * srw * srw;                         // :112
                    else qg[a][c] += dt * dt * srw * srw;                                          // :114
                }
            }
        }
        if (s % PROP_KCH == 0) {
            const int cnt = min(PROP_KCH, k - s);
            // all of the chunk's loads are issued before the first LDS store (a rolled loop pays one memory latency per pass)
            constexpr int PER = (PROP_KCH * 405 + PROP_THREADS - 1) / PROP_THREADS;
            double v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * PROP_THREADS, q = e / 405, w = e - q * 405;
                v[u] = e < cnt * 405 ? (w < 225 ? PhiB[(s + q) * 225 + w] : GB[(s + q) * 180 + (w - 225)]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int e = tid + u * PROP_THREADS, w = e % 405;
                if (e < cnt * 405) sAll[e] = w < 225 ? v[u] : v[u] * (w < 225 + 45 ? sg0 : w < 225 + 90 ? sg1 : w < 225 + 135 ? sg2 : sg3);   // G_tmp, :92-96
            }
            __syncthreads();
            if (s == 0) dbg_stamp(21);
        }
        const double* sStep = sAll + (s % PROP_KCH) * 405;
        const double* sG = sStep + 225;
        const double dt = sDt[s];
        const int i = tid % 15, j = tid / 15;
        // the dot products are unrolled so that all operand reads are in flight before the first FMA (rolled, every FMA waited
        // for its own two LDS reads: 6.4 k cycles per step)
        if (tid < 180) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < 15; ++l) a += sStep[i + 15 * l] * sG[l + 15 * j];
            sPG[tid] = a;                                                   // Phi * G_tmp
        }
        if (tid < 225) {
            double a = 0.0, c = 0.0;
#pragma unroll
            for (int l = 0; l < 15; ++l) { a += sStep[i + 15 * l] * sQ[l + 15 * j]; c += sStep[i + 15 * l] * sPhi[l + 15 * j]; }
            sT1[tid] = a; sT2[tid] = c;
        }
        __syncthreads();
        if (tid < 225) {
            double a = 0.0, q = 0.0;
#pragma unroll
            for (int l = 0; l < 15; ++l) a += sT1[i + 15 * l] * sStep[j + 15 * l];
#pragma unroll
            for (int l = 0; l < 12; ++l) q += sPG[i + 15 * l] * sPG[j + 15 * l];
            sQ[tid] = a + dt * q;                                           // :51 + :97 composed
            sPhi[tid] = sT2[tid];
        }
        __syncthreads();
    }
    dbg_stamp(17);
    for (int a = tid; a < NA_MAX * NA_MAX; a += PROP_THREADS) { sPhiA[a] = 0.0; sQA[a] = 0.0; }
    if (tid == PROP_THREADS - 1) {
        for (int a = 0; a < 5; ++a) for (int c = 0; c < 5; ++c) sQg[5 * a + c] = qg[a][c];
        sQg[25] = Tg;
    }
    __syncthreads();
    // active set + GNSS clock block (thread 0, bookkeeping only)
    if (tid == 0) {
        int gi[5], na = 15;
        for (int g = 0; g < 5; ++g) gi[g] = (enable_gnss && gnss_idx) ? gnss_idx[bl * 5 + g] : -1;
        for (int a = 0; a < 15; ++a) sA[a] = a;
        int loc[5];
        for (int g = 0; g < 5; ++g) { loc[g] = -1; if (gi[g] >= 0) { loc[g] = na; sA[na++] = gi[g]; } }
        sNA = na;
        for (int a = na; a < NA_MAX; ++a) sA[a] = 0;      // padding: loads stay unconditional, Phi_A rows/cols there are zero
        const bool has_fs = gi[4] >= 0;
        const double T = sQg[25];
        const double (*qg)[5] = reinterpret_cast<const double (*)[5]>(sQg);
        for (int a = 0; a < 5; ++a) {
            if (loc[a] < 0) continue;
            sPhiA[loc[a] * NA_MAX + loc[a]] = 1.0;
            if (a < 4 && has_fs) sPhiA[loc[a] * NA_MAX + loc[4]] = T;
            for (int c = 0; c < 5; ++c) if (loc[c] >= 0) sQA[loc[a] * NA_MAX + loc[c]] = qg[a][c];
        }
    }
    if (tid < 225) {
        const int a = tid % 15, c = tid / 15;
        sPhiA[a * NA_MAX + c] = sPhi[a + 15 * c];
        sQA[a * NA_MAX + c] = sQ[a + 15 * c];
    }
    __syncthreads();
    dbg_stamp(18);
    const int na = sNA;
    // strip rows outside A
    const int r = blockIdx.x * PROP_THREADS + tid;
    bool inA = r < 15;
    for (int a = 15; a < na; ++a) inA |= (sA[a] == r);
    if (r < n && !inA) {
        double s[NA_MAX];
#pragma unroll
        for (int a = 0; a < NA_MAX; ++a) s[a] = P[r + (size_t)sA[a] * ld];      // 20 independent loads in flight
        // one output column at a time: keeps the live set at s[] + one row of Phi_A (the fully unrolled
        // 20x20 form hoists all 400 LDS reads and spills)
#pragma unroll 1
        for (int a = 0; a < na; ++a) {
            const double2* ph = reinterpret_cast<const double2*>(sPhiA + a * NA_MAX);
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int c = 0; c < NA_MAX / 2; ++c) { const double2 w = ph[c]; acc0 += s[2 * c] * w.x; acc1 += s[2 * c + 1] * w.y; }
            const double o = acc0 + acc1;
            P[r + (size_t)sA[a] * ld] = o;
            sStrip[a * (PROP_THREADS + 1) + tid] = o;
        }
    }
    lds_barrier();
    // upper strip = transpose (:89): lanes run along the active index a so that each store instruction
    // covers 16-element row segments instead of 64 different columns
    {
        const int a = tid & 15, rr0 = tid >> 4;
        for (int rr = rr0; rr < PROP_THREADS; rr += PROP_THREADS / 16) {
            const int r2 = blockIdx.x * PROP_THREADS + rr;
            bool in2 = r2 < 15;
            for (int q = 15; q < na; ++q) in2 |= (sA[q] == r2);
            if (r2 < n && !in2) {
                if (a < na) P[sA[a] + (size_t)r2 * ld] = sStrip[a * (PROP_THREADS + 1) + rr];
                if (a + 16 < na) P[sA[a + 16] + (size_t)r2 * ld] = sStrip[(a + 16) * (PROP_THREADS + 1) + rr];
            }
        }
    }
    dbg_stamp(19);
    // A x A block, tile 0 only
    if (blockIdx.x == 0) {
        for (int e = tid; e < na * na; e += PROP_THREADS) {
            const int a = e / na, c = e % na;
            sX[a * NA_MAX + c] = P[sA[a] + (size_t)sA[c] * ld];
        }
        lds_barrier();
        for (int e = tid; e < na * na; e += PROP_THREADS) {
            const int a = e / na, c = e % na;
            double acc = 0.0;
            for (int l = 0; l < na; ++l) acc += sPhiA[a * NA_MAX + l] * sX[l * NA_MAX + c];
            sY[a * NA_MAX + c] = acc;
        }
        lds_barrier();
        for (int e = tid; e < na * na; e += PROP_THREADS) {
            const int a = e / na, c = e % na;
            double acc = 0.0;
            for (int l = 0; l < na; ++l) acc += sY[a * NA_MAX + l] * sPhiA[c * NA_MAX + l];
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
__global__ __launch_bounds__(256) void k_augment(CovView cv, int b0, const double* __restrict__ Rs)
{
    __shared__ double sJP[6][21];
    augment_filter<256>(cv, b0 + blockIdx.x, Rs + blockIdx.x * 9, sJP);
}

// ---------------------------------------------------------------------------------------------
// K12: StateManager::marginalize covariance part (StateManager.cpp:163-177), out of place into
// the filter's other ping-pong buffer; k_post then flips `cur` and shrinks n.  grid = (col tiles, nb)
// ---------------------------------------------------------------------------------------------
#define MARG_COLS 8
__global__ __launch_bounds__(256) void k_marginalize(CovView cv, int b0, const int* __restrict__ idxs, int size)
{
    const int bl = blockIdx.y, b = b0 + bl, tid = threadIdx.x;
    const int idx = idxs[bl];
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
            dst[i + (size_t)j * ld] = src[si + (size_t)sj * ld];
        }
    }
}

__global__ void k_post_marg(CovView cv, int b0, int nb, const int* __restrict__ idxs, int size)
{
    const int bl = blockIdx.x * blockDim.x + threadIdx.x;
    if (bl >= nb) return;
    if (idxs[bl] < 0) return;
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

// snapshot / restore of every filter's live covariance (benchmark hygiene: each timed step starts
// from the same prior).  grid = (col tiles, B).
__global__ __launch_bounds__(256) void k_snapshot(CovView cv, double* __restrict__ snap, int* __restrict__ n_snap)
{
    const int b = blockIdx.y, tid = threadIdx.x, n = cv.n[b], ld = cv.ldp;
    const double* src = cov_ptr(cv, b);
    double* dst = snap + (size_t)b * ld * ld;
    for (int jj = 0; jj < MARG_COLS; ++jj) {
        const int j = blockIdx.x * MARG_COLS + jj;
        if (j >= n) break;
        for (int i = tid; i < n; i += 256) dst[i + (size_t)j * ld] = src[i + (size_t)j * ld];
    }
    if (blockIdx.x == 0 && tid == 0) n_snap[b] = n;
}
__global__ __launch_bounds__(256) void k_restore(CovView cv, const double* __restrict__ snap, const int* __restrict__ n_snap)
{
    const int b = blockIdx.y, tid = threadIdx.x, n = n_snap[b], ld = cv.ldp;
    const double* src = snap + (size_t)b * ld * ld;
    double* dst = cv.Pbase + (size_t)b * ld * ld;          // half 0
    for (int jj = 0; jj < MARG_COLS; ++jj) {
        const int j = blockIdx.x * MARG_COLS + jj;
        if (j >= n) break;
        for (int i = tid; i < n; i += 256) dst[i + (size_t)j * ld] = src[i + (size_t)j * ld];
    }
}
// Partial restore after a fused frame step (propagate + clone + out-of-place update/marginalise): half 0 still holds
// the snapshot except for the rows/columns of the propagation's active set A = {0..14} + clock states (the six clone
// rows/cols lie beyond n_snap).  grid = (ceil(n_cap/256), B), thread = row/column index.
__global__ __launch_bounds__(256) void k_restore_strips(CovView cv, const double* __restrict__ snap, const int* __restrict__ n_snap,
                                                        const int* __restrict__ gnss_idx)
{
    const int b = blockIdx.y, n = n_snap[b], ld = cv.ldp;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) { cv.cur[b] = 0; cv.n[b] = n; }           // nothing in this kernel reads cur / n
    if (r >= n) return;
    const double* src = snap + (size_t)b * ld * ld;
    double* dst = cv.Pbase + (size_t)b * ld * ld;          // half 0
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
__global__ void k_post_restore(CovView cv, const int* __restrict__ n_snap)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < cv.B) { cv.cur[b] = 0; cv.n[b] = n_snap[b]; }
}

// ---------------------------------------------------------------------------------------------
#include "launch_ekf.h"
void launch_propagate(CovView cv, int b0, int nb, int n_cap, const double* Phi, const double* G, const double* dt, int k,
                      const int* gnss_idx, const double sigma[4], int enable_gnss, double scb, double srw, hipStream_t st,
                      const double* augR, int* status_clear)
{
    const int tiles = (n_cap + PROP_THREADS - 1) / PROP_THREADS;
    const bool fuse = augR && tiles == 1;
    hipLaunchKernelGGL(k_propagate, dim3(tiles, nb), dim3(PROP_THREADS), 0, st, cv, b0, Phi, G, dt, k, gnss_idx,
                       sigma[0], sigma[1], sigma[2], sigma[3], enable_gnss, scb, srw, fuse ? augR : nullptr, status_clear);
    if (augR && !fuse) hipLaunchKernelGGL(k_augment, dim3(nb), dim3(256), 0, st, cv, b0, augR);
}
void launch_augment(CovView cv, int b0, int nb, const double* R, hipStream_t st)
{
    hipLaunchKernelGGL(k_augment, dim3(nb), dim3(256), 0, st, cv, b0, R);
}
void launch_marginalize(CovView cv, int b0, int nb, int n_cap, const int* idx, int size, hipStream_t st)
{
    hipLaunchKernelGGL(k_marginalize, dim3((n_cap + MARG_COLS - 1) / MARG_COLS, nb), dim3(256), 0, st, cv, b0, idx, size);
    hipLaunchKernelGGL(k_post_marg, dim3((nb + 255) / 256), dim3(256), 0, st, cv, b0, nb, idx, size);
}
void launch_post_marg(CovView cv, int b0, int nb, const int* idx, int size, hipStream_t st)
{
    hipLaunchKernelGGL(k_post_marg, dim3((nb + 255) / 256), dim3(256), 0, st, cv, b0, nb, idx, size);
}
void launch_append(CovView cv, int b0, int nb, int size, const double* blk, hipStream_t st)
{
    hipLaunchKernelGGL(k_append, dim3(nb), dim3(256), 0, st, cv, b0, size, blk);
}
void launch_copy_ints(int* dst, const int* src, int count, hipStream_t st)
{
    hipLaunchKernelGGL(k_copy_ints, dim3((count + 255) / 256), dim3(256), 0, st, dst, src, count);
}
void launch_snapshot(CovView cv, int n_cap, double* snap, int* n_snap, hipStream_t st)
{
    hipLaunchKernelGGL(k_snapshot, dim3((n_cap + MARG_COLS - 1) / MARG_COLS, cv.B), dim3(256), 0, st, cv, snap, n_snap);
}
void launch_restore_strips(CovView cv, int n_cap, const double* snap, const int* n_snap, const int* gnss_idx, hipStream_t st)
{
    hipLaunchKernelGGL(k_restore_strips, dim3((n_cap + 255) / 256, cv.B), dim3(256), 0, st, cv, snap, n_snap, gnss_idx);
}
void launch_restore(CovView cv, int n_cap, const double* snap, const int* n_snap, hipStream_t st)
{
    hipLaunchKernelGGL(k_restore, dim3((n_cap + MARG_COLS - 1) / MARG_COLS, cv.B), dim3(256), 0, st, cv, snap, n_snap);
    hipLaunchKernelGGL(k_post_restore, dim3((cv.B + 255) / 256), dim3(256), 0, st, cv, n_snap);
}

int dbg_read_cov(long long* out, int n) { return dbg_read_local(out, n); }
