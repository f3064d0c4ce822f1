// block64.h — one 64 x 64 block of C = A B^T on v_mfma_f64_16x16x4 by a 256-thread workgroup (four waves, a 32 x 32 quadrant each):
// the two 64-row operand panels are staged through LDS 16 columns of K at a time, so an operand element is read from L2 once per
// workgroup instead of once per 16 x 16 tile.  loadA(r, k) / loadB(r, k): element k of operand row r (0..63) of this block, zero
// outside the problem.  Quadrant (wi, wj) = (wave >> 1, wave & 1); the accumulators are the four 16 x 16 tiles of the quadrant in
// the MFMA C/D layout (lane (kq, l15), reg r <-> row kq + 4 r, column l15).  Every thread must call it (two LDS barriers per K chunk: the next chunk's global loads stay in flight across them).
#pragma once
#include "dev_common.h"

typedef double b64_d4 __attribute__((ext_vector_type(4)));

struct Block64Lds { double a[16][68]; double b[16][68]; };

template <class FA, class FB>
__device__ __forceinline__ void block64_mma(Block64Lds& s, int K, FA loadA, FB loadB, bool quad_on, b64_d4 (&c)[4])
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const int sr = tid & 63, sk = tid >> 6;                              // staging role: row of the panel, first of its 4 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = b64_d4{ 0.0, 0.0, 0.0, 0.0 };
    double va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { va[u] = loadA(sr, sk + 4 * u); vb[u] = loadB(sr, sk + 4 * u); }
    for (int k0 = 0; k0 < K; k0 += 16) {
        lds_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.a[sk + 4 * u][sr] = va[u]; s.b[sk + 4 * u][sr] = vb[u]; }
        lds_barrier();
        if (k0 + 16 < K) {                                               // the next chunk travels under this chunk's MFMAs
#pragma unroll
            for (int u = 0; u < 4; ++u) { va[u] = loadA(sr, k0 + 16 + sk + 4 * u); vb[u] = loadB(sr, k0 + 16 + sk + 4 * u); }
        }
        if (quad_on) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double a0 = s.a[4 * s4 + kq][32 * wi + l15], a1 = s.a[4 * s4 + kq][32 * wi + 16 + l15];
                const double q0 = s.b[4 * s4 + kq][32 * wj + l15], q1 = s.b[4 * s4 + kq][32 * wj + 16 + l15];
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, q0, c[0], 0, 0, 0);
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, q1, c[1], 0, 0, 0);
                c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, q0, c[2], 0, 0, 0);
                c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, q1, c[3], 0, 0, 0);
            }
        }
    }
}

// the quadrant's 32 x 32 values, row-fast, into sv[32][33] (wave-private); call with the accumulators of block64_mma
__device__ __forceinline__ void block64_to_lds(const b64_d4 (&c)[4], double (*sv)[33])
{
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sv[kq + 4 * r][l15] = c[0][r]; sv[kq + 4 * r][16 + l15] = c[1][r];
        sv[16 + kq + 4 * r][l15] = c[2][r]; sv[16 + kq + 4 * r][16 + l15] = c[3][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- round 5: the same block product with the operands prefetched TWO chunks ahead (two register sets) and a double-buffered
//      LDS stage, i.e. ONE barrier per 16 columns of K.  block64_mma above issues the next chunk's loads only one chunk (16 MFMAs,
//      about 0.4 us) before it needs them: with L2 / HBM latencies of 0.5-2 us under load the four waves of a workgroup stood at
//      `s_waitcnt vmcnt` most of the time (k_apply_T64: 111 us where its MFMAs need 37).
struct Block64Lds2 { double a[2][16][68]; double b[2][16][68]; };

// KA / KB: staging role of the operand.  false: a thread takes ONE ROW's element at 4 values of k, consecutive threads consecutive
// ROWS (coalesced when the operand is row-contiguous in memory: element (r, k) at r + k ld); true: a thread takes 4 CONSECUTIVE k of
// one row, four threads a row's 16 k (coalesced when the operand is k-contiguous: element (r, k) at k + r ld).
template <bool KA = false, bool KB = false, class FA, class FB>
__device__ __forceinline__ void block64_mma2(Block64Lds2& s, int K, FA loadA, FB loadB, bool quad_on, b64_d4 (&c)[4])
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const int ar = KA ? tid >> 2 : tid & 63, ak0 = KA ? 4 * (tid & 3) : tid >> 6, aks = KA ? 1 : 4;      // row, first k, k step of this thread's 4 elements
    const int br = KB ? tid >> 2 : tid & 63, bk0 = KB ? 4 * (tid & 3) : tid >> 6, bks = KB ? 1 : 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = b64_d4{ 0.0, 0.0, 0.0, 0.0 };
    const int nch = (K + 15) >> 4;
    if (nch == 0) return;
    double ra0[4], rb0[4], ra1[4], rb1[4];
    auto fetch0 = [&](int ch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { ra0[u] = loadA(ar, 16 * ch + ak0 + aks * u); rb0[u] = loadB(br, 16 * ch + bk0 + bks * u); }
    };
    auto fetch1 = [&](int ch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { ra1[u] = loadA(ar, 16 * ch + ak0 + aks * u); rb1[u] = loadB(br, 16 * ch + bk0 + bks * u); }
    };
    auto put0 = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.a[buf][ak0 + aks * u][ar] = ra0[u]; s.b[buf][bk0 + bks * u][br] = rb0[u]; }
    };
    auto put1 = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.a[buf][ak0 + aks * u][ar] = ra1[u]; s.b[buf][bk0 + bks * u][br] = rb1[u]; }
    };
    auto mma = [&](int buf) {
        if (!quad_on) return;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const double a0 = s.a[buf][4 * s4 + kq][32 * wi + l15], a1 = s.a[buf][4 * s4 + kq][32 * wi + 16 + l15];
            const double q0 = s.b[buf][4 * s4 + kq][32 * wj + l15], q1 = s.b[buf][4 * s4 + kq][32 * wj + 16 + l15];
            c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, q0, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, q1, c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, q0, c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, q1, c[3], 0, 0, 0);
        }
    };
    // every fetch is UNCONDITIONAL (past the end the last chunk is fetched again and never used): a branch around a group of loads
    // makes the number of loads in flight unknown at the join, and the compiler then drains the queue (vmcnt(0)) at the next stage
    const int last = nch - 1;
    fetch0(0);
    fetch1(min(1, last));
    put0(0);
    fetch0(min(2, last));
    lds_barrier();
    int k = 0;
    for (; k + 1 < nch; k += 2) {
        mma(0);                                                         // chunk k from buffer 0
        put1(1);                                                        // chunk k + 1
        fetch1(min(k + 3, last));
        lds_barrier();
        mma(1);                                                         // chunk k + 1 from buffer 1
        put0(0);                                                        // chunk k + 2 (or a repeat of the last)
        fetch0(min(k + 4, last));
        lds_barrier();
    }
    if (k < nch) { mma(0); lds_barrier(); }
}
