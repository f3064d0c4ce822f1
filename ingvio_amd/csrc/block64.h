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
