// dev_common.h — shared declarations of libingvio_hip.so's device side (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

// Device view of a context: B filters, two ping-pong covariance buffers per filter.
// Filter b's current covariance = Pbase + (cur[b] * B + b) * pstride, column-major, ld = ldp.
struct CovView {
    double* Pbase;
    int* cur;     // [B] 0/1: which ping-pong half is live
    int* n;       // [B] current state dimension
    int ldp;
    int B;
    size_t pstride;   // doubles between consecutive filters' covariances: ldp * ldp + a pad that is NOT a power of two (see ingvio_ctx_create)
};

__device__ __forceinline__ double* cov_ptr(const CovView& v, int b)
{
    return v.Pbase + ((size_t)v.cur[b] * v.B + b) * v.pstride;
}
__device__ __forceinline__ double* cov_alt_ptr(const CovView& v, int b)
{
    return v.Pbase + ((size_t)(1 - v.cur[b]) * v.B + b) * v.pstride;
}

// MSCKF frame inputs, SoA over the batch (strides are the context maxima).
struct FrameView {
    const int* clone_idx;               // [B][cmax]
    const double* clone_R;              // [B][cmax][9]
    const double* clone_p;              // [B][cmax][3]
    const int* n_clones;                // [B]
    const int* n_feat;                  // [B]
    const double* pf;                   // [B][fmax][3]
    const int* anchor;                  // [B][fmax]
    const unsigned long long* obs_mask; // [B][fmax]
    const double* uv;                   // [B][fmax][cmax][4]
    const int* dof;                     // [B][fmax]
    int cmax, fmax;
};

struct MsckfOpts {
    double R_lr[9];
    double t_lr[3];
    double var;            // visual_noise^2
    int max_accept;
    int selected_variant;
    const double* chi2;    // device table
    int chi2_len;
};

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
    return x;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a workgroup-scope release/acquire fence around
// s_barrier: it also waits for the thread's outstanding GLOBAL loads and stores (s_waitcnt vmcnt(0)) - after a phase that stores
// results to HBM that is microseconds of store latency per barrier for nothing (k_propagate's fused clone: 42 k -> 8 k cycles).
// Use only where no thread reads GLOBAL memory another thread of the workgroup wrote before the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// value of lane (i - N) mod 16 of the same 16-lane row (DPP row_ror:N on both halves of the double): no LDS traffic, unlike __shfl.
// bound_ctrl with full row / bank masks: every lane is written, so the compiler may leave `old` undefined - with bound_ctrl off
// (rounds 4-6) every half was preceded by a v_mov_b32 0 of its destination: four VALU instructions per rotated double instead of two.
template <int N>
__device__ __forceinline__ double row_ror_f64(double v)
{
    static_assert(N >= 1 && N <= 15, "row_ror");
#ifdef INGVIO_DPP32      // A/B switch (tools/build_tu_variant.sh dpp32 kernels_factored -DINGVIO_DPP32): the forms of rounds 4-6
    constexpr bool BC = false;
#else
    constexpr bool BC = true;
#endif
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, BC);
    return __hiloint2double(hi, lo);
}

// value of lane `src` (0..15, a run-time constant after unrolling) of the same 16-lane row.  row_newbcast is the one DPP control
// gfx90a+ executes on 64-bit operands: ONE v_mov_b64_dpp per double (the 32-bit pair with `old` = 0 was four instructions - 270 +
// 270 of the 2900 VALU instructions of a k_feat_gate5<11> wave, all in the register finish of the eliminations)
__device__ __forceinline__ double row_bcast_f64(double v, int src)
{
    switch (src & 15) {
#ifdef INGVIO_DPP32
#define INGVIO_BC(N) case N: return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + N, 0xf, 0xf, false), \
                                                     __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + N, 0xf, 0xf, false));
#else
#define INGVIO_BC(N) case N: return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xf, 0xf, true);
#endif
        INGVIO_BC(0) INGVIO_BC(1) INGVIO_BC(2) INGVIO_BC(3) INGVIO_BC(4) INGVIO_BC(5) INGVIO_BC(6) INGVIO_BC(7)
        INGVIO_BC(8) INGVIO_BC(9) INGVIO_BC(10) INGVIO_BC(11) INGVIO_BC(12) INGVIO_BC(13) INGVIO_BC(14) INGVIO_BC(15)
#undef INGVIO_BC
    }
    return v;
}

// 1/x to full FP64 precision: v_rcp_f64 + two Newton steps (the IEEE division expands to ~3x the latency).  Measured on MI355X
// (tools/micro/rcp_f64.hip, 4 M doubles over 2^+-300): v_rcp_f64 alone 4.6e-8 relative, one step 2.2e-15, two steps 1.1e-16.
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Streaming (non-temporal) accesses for data a kernel touches exactly once - a whole covariance written as the posterior, the prior
// tiles it is computed from.  Write-allocating 240 MB per launch evicts what the NEXT kernels read from L2 / MALL; measured on the
// 512-filter step: 0.730 -> 0.693 ms with k_info_apply alone converted (kernels_factored.hip).
#ifdef INGVIO_NO_NT
#define NT_STORE(p, v) (*(p) = (v))
#define NT_LOAD(p) (*(p))
#else
#define NT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#endif

// phase-timing probes (debug): block (INGVIO_DBG_BLOCK, 0) lane 0 stamps the shader clock; read with ingvio_debug_read
static __device__ long long g_dbg[64];      // one copy per translation unit (no -fgpu-rdc)
__device__ __forceinline__ void dbg_stamp(int slot)
{
#ifdef INGVIO_DBG_STAMPS      // build with INGVIO_DBG_STAMPS=1 python ingvio_amd/build.py --force (tools/gpu_phase_times.py)
#ifndef INGVIO_DBG_BLOCK
#define INGVIO_DBG_BLOCK 0      // a block in the middle of the grid shows contended timings
#endif
#ifndef INGVIO_DBG_BLOCK_Y
#define INGVIO_DBG_BLOCK_Y 0
#endif
    if (blockIdx.x == INGVIO_DBG_BLOCK && blockIdx.y == INGVIO_DBG_BLOCK_Y && threadIdx.x == 0) g_dbg[slot] = clock64();
#else
    (void)slot;
#endif
}
static inline int dbg_read_local(long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(long long) * n);
}
