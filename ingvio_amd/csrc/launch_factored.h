// launch_factored.h — host-side launch descriptor of kernels_factored.hip.
#pragma once
#include "dev_common.h"

#ifndef INGVIO_APPLY_FLAT_NB
#define INGVIO_APPLY_FLAT_NB 64
#endif
constexpr int APPLY_FLAT_NB = INGVIO_APPLY_FLAT_NB;      // up to this many filters per launch take the one-wave-per-tile apply (kernels_factored.hip)

struct FactoredLaunch {
    int stage;            // 0 gate, 1 gram, 2 info solve, 3 info apply (+ downdate); large windows also 5: the part of the solve that needs
                          // the prior only (gauge reference, [Pdd; I] -> [L; L^-T], Pc) - must have run before stage 2, may overlap 0 and 1
    int stereo;
    CovView cv;
    FrameView fv;
    MsckfOpts op;
    int b0, nb, fmax_used;
    double* gamma;
    int* accept;
    int* used;
    double* rec;          // [B][fmax][rec_size] per-feature records (gate -> gram)
    double* Apart;
    double* Asum;         // windows up to 16 clones, G > 1: [nb][rstride] the chunk partials summed before the solve (k_chunk_sum), or nullptr
    int* used_sum;        // [nb]
    double* Tflat;        // windows up to 16 clones, at most flat_nb filters: T = Pc M of the two-launch apply (k_apply_T_flat), [flat_nb][tfstride]
    size_t tfstride;
    int flat_nb;
    int* chunk_used;
    int G, rstride;
    const double* noise;
    double* T;            // M (ncol x mp row-major) followed by t, per filter
    int mstride, n_cap;
    double* Pc;
    int ystride;
    double* dx;
    int* m_out;
    int* nc_out;
    int* status;
    const int* marg_idx;  // stage 3: fused StateManager::marginalize, per filter state index or -1 (nullptr: none)
    int marg_size;
    int* pc_base;         // [nb] stage 2 -> 3: first clone column when Pc is read straight from P, else -1
    int* flip_cnt;        // stage 3 with a fused marginalisation: [nb] arrival counters (zero between launches) - the LAST workgroup of a
                          // filter to START flips the ping-pong halves and shrinks n (what k_post_marg did in a launch of its own), or nullptr
    int* did_flip;        // host flag, set by launch_factored when the stage-3 kernel it chose does that flip
    double* big_sg;       // large-window path: [nb][G][36][36][34] sparse sums (kernels_bigwin.hip)
    double* big_wk;       // large-window path: per-filter solve workspace (bigwin_wk_doubles)
    int ncol_cap;         // 6 * (context c_max): size class of the large-window solve
    int c_used;           // windows up to 16 clones: the largest n_clones among the staged frames [b0, b0 + nb) (0: unknown, use fv.cmax).
                          // The kernels are instantiated per window CLASS (6 / 11 / 16 clones); the class follows the frames, not the
                          // context's capacity: a filter configured for 11 + 1 clones (the shim allocates max_sliding_window_poses + 1)
                          // otherwise ran every update on the 16-clone class - Gauss-Jordan solve, one-feature-per-wave gate
    // in-frame GNSS update (windows up to 16 clones): stage 4 = k_post_cols writes gW [nb][gWstride] (columns gcolmap of the posterior);
    // stage 3 with gY != nullptr folds the rank-16 downdate Yg Yg^T (gm[bl] rows, 0 = none) into the same sweep
    const int* gcolmap;
    const int* gnc;
    int gcstride;
    double* gW;
    size_t gWstride;
    const double* gY;
    size_t gYstride;
    const int* gm;
    // profiling of the gate (stage 0, one kernel whatever the class): when set, the kernel is launched with these events ATTACHED to its
    // dispatch (hipExtLaunchKernelGGL: the kernel's own start / stop timestamps) instead of an event record in front of and behind it -
    // two barrier packets of 2-3 us each per step inside bench.py's timed region (round 6)
    hipEvent_t prof_a, prof_b;
    int* prof_used;       // set to 1 by the launch site that took the events
};
#include <hip/hip_ext.h>
#define LAUNCH_GATE(L, kernel, grid, block, shmem, st, ...)                                                             \
    do {                                                                                                                \
        if ((L).prof_a) { hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, (L).prof_a, (L).prof_b, 0, __VA_ARGS__); if ((L).prof_used) *(L).prof_used = 1; } \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);                                           \
    } while (0)

int launch_factored(const FactoredLaunch& L, hipStream_t st);
// kernels_solve.hip: stage 2 in symmetric (LDL^T) form on the matrix cores; non-zero when the window class is not covered
int launch_info_solve(const FactoredLaunch& L, hipStream_t st);
int dbg_read_solve(long long* out, int n);
int factored_rec_size(int cmax);
int dbg_read_factored(long long* out, int n);

// kernels_bigwin.hip: windows of 17..36 clones
int launch_bigwin(const FactoredLaunch& L, hipStream_t st);
void launch_apply64(const FactoredLaunch& L, hipStream_t st, int mp, double* T, size_t tstride, int ldt);
size_t bigwin_sg_doubles(int G);
size_t bigwin_wk_doubles();
int bigwin_rec_size();
int bigwin_cmax();
int dbg_read_bigwin(long long* out, int n);
