// launch_tracks.h — host-side descriptors of kernels_tracks.hip (device-resident track store + per-frame delta hand-over).
#pragma once
#include "dev_common.h"

// Per-filter header of a staged delta (ints).  Offsets TRK_OFF_* are positions of the filter's slice in the three upload pools
// (ints / doubles / 64-bit masks); TRK_I_* / TRK_D_* are positions INSIDE that slice.
enum {
    TRK_N_DROP, TRK_APPEND, TRK_N_OBS, TRK_N_FREE, TRK_N_PF, TRK_N_CLONES, TRK_N_FEAT, TRK_K, TRK_HAS_SEL, TRK_MARG,
    TRK_OFF_I, TRK_OFF_D, TRK_OFF_M,
    TRK_I_DROP, TRK_I_FREE, TRK_I_OBS, TRK_I_PF, TRK_I_CIDX, TRK_I_FEAT, TRK_I_GNSS,
    TRK_D_OBS, TRK_D_PF, TRK_D_CR, TRK_D_CP, TRK_D_IMU, TRK_D_STATE,
    TRK_HDR_USED, TRK_HDR = 32
};

struct TrackStage {                     // one staged delta of filters [b0, b0 + nb): views into the uploaded slab
    const int* hdr;                     // [nb][TRK_HDR]
    const int* ipool;
    const double* dpool;
    const unsigned long long* mpool;
};

struct TrackStore {                     // persistent, per context
    double* uv;                         // [B][tmax][cmax][4]
    unsigned long long* mask;           // [B][tmax]
    double* pf;                         // [B][tmax][3]
    int tmax, cmax;
};

struct FrameOut {                       // FrameView with writable pointers (the gather kernel fills the staged frame)
    int* clone_idx; double* clone_R; double* clone_p; int* n_clones; int* n_feat; double* pf; int* anchor; unsigned long long* obs_mask;
    double* uv; int* dof; int cmax, fmax;
};

void launch_imu_steps(const TrackStage& ts, int b0, int nb, int k, double* Phi, double* G, double* dt, double* R, hipStream_t st);
void launch_tracks_apply(const TrackStage& ts, const TrackStore& store, int b0, int nb, hipStream_t st);
void launch_tracks_gather(const TrackStage& ts, const TrackStore& store, const FrameOut& fv, int b0, int nb, int* idx_marg, int* gnss_idx, hipStream_t st);
