// launch_tri.h — host-side launch descriptor of kernels_tri.hip.
#pragma once
#include "dev_common.h"

struct TriLaunch {
    FrameView fv;                 // clone poses, observation masks, uv of the staged frames
    int b0;
    double R_lr[9], t_lr[3];      // T_cl2cr
    double trans_thres, huber_epsilon, conv_precision, init_damping, max_depth, min_depth;
    int outer_loop_max_iter, inner_loop_max_iter;
    double* pf;                   // [B][fmax][3]  world points (0 on failure)
    int* ok;                      // [B][fmax]
    int mask_failed;              // also clear the observation mask of failed features (they then drop out of the update)
    unsigned long long* mask_rw;  // [B][fmax]
    // update-with-triangulation call (ingvio_msckf_update_tri): the staged frame carries anchors - a point behind its anchor camera
    // fails as well (FeatureInfoManager::triangulateFeatureInfoStereo, MapServerManager.cpp:325) - and the flags / points are
    // mirrored into the result slab the update's fetch copies (nullptr: not mirrored)
    int check_anchor;             // a point behind its anchor camera: ok = 2 (attempt counted, MapServerManager.cpp:287), feature dropped
    const unsigned long long* tri_mask;      // [B][fmax] observations the triangulation uses, or nullptr: fv.obs_mask
    int* ok2;                     // [B][fmax] or nullptr
    double* pf2;                  // [B][fmax][3] or nullptr
};

int launch_triangulate(const TriLaunch& L, int nb, int fmax_used, int stereo, hipStream_t st);
