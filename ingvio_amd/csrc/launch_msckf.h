// launch_msckf.h — host-side launch descriptor shared by kernels_msckf.hip and capi.hip.
#pragma once
#include "dev_common.h"

struct MsckfLaunch {
    int stage;            // 0 gate, 1 fold, 2 merge, 3 dense fold (qr_compress)
    int stereo;
    CovView cv;
    FrameView fv;
    MsckfOpts op;
    int b0, nb;
    int fmax_used;        // grid.x of the gate kernel
    double* gamma;
    int* accept;
    int* used;
    double* Rpart;
    int* chunk_used;
    int G, rstride;
    double* Hout;
    double* res_out;
    int* colmap;
    int* m_out;
    int* nc_out;
    int mld, hstride, cstride;
    // dense fold
    const double* dH;
    const double* dres;
    int ldh, m, ncol;
};

int msckf_cmax_class(int cmax);
int launch_msckf(const MsckfLaunch& L, hipStream_t st);
